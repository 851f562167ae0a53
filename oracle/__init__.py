"""ctypes face of the CPU oracle (oracle/pic1dp_oracle.c) and of the reference's
own multirand module built into oracle/_ref/.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (pic1dp_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libmultirand_ref.so")

MAX_SPECIES = 8
MAX_MODES = 4096
MAX_INIT_MODES = 16
NSEED = 20635
MAX_OPT = 32


class OrcInput(C.Structure):
    """mirrors struct orc_input (and src/pic1dp_input.F90:32-256)"""
    _fields_ = [
        ("ntime_max", C.c_int32), ("linear", C.c_int32), ("iptcldist", C.c_int32),
        ("nspecies", C.c_int32), ("nmode", C.c_int32), ("init_nmode", C.c_int32),
        ("deltaf", C.c_int32), ("imarker", C.c_int32), ("nx", C.c_int32),
        ("nv", C.c_int32), ("iptclshape", C.c_int32), ("nx_opd", C.c_int32),
        ("nv_opd", C.c_int32), ("multirand_al_int", C.c_int32),
        ("multirand_seed_type", C.c_int32), ("multirand_warmup", C.c_int32),
        ("multirand_selftest", C.c_int32), ("pad0", C.c_int32),
        ("nparticle_max", C.c_int64),
        ("species_nparticle_init", C.c_int64 * MAX_SPECIES),
        ("time_max", C.c_double), ("lx", C.c_double), ("dt", C.c_double),
        ("v_max", C.c_double), ("output_interval", C.c_double),
        ("species_charge", C.c_double * MAX_SPECIES),
        ("species_mass", C.c_double * MAX_SPECIES),
        ("species_temperature", C.c_double * MAX_SPECIES),
        ("species_temperature2", C.c_double * MAX_SPECIES),
        ("species_density", C.c_double * MAX_SPECIES),
        ("species_v0", C.c_double * MAX_SPECIES),
        ("modes", C.c_int32 * MAX_MODES),
        ("init_mode", C.c_int32 * MAX_INIT_MODES),
        ("init_mode_cos", C.c_double * MAX_INIT_MODES),
        ("init_mode_sin", C.c_double * MAX_INIT_MODES),
        ("nmerge", C.c_int32), ("nremove", C.c_int32), ("nsplit", C.c_int32),
        ("typeremove", C.c_int32), ("split_ngroup", C.c_int32), ("pad1", C.c_int32),
        ("remove_frac", C.c_double), ("split_dv_sig_frac", C.c_double),
        ("tmerge", C.c_double * MAX_OPT), ("thshmerge", C.c_double * MAX_OPT),
        ("tremove", C.c_double * MAX_OPT), ("thshremove", C.c_double * MAX_OPT),
        ("tsplit", C.c_double * MAX_OPT), ("thshsplit", C.c_double * MAX_OPT),
    ]


# defaults = the reference's input file (src/pic1dp_input.F90), except the
# reproducible seed_type=1 (the shipped default 3 reads /dev/urandom)
DEFAULTS = dict(
    ntime_max=900000, time_max=500.0, linear=0,
    lx=2.0 * 3.1415926535897932384626 / 0.36, iptcldist=3, nspecies=1,
    species_charge=[-1.0], species_mass=[1.0], species_temperature=[1.0],
    species_temperature2=[1.0], species_density=[0.9], species_v0=[5.0],
    nmode=1, modes=[1], init_nmode=1, init_mode=[1], init_mode_cos=[0.0],
    init_mode_sin=[1e-5], deltaf=1, dt=0.05, nparticle_max=6400000,
    species_nparticle_init=None, imarker=2, v_max=8.0, nx=192, nv=128,
    iptclshape=4, multirand_al_int=3, multirand_seed_type=1,
    multirand_warmup=5, multirand_selftest=1, output_interval=0.5,
    nx_opd=64, nv_opd=64,
    # marker optimisation (src/pic1dp_input.F90:141-206): off by default; the time and
    # threshold lists default to the reference's implied-do formulas when left None
    nmerge=0, nremove=0, nsplit=0, typeremove=2, split_ngroup=5, remove_frac=0.9,
    split_dv_sig_frac=0.1, tmerge=None, thshmerge=None, tremove=None, thshremove=None,
    tsplit=None, thshsplit=None,
)


def opt_defaults(d):
    """the implied-do lists of src/pic1dp_input.F90:149-158,165-180,191-200"""
    for kind, sign in (("merge", None), ("remove", None), ("split", None)):
        n = d["n" + kind]
        if d["t" + kind] is None:
            d["t" + kind] = [50.0 + i * 0.5 for i in range(1, n + 1)]
        if d["thsh" + kind] is None:
            if kind == "split":
                d["thsh" + kind] = [1.0 - 0.9 / max(n, 1) * float(i) for i in range(1, n + 1)]
            else:
                d["thsh" + kind] = [0.1 / max(n, 1) * float(i) for i in range(1, n + 1)]
    return d


def make_input(**kw):
    d = dict(DEFAULTS)
    for k in kw:
        if k not in d:
            raise KeyError(k)
    d.update(kw)
    if d["species_nparticle_init"] is None:
        d["species_nparticle_init"] = [d["nparticle_max"]] * d["nspecies"]
    opt_defaults(d)
    inp = OrcInput()
    for name, _ in OrcInput._fields_:
        if name in ("pad0", "pad1"):
            continue
        val = d[name]
        cur = getattr(inp, name)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(val):
                cur[i] = x
        else:
            setattr(inp, name, val)
    return inp


def build(force=False):
    """compile liboracle.so (and oracle/_ref when the reference tree exists)"""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "pic1dp_oracle.c"))):
        subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(REF_LIB_PATH) and os.path.exists("/root/reference/src/multirand.F90"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_ip32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    IN = C.POINTER(OrcInput)
    sig = {
        "orc_multirand_new": (P, []),
        "orc_multirand_free": (None, [P]),
        "orc_multirand_init": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_multirand_selftest": (C.c_int, [P, C.c_int]),
        "orc_multirand_default_seeds": (None, [P, C.c_int]),
        "orc_multirand_int64": (C.c_int64, [P]),
        "orc_multirand_real64": (C.c_double, [P]),
        "orc_multirand_int_array64": (None, [P, _ip64, C.c_int64]),
        "orc_multirand_real_array64": (None, [P, _dp, C.c_int64]),
        "orc_multirand_gaussian_array64": (None, [P, _dp, C.c_int64]),
        "orc_local_size": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
        "orc_particle_np": (C.c_int64, [IN, C.c_int, C.c_int, C.c_int]),
        "orc_particle_load_species": (None, [IN, C.c_int, P, C.c_int64, _dp, _dp, _dp, _dp]),
        "orc_deposit_species": (None, [IN, C.c_int64, _dp, _dp, _dp]),
        "orc_deposit_species_idx": (None, [IN, C.c_int64, _dp, _dp, _dp, _ip32, _ip64]),
        "orc_chargeden_from_charge": (None, [IN, _dp, _dp]),
        "orc_exp_array": (None, [_dp, _dp, C.c_int64]),
        "orc_push_backup": (None, [C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int]),
        "orc_push_species": (None, [IN, C.c_int, C.c_int, _dp, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
        "orc_field_new": (P, [IN]),
        "orc_field_free": (None, [P]),
        "orc_field_solve": (None, [IN, P, _dp, _dp, _dp, _dp]),
        "orc_field_solve_ranks": (None, [IN, P, C.c_int, _dp, _dp, _dp, _dp]),
        "orc_field_energy": (C.c_double, [IN, _dp]),
        "orc_field_solve_fd": (None, [IN, _dp, _dp]),
        "orc_energy_sums": (None, [C.c_int64, _dp, _dp, _dp, C.c_int, _dp]),
        "orc_ptcldist": (None, [IN, C.c_int64, _dp, _dp, _dp, _dp] + [_dp] * 6),
        "orc_sim_new": (P, [IN, C.c_int]),
        "orc_sim_free": (None, [P]),
        "orc_sim_load": (C.c_int, [P]),
        "orc_sim_set_threads": (None, [P, C.c_int]),
        "orc_sim_collect_charge": (None, [P]),
        "orc_sim_solve_field": (None, [P]),
        "orc_sim_push": (None, [P, C.c_int]),
        "orc_sim_step": (None, [P, C.c_int]),
        "orc_sim_itime": (C.c_int32, [P]),
        "orc_sim_time": (C.c_double, [P]),
        "orc_sim_field_energy": (C.c_double, [P]),
        "orc_sim_get_field": (None, [P, _dp, _dp, _dp, _dp]),
        "orc_sim_set_field": (None, [P, _dp]),
        "orc_sim_rank_np": (C.c_int64, [P, C.c_int, C.c_int]),
        "orc_sim_rank_nalloc": (C.c_int64, [P, C.c_int]),
        "orc_sim_array": (C.POINTER(C.c_double), [P, C.c_int, C.c_int, C.c_int]),
        "orc_sim_energy_sums": (None, [P, C.c_int, _dp]),
        "orc_ptcldist_finish": (None, [IN, C.c_int] + [_dp] * 6),
        "orc_sim_ptcldist": (None, [P, C.c_int, C.c_int] + [_dp] * 6),
        "orc_sim_output_scalars": (None, [P, _dp]),
        "orc_dist_pertb_abs_v": (None, [IN, C.c_int64, _dp, _dp, _dp]),
        "orc_particle_merge": (None, [IN, C.c_double, _dp, C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]),
        "orc_particle_remove": (None, [IN, C.c_double, _dp, P, C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]),
        "orc_particle_split": (None, [IN, C.c_double, _dp, P, C.c_int64, C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]),
        "orc_sim_optimize": (C.c_int, [P, C.c_int]),
        "orc_sim_rank_rng": (P, [P, C.c_int]),
        "orc_sim_set_rank_np": (None, [P, C.c_int, C.c_int, C.c_int64]),
        "orc_check_termination": (C.c_int, [IN, C.c_int32, C.c_double]),
        "orc_output_due": (C.c_int, [IN, C.c_double, C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


class Multirand:
    """restated multirand generator (src/multirand.F90)"""

    def __init__(self):
        self.g = lib().orc_multirand_new()

    def __del__(self):
        if getattr(self, "g", None):
            lib().orc_multirand_free(self.g)
            self.g = None

    def init(self, al_int=3, seed_type=1, mype=0, warmup=5, selftest=True):
        return lib().orc_multirand_init(self.g, al_int, seed_type, mype, warmup, int(selftest))

    def selftest(self, al_int):
        return lib().orc_multirand_selftest(self.g, al_int)

    def default_seeds(self, al_int):
        lib().orc_multirand_default_seeds(self.g, al_int)

    def int64(self):
        return lib().orc_multirand_int64(self.g)

    def int_array(self, n):
        a = np.empty(n, dtype=np.int64)
        lib().orc_multirand_int_array64(self.g, a, n)
        return a

    def real_array(self, n):
        a = np.empty(n, dtype=np.float64)
        lib().orc_multirand_real_array64(self.g, a, n)
        return a

    def gaussian_array(self, n):
        a = np.empty(n, dtype=np.float64)
        lib().orc_multirand_gaussian_array64(self.g, a, n)
        return a


class RefMultirand:
    """the reference's own multirand module (oracle/_ref, flang build).
    Module-level state: one generator per process, like the reference."""

    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(REF_LIB_PATH)

    def __init__(self):
        if RefMultirand._lib is None:
            L = C.CDLL(REF_LIB_PATH)
            L.ref_multirand_init.argtypes = [C.c_int] * 5
            L.ref_multirand_init.restype = None
            L.ref_multirand_int_array64.argtypes = [_ip64, C.c_int64]
            L.ref_multirand_int_array64.restype = None
            L.ref_multirand_real_array64.argtypes = [_dp, C.c_int64]
            L.ref_multirand_real_array64.restype = None
            L.ref_multirand_gaussian_array64.argtypes = [_dp, C.c_int64]
            L.ref_multirand_gaussian_array64.restype = None
            RefMultirand._lib = L
        self.L = RefMultirand._lib

    def init(self, al_int=3, seed_type=1, mype=0, warmup=5, selftest=True):
        self.L.ref_multirand_init(al_int, seed_type, mype, warmup, int(selftest))

    def int_array(self, n):
        a = np.empty(n, dtype=np.int64)
        self.L.ref_multirand_int_array64(a, n)
        return a

    def real_array(self, n):
        a = np.empty(n, dtype=np.float64)
        self.L.ref_multirand_real_array64(a, n)
        return a

    def gaussian_array(self, n):
        a = np.empty(n, dtype=np.float64)
        self.L.ref_multirand_gaussian_array64(a, n)
        return a


class Field:
    def __init__(self, inp):
        self.inp = inp
        self.f = lib().orc_field_new(C.byref(inp))

    def __del__(self):
        if getattr(self, "f", None):
            lib().orc_field_free(self.f)
            self.f = None

    def tables(self):
        """(fourier_re, fourier_im, grad_inv) of the orc_field, copied: [nx, nmode], [nx, nmode], [nmode]"""
        class _F(C.Structure):
            _fields_ = [("nx", C.c_int32), ("nmode", C.c_int32), ("re", C.POINTER(C.c_double)),
                        ("im", C.POINTER(C.c_double)), ("ginv", C.POINTER(C.c_double))]
        f = C.cast(self.f, C.POINTER(_F)).contents
        n = f.nx * f.nmode
        return (np.ctypeslib.as_array(f.re, (n,)).reshape(f.nx, f.nmode).copy(),
                np.ctypeslib.as_array(f.im, (n,)).reshape(f.nx, f.nmode).copy(),
                np.ctypeslib.as_array(f.ginv, (f.nmode,)).copy())

    def solve(self, rho, npe=1):
        """field_solve_electric in the summation order of an npe-rank reference run (1: SeqAIJ)"""
        nx, nm = self.inp.nx, self.inp.nmode
        E = np.empty(nx)
        re = np.empty(nm)
        im = np.empty(nm)
        lib().orc_field_solve_ranks(C.byref(self.inp), self.f, npe, np.ascontiguousarray(rho, dtype=np.float64), E, re, im)
        return E, re, im


ARR = dict(x=0, v=1, p=2, w=3, xb=4, vb=5, wb=6)


class Sim:
    """the reference driver (src/pic1dp.F90:64-109) on npe virtual ranks"""

    def __init__(self, inp, npe=1, nthreads=1):
        self.inp = inp
        self.npe = npe
        self.s = lib().orc_sim_new(C.byref(inp), npe)
        lib().orc_sim_set_threads(self.s, nthreads)

    def __del__(self):
        if getattr(self, "s", None):
            lib().orc_sim_free(self.s)
            self.s = None

    def load(self):
        return lib().orc_sim_load(self.s)

    def collect_charge(self):
        lib().orc_sim_collect_charge(self.s)

    def solve_field(self):
        lib().orc_sim_solve_field(self.s)

    def push(self, irk):
        lib().orc_sim_push(self.s, irk)

    def step(self, n=1):
        lib().orc_sim_step(self.s, n)

    def optimize(self, irk=2):
        """particle_optimize on every rank; True when a merge/remove/split ran"""
        return bool(lib().orc_sim_optimize(self.s, irk))

    def set_rank_np(self, rank, np_valid, isp=0):
        lib().orc_sim_set_rank_np(self.s, rank, isp, np_valid)

    @property
    def itime(self):
        return lib().orc_sim_itime(self.s)

    @property
    def time(self):
        return lib().orc_sim_time(self.s)

    def field_energy(self):
        return lib().orc_sim_field_energy(self.s)

    def get_field(self):
        nx, nm = self.inp.nx, self.inp.nmode
        E, rho, re, im = np.empty(nx), np.empty(nx), np.empty(nm), np.empty(nm)
        lib().orc_sim_get_field(self.s, E, rho, re, im)
        return E, rho, re, im

    def set_field(self, E):
        lib().orc_sim_set_field(self.s, np.ascontiguousarray(E, dtype=np.float64))

    def rank_np(self, rank, isp=0):
        return lib().orc_sim_rank_np(self.s, rank, isp)

    def rank_nalloc(self, rank):
        return lib().orc_sim_rank_nalloc(self.s, rank)

    def array(self, rank, isp, which):
        """numpy view (no copy) of a rank-owned particle array"""
        n = self.rank_nalloc(rank)
        ptr = lib().orc_sim_array(self.s, rank, isp, ARR[which] if isinstance(which, str) else which)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    def gather(self, which, isp=0):
        """concatenate the valid (np) part of every rank block"""
        return np.concatenate([self.array(r, isp, which)[: self.rank_np(r, isp)].copy()
                               for r in range(self.npe)])

    def energy_sums(self, isp=0):
        out = np.empty(3)
        lib().orc_sim_energy_sums(self.s, isp, out)
        return out

    def output_scalars(self):
        out = np.empty(2 + 3 * self.inp.nspecies)
        lib().orc_sim_output_scalars(self.s, out)
        return out

    def ptcldist(self, isp=0, finish=True):
        nxo, nvo = self.inp.nx_opd, self.inp.nv_opd
        names = ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v")
        out = [np.zeros(nxo * nvo) for _ in range(3)] + [np.zeros(nvo) for _ in range(3)]
        lib().orc_sim_ptcldist(self.s, isp, int(finish), *out)
        return dict(zip(names, out))


def growthrate_energy_fit(t, energy, time1, time2):
    """least-squares slope of ln(int E^2 dx) over [time1, time2): the definition
    of tools/OutputData.py:153-170 (gamma of the mode = half of it,
    tools/runinfo.py:116)."""
    t = np.asarray(t)
    energy = np.asarray(energy)
    i1 = int(np.searchsorted(t, time1)) - 1
    i2 = int(np.searchsorted(t, time2))
    tt = t[i1:i2]
    ln = np.log(energy[i1:i2])
    n = i2 - i1
    return (n * np.sum(tt * ln) - np.sum(tt) * np.sum(ln)) / (n * np.sum(tt * tt) - np.sum(tt) ** 2)
