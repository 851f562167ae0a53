"""ctypes binding of libpic1dp_probe.so (include/pic1dp_probe.h): MEASUREMENT and test support.

Streaming-rate probes with the marker kernels' access shapes (bench.py's second roofline
denominator, tools/) and array evaluations of the device functions the marker kernels call
(the parity tests).  Not imported by the package: the product is libpic1dp_hip.so alone.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIC1DP_PROBE_LIB") or os.path.join(HERE, "lib", "libpic1dp_probe.so")


class Species(C.Structure):
    """struct pic1dp_probe_species: one species of the input (src/pic1dp_input.F90:43-72)"""
    _fields_ = [("iptcldist", C.c_int32), ("charge", C.c_double), ("mass", C.c_double),
                ("temperature", C.c_double), ("temperature2", C.c_double), ("density", C.c_double),
                ("v0", C.c_double)]


_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)
_SP = C.POINTER(Species)
SIGNATURES = {
    "pic1dp_probe_last_error": [],
    "pic1dp_probe_stream": [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _D],
    "pic1dp_probe_layout": [C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _D],
    "pic1dp_probe_release": [],
    "pic1dp_probe_div_lx": [C.c_int32, C.c_double, C.c_int32, C.c_int64, C.c_uint64, _I64],
    "pic1dp_probe_host_div_lx": [C.c_double, C.c_int32, C.c_int64, C.c_uint64, _I64],
    "pic1dp_probe_div_const": [C.c_int32, C.c_double, C.c_int64, C.c_uint64, _I64],
    "pic1dp_probe_host_div_const": [C.c_double, C.c_int64, C.c_uint64, _I64],
    "pic1dp_probe_host_optimize": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_uint64, C.c_int64,
                                   C.c_int64, _I64, _I64],
    "pic1dp_probe_exp": [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64],
    "pic1dp_probe_species_const": [_SP, _I32, _I32, _I32, _I32, _D],
    "pic1dp_probe_dlnf0": [C.c_int32, _SP, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64],
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpic1dp_probe.so not found at %s -- build it with `python pic1dp_amd/build.py`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "pic1dp_probe_last_error" else C.c_int
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("pic1dp_probe: %s" % (load().pic1dp_probe_last_error() or b"").decode())


def species(inp, isp=0):
    """the probe's species struct from an input object (field names of src/pic1dp_input.F90)"""
    return Species(inp.iptcldist, inp.species_charge[isp], inp.species_mass[isp], inp.species_temperature[isp],
                   inp.species_temperature2[isp], inp.species_density[isp], inp.species_v0[isp])


def stream(nread, nwrite, n, reps=10, device=0, blocks=0, threads=0, variant=None):
    """GB/s of a pure streaming pass: nread arrays of n doubles read, nwrite written (non-temporal by default;
    PIC1DP_PROBE_VARIANT: 0 plain, 2 plain with two pairs per lane)"""
    if variant is None:
        variant = int(os.environ.get("PIC1DP_PROBE_VARIANT", "1"))
    g = C.c_double()
    _check(load().pic1dp_probe_stream(device, nread, nwrite, int(n), reps, blocks, threads, variant, C.byref(g)))
    return g.value


def layout(n, log2_tile=12, reps=10, device=0, stagger_bytes=0, keep=0, blocks=0, threads=0):
    """ms per launch of the second sub-step kernel's traffic (4 arrays read, 3 written back in place) over a fresh
    slab: [arrays apart r/w, tiled r/w, arrays apart read-only, tiled read-only, tiled r/w one workgroup per tile,
    the same read-only]"""
    ms = (C.c_double * 6)()
    _check(load().pic1dp_probe_layout(device, int(n), log2_tile, int(stagger_bytes), reps, keep, blocks, threads, ms))
    return list(ms)


def release():
    _check(load().pic1dp_probe_release())


def div_lx_mismatches(lx, nx, n, seed, device=0, host=False):
    m = C.c_int64(-1)
    L = load()
    if host:
        _check(L.pic1dp_probe_host_div_lx(lx, nx, n, seed, C.byref(m)))
    else:
        _check(L.pic1dp_probe_div_lx(device, lx, nx, n, seed, C.byref(m)))
    return m.value


def div_const_mismatches(divisor, n, seed, device=0, host=False):
    m = C.c_int64(-1)
    L = load()
    if host:
        _check(L.pic1dp_probe_host_div_const(divisor, n, seed, C.byref(m)))
    else:
        _check(L.pic1dp_probe_div_const(device, divisor, n, seed, C.byref(m)))
    return m.value


def device_exp(x, device=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    _check(load().pic1dp_probe_exp(device, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size))
    return y


def species_const(sp):
    """dict(pow2, unit, fastc, one_exp, f=[fq2, fq1, fq0, fm1, fm0, fd1, fd0]) as the library would form them"""
    a, b, c_, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    f = (C.c_double * 7)()
    _check(load().pic1dp_probe_species_const(C.byref(sp), C.byref(a), C.byref(b), C.byref(c_), C.byref(d), f))
    return dict(pow2=a.value, unit=b.value, fastc=c_.value, one_exp=d.value, f=list(f))


def dlnf0(sp, v, form, device=0):
    """-f0'/f0 at v as the marker kernels evaluate it: form 0 reference operation order, 1 one-exp form"""
    v = np.ascontiguousarray(v, dtype=np.float64)
    y = np.empty_like(v)
    _check(load().pic1dp_probe_dlnf0(device, C.byref(sp), form, v.ctypes.data_as(C.c_void_p),
                                     y.ctypes.data_as(C.c_void_p), v.size))
    return y


def host_optimize_mismatches(kind, np_, nalloc, threshold, seed=1, typeremove=2, nx=32, nv=64, split_ngroup=3):
    """the key-walking planners of the GPU marker optimisation against the host routines on whole markers (0 merge, 1
    remove, 2 split), on the host: (slots that differ (-1 / -2: marker counts / random stream differ), markers after the event)"""
    m, after = C.c_int64(), C.c_int64()
    _check(load().pic1dp_probe_host_optimize(kind, typeremove, nx, nv, split_ngroup, threshold, seed, int(np_), int(nalloc),
                                             C.byref(m), C.byref(after)))
    return m.value, after.value
