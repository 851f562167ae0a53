"""Host-side mirror of the reference's module procedures for the time-step hot
path, on top of the C ABI (include/pic1dp_hip.h).

Method names are the reference's procedure names so that a parity test reads
like the reference driver (src/pic1dp.F90:64-109):

    sim = Pic1dp(make_input(nparticle_max=10**7, nx=256))
    sim.particle_load()
    sim.interaction_collect_charge(); sim.field_solve_electric()
    for irk in (1, 2):
        sim.interaction_push_particle(irk)
        sim.interaction_collect_charge(); sim.field_solve_electric()

All compute runs in hand-written HIP kernels on one MI355X; nothing here
computes, and there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Input, Layout, Pic1dpError, check  # noqa: F401


def make_input(**kw):
    """pic1dp_input with the reference's default input file
    (src/pic1dp_input.F90) except multirand_seed_type=1; keyword names are the
    reference names without the input_ prefix.  Array-valued parameters take
    sequences."""
    L = _lib.load()
    inp = Input()
    check(L.pic1dp_hip_input_defaults(C.byref(inp)))
    names = {n for n, _ in Input._fields_}
    explicit_init = "species_nparticle_init" in kw
    for k, val in kw.items():
        if k not in names or k == "abi_version":
            raise KeyError("unknown input parameter %r" % k)
        cur = getattr(inp, k)
        if hasattr(cur, "__len__"):
            if len(val) > len(cur):
                raise ValueError("%s: at most %d entries" % (k, len(cur)))
            for i, x in enumerate(val):
                cur[i] = x
        else:
            setattr(inp, k, val)
    if not explicit_init:
        # input_species_nparticle_init = input_nparticle_max (src/pic1dp_input.F90:117)
        for s in range(max(0, min(inp.nspecies, _lib.MAX_SPECIES))):
            inp.species_nparticle_init[s] = inp.nparticle_max
    # marker-optimisation lists not given: the reference's implied-do formulas of the
    # counts (src/pic1dp_input.F90:149-158, 165-180, 191-200)
    for kind in ("merge", "remove", "split"):
        n = max(0, min(getattr(inp, "n" + kind), _lib.MAX_OPT))
        if "t" + kind not in kw:
            for i in range(1, n + 1):
                getattr(inp, "t" + kind)[i - 1] = 50.0 + i * 0.5
        if "thsh" + kind not in kw:
            for i in range(1, n + 1):
                val = 1.0 - 0.9 / max(n, 1) * float(i) if kind == "split" else 0.1 / max(n, 1) * float(i)
                getattr(inp, "thsh" + kind)[i - 1] = val
    return inp


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Pic1dp:
    """one process = one GPU = one context"""

    def __init__(self, inp, rank=0, nranks=1, npe=0, device=-1):
        self.L = _lib.load()
        self.inp = inp
        self.rank, self.nranks = rank, nranks
        self.npe = npe or nranks
        self._ctx = C.c_void_p()
        lay = Layout(rank, nranks, self.npe, device)
        check(self.L.pic1dp_hip_create(C.byref(inp), C.byref(lay), C.byref(self._ctx)))

    # -- life cycle (particle_final / field_final) ---------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self.L.pic1dp_hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- sizes -----------------------------------------------------------------
    def local_sizes(self, ispecies=0):
        na, npv = C.c_int64(), C.c_int64()
        check(self.L.pic1dp_hip_local_sizes(self._ctx, ispecies, C.byref(na), C.byref(npv)))
        return na.value, npv.value

    # -- particle_load (src/pic1dp_particle.F90:145-269) -----------------------
    def particle_load(self):
        check(self.L.pic1dp_hip_particle_load(self._ctx))

    def set_seed_offset(self, offset):
        """ensemble member: reference block b of the next particle_load draws from RNG stream mype = b + offset
        (0: the reference's constant-seed run)"""
        check(self.L.pic1dp_hip_set_seed_offset(self._ctx, int(offset)))

    def particles_upload(self, x, v, p, w, ispecies=0, np_valid=None):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, v, p, w)]
        n = arrs[0].size
        if any(a.size != n for a in arrs):
            raise ValueError("x, v, p, w must have one length")
        check(self.L.pic1dp_hip_particles_upload(
            self._ctx, ispecies, *[_ptr(a) for a in arrs], n, n if np_valid is None else np_valid))

    def particles_download(self, ispecies=0):
        n, _ = self.local_sizes(ispecies)
        out = [np.empty(n) for _ in range(4)]
        check(self.L.pic1dp_hip_particles_download(self._ctx, ispecies, *[_ptr(a) for a in out], n))
        return dict(zip("xvpw", out))

    def particles_download_bak(self, ispecies=0):
        n, npv = self.local_sizes(ispecies)
        out = [np.zeros(n) for _ in range(3)]
        check(self.L.pic1dp_hip_particles_download_bak(self._ctx, ispecies, *[_ptr(a) for a in out], n))
        return dict(zip(("xb", "vb", "wb"), [a[:npv] for a in out]))

    # -- the hot path, under the reference's names -----------------------------
    def interaction_collect_charge(self):
        check(self.L.pic1dp_hip_collect_charge(self._ctx))

    def field_solve_electric(self):
        check(self.L.pic1dp_hip_solve_field(self._ctx))

    def interaction_push_particle(self, irk):
        check(self.L.pic1dp_hip_push(self._ctx, irk))

    def particle_optimize(self, irk=2):
        """particle_optimize (src/pic1dp_particle.F90:724-783) after the push of
        sub-step irk; True when a merge / remove / split was performed"""
        flag = C.c_int32()
        check(self.L.pic1dp_hip_particle_optimize(self._ctx, irk, C.byref(flag)))
        return bool(flag.value)

    def substep(self, irk):
        """push(irk) + collect_charge + solve_field, push and deposit fused"""
        check(self.L.pic1dp_hip_substep(self._ctx, irk))

    def step(self, nsteps=1):
        check(self.L.pic1dp_hip_step(self._ctx, nsteps))

    def sync(self):
        check(self.L.pic1dp_hip_sync(self._ctx))

    def set_step_mode(self, mode):
        """0: whole-step kernels (half-step state recomputed, default);
        1: two fused sub-steps through the RK ping-pong sets"""
        check(self.L.pic1dp_hip_set_step_mode(self._ctx, mode))

    def chargeden_kept_mode_only(self):
        """True: field_chargeden holds only the kept mode's content of the half-step charge density (between the
        sub-steps of a step whose half-step charge was predicted as six sums, where the library could not rebuild the
        reference's vector: several ranks); False: it is the reference's vector"""
        k = C.c_int32()
        check(self.L.pic1dp_hip_chargeden_state(self._ctx, C.byref(k)))
        return bool(k.value)

    def predict_kind(self):
        """how step mode 0 predicts the next first sub-step's charge: 0 not (two passes per step),
        1 prediction tiles (k_step_one), 2 six sums (k_step_sums, large grids)"""
        k = C.c_int32(0)
        check(self.L.pic1dp_hip_predict_kind(self._ctx, C.byref(k)))
        return k.value

    def set_output_fusion(self, on=True):
        """diagnostics of output_all taken inside the step that precedes it: False / 0 never, True / 1 where it pays
        (not on a predicted one-pass step, whose prediction k_step_full<DIAG> would cost), 2 always"""
        check(self.L.pic1dp_hip_set_output_fusion(self._ctx, int(on)))

    def set_field_solver(self, kind):
        """0: the reference's mode-filter solve (default); 1: opt-in finite-difference
        tridiagonal solve by parallel cyclic reduction (not in the reference)"""
        check(self.L.pic1dp_hip_set_field_solver(self._ctx, kind))

    def get_field_half(self):
        """field_electric between the two sub-steps of the last step()"""
        E = np.empty(self.inp.nx)
        check(self.L.pic1dp_hip_get_field_half(self._ctx, _ptr(E)))
        return E

    # -- driver scalars (global_itime, global_time) -----------------------------
    @property
    def itime(self):
        it, t = C.c_int32(), C.c_double()
        check(self.L.pic1dp_hip_get_time(self._ctx, C.byref(it), C.byref(t)))
        return it.value

    @property
    def time(self):
        it, t = C.c_int32(), C.c_double()
        check(self.L.pic1dp_hip_get_time(self._ctx, C.byref(it), C.byref(t)))
        return t.value

    def set_time(self, itime, time):
        check(self.L.pic1dp_hip_set_time(self._ctx, itime, time))

    def check_termination(self):
        f = C.c_int32()
        check(self.L.pic1dp_hip_check_termination(self._ctx, C.byref(f)))
        return f.value

    def output_due(self, itermination=0):
        f = C.c_int32()
        check(self.L.pic1dp_hip_output_due(self._ctx, itermination, C.byref(f)))
        return f.value

    def steps_to_output(self):
        """iterations of the driver loop (src/pic1dp.F90:78-109) up to and including the one output_all follows"""
        n = C.c_int32()
        check(self.L.pic1dp_hip_steps_to_output(self._ctx, C.byref(n)))
        return n.value

    def check_state(self, deep=True):
        """debugging aid: the relations between the flags of the library's state machine that hold between any two
        calls; deep also looks at the device's accumulator sets.  Raises Pic1dpError naming the
        relation that does not hold."""
        check(self.L.pic1dp_hip_check_state(self._ctx, 1 if deep else 0))

    # -- field access -------------------------------------------------------------
    def get_field(self, chargeden=True):
        """field_electric, field_chargeden, field_mode_re / _im (src/pic1dp_field.F90:27-31).  chargeden=False
        leaves field_chargeden alone (no key in the result): asking for it between the sub-steps of a step
        whose half-step charge was predicted as six sums makes the library push the half-step state into
        memory and deposit it after all -- the reference's vector, at the price of that step's short cut"""
        nx, nm = self.inp.nx, self.inp.nmode
        E, re, im = np.empty(nx), np.empty(nm), np.empty(nm)
        cd = np.empty(nx) if chargeden else None
        check(self.L.pic1dp_hip_get_field(self._ctx, _ptr(E), _ptr(cd) if chargeden else None, _ptr(re), _ptr(im)))
        out = dict(electric=E, mode_re=re, mode_im=im)
        if chargeden:
            out["chargeden"] = cd
        return out

    def set_electric(self, E):
        E = np.ascontiguousarray(E, dtype=np.float64)
        if E.size != self.inp.nx:
            raise ValueError("E must have nx entries")
        check(self.L.pic1dp_hip_set_electric(self._ctx, _ptr(E)))

    def set_chargeden(self, cd):
        cd = np.ascontiguousarray(cd, dtype=np.float64)
        if cd.size != self.inp.nx:
            raise ValueError("chargeden must have nx entries")
        check(self.L.pic1dp_hip_set_chargeden(self._ctx, _ptr(cd)))

    def field_energy(self):
        e = C.c_double()
        check(self.L.pic1dp_hip_field_energy(self._ctx, C.byref(e)))
        return e.value

    def energy_history(self):
        cnt = C.c_int64()
        check(self.L.pic1dp_hip_energy_history(self._ctx, None, 0, C.byref(cnt)))
        out = np.empty(cnt.value)
        if cnt.value:
            check(self.L.pic1dp_hip_energy_history(self._ctx, _ptr(out), cnt.value, C.byref(cnt)))
        return out

    def energy_history_reset(self):
        check(self.L.pic1dp_hip_energy_history_reset(self._ctx))

    def energy_sums(self, ispecies=0):
        out = np.empty(3)
        check(self.L.pic1dp_hip_energy_sums(self._ctx, ispecies, _ptr(out)))
        return out

    def cell_indices(self, ispecies=0, want_ix=True):
        """cell of every valid marker and markers per cell (want_ix=False: the counts only -- no host array per marker)"""
        _, npv = self.local_sizes(ispecies)
        ix = np.empty(npv, dtype=np.int32) if want_ix else None
        cnt = np.empty(self.inp.nx, dtype=np.int64)
        check(self.L.pic1dp_hip_cell_indices(self._ctx, ispecies, _ptr(ix) if want_ix else None, _ptr(cnt)))
        return ix, cnt

    # -- diagnostics of output_all (src/pic1dp_output.F90) ------------------------
    def output_scalars(self):
        """realbuf of output_field: time, int E^2 dx, then 3 sums per species"""
        n = 2 + 3 * self.inp.nspecies
        out = np.empty(n)
        check(self.L.pic1dp_hip_output_scalars(self._ctx, _ptr(out), n))
        return out

    def ptcldist(self, ispecies=0, finish=True):
        """(x,v) and v distributions of output_ptcldist, computed on the GPU"""
        nxo, nvo = self.inp.nx_opd, self.inp.nv_opd
        names = ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v")
        out = [np.empty(nxo * nvo) for _ in range(3)] + [np.empty(nvo) for _ in range(3)]
        check(self.L.pic1dp_hip_ptcldist(self._ctx, ispecies, int(finish), *[_ptr(a) for a in out]))
        return dict(zip(names, out))

    def output_all(self):
        """everything output_all writes in one call and one wait: (scalars, field dict, [per-species ptcldist dicts])"""
        inp = self.inp
        ns, nx, nm = inp.nspecies, inp.nx, inp.nmode
        nxv, nvo = inp.nx_opd * inp.nv_opd, inp.nv_opd
        ntot = 3 * nxv + 3 * nvo
        scal = np.empty(2 + 3 * ns)
        E, cd, re, im = np.empty(nx), np.empty(nx), np.empty(nm), np.empty(nm)
        dist = np.empty(ns * ntot)
        check(self.L.pic1dp_hip_output_all(self._ctx, _ptr(scal), scal.size, _ptr(E), _ptr(cd), _ptr(re), _ptr(im), _ptr(dist)))
        names = ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v")
        per = []
        for s in range(ns):
            d = dist[s * ntot:(s + 1) * ntot]
            parts = [d[0:nxv], d[nxv:2 * nxv], d[2 * nxv:3 * nxv], d[3 * nxv:3 * nxv + nvo], d[3 * nxv + nvo:3 * nxv + 2 * nvo],
                     d[3 * nxv + 2 * nvo:]]
            per.append(dict(zip(names, parts)))
        return scal, dict(electric=E, chargeden=cd, mode_re=re, mode_im=im), per

    # -- split-phase diagnostics (a host that owns the reductions) --------------------
    def output_scalars_from(self, sums):
        """realbuf of output_field from the kinetic sums (energy_sums per species) summed over ranks"""
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        n = 2 + 3 * self.inp.nspecies
        out = np.empty(n)
        check(self.L.pic1dp_hip_output_scalars_from(self._ctx, _ptr(sums), _ptr(out), n))
        return out

    def ptcldist_finish(self, raw, ispecies=0):
        """ptcldist(finish=False) summed over ranks -> what output_ptcldist writes"""
        names = ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v")
        out = [np.array(raw[k], dtype=np.float64).ravel().copy() for k in names]
        check(self.L.pic1dp_hip_ptcldist_finish(self._ctx, ispecies, *[_ptr(a) for a in out]))
        return dict(zip(names, out))

    # -- split-phase deposit -------------------------------------------------------
    def charge_local(self):
        out = np.empty(self.inp.nx)
        check(self.L.pic1dp_hip_charge_local(self._ctx, _ptr(out)))
        return out

    def charge_reduced(self, charge1):
        a = np.ascontiguousarray(charge1, dtype=np.float64)
        if a.size != self.inp.nx:
            raise ValueError("charge must have nx entries")
        check(self.L.pic1dp_hip_charge_reduced(self._ctx, _ptr(a)))

    # -- RCCL ------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        check(self.L.pic1dp_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid):
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.COMM_ID_BYTES)
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        check(self.L.pic1dp_hip_comm_init(self._ctx, buf))

    def comm_available(self):
        """raises unless librccl can be loaded in this process"""
        check(self.L.pic1dp_hip_comm_available())

    # -- one-hop charge exchange (alternative to the RCCL all-reduce) ---------------
    def xchg_create(self):
        buf = (C.c_ubyte * _lib.XCHG_HANDLE_BYTES)()
        check(self.L.pic1dp_hip_xchg_create(self._ctx, buf))
        return bytes(buf)

    def xchg_connect(self, handles):
        if len(handles) != _lib.XCHG_HANDLE_BYTES * self.nranks:
            raise ValueError("need %d handle bytes per rank" % _lib.XCHG_HANDLE_BYTES)
        buf = (C.c_ubyte * len(handles)).from_buffer_copy(handles)
        check(self.L.pic1dp_hip_xchg_connect(self._ctx, buf))

    def set_allreduce(self, kind):
        """0 auto (RCCL when a communicator exists), 1 RCCL, 2 one-hop exchange"""
        check(self.L.pic1dp_hip_set_allreduce(self._ctx, kind))

    def xchg_info(self):
        kind, n = C.c_int32(), C.c_int64()
        check(self.L.pic1dp_hip_xchg_info(self._ctx, C.byref(kind), C.byref(n)))
        return kind.value, n.value

    def xchg_time(self, reset=False):
        """(ms, exchanges): device time spent inside the one-hop exchanges enqueued while the timers were on (the
        stores into the peers' slots, the wait for their flags, the rank-order sum) -- what splits the charge sum
        from the field solve inside the launch they share"""
        ms, n = C.c_double(), C.c_int64()
        check(self.L.pic1dp_hip_xchg_time(self._ctx, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # -- timers / knobs -----------------------------------------------------------------
    def timers_enable(self, on=True):
        """False / 0 off; True / 1 HIP events around every launch; n >= 2 around every n-th block of 64 launches of a timer id (scaled)"""
        check(self.L.pic1dp_hip_timers_enable(self._ctx, int(on)))

    def timer_ms(self, iwt):
        ms = C.c_double()
        check(self.L.pic1dp_hip_timer_ms(self._ctx, iwt, C.byref(ms)))
        return ms.value

    def timers_reset(self):
        check(self.L.pic1dp_hip_timers_reset(self._ctx))

    def set_launch(self, threads=0, blocks_per_cu=0):
        check(self.L.pic1dp_hip_set_launch(self._ctx, threads, blocks_per_cu))

    def kernel_stats_enable(self, on=True):
        check(self.L.pic1dp_hip_kernel_stats_enable(self._ctx, int(on)))

    def kernel_bytes(self, which=6):
        """what the marker kernel launched last under `which` (kernel_stats numbering) moves per marker and
        launch: dict(read, written, carry, name) -- read + written are compulsory for its data flow, carry
        is the traffic it chooses to spend on not evaluating -f0'/f0 again"""
        rd, wr, ca = C.c_double(), C.c_double(), C.c_double()
        name = C.create_string_buffer(64)
        check(self.L.pic1dp_hip_kernel_bytes(self._ctx, which, C.byref(rd), C.byref(wr), C.byref(ca), name, 64))
        return dict(read=rd.value, written=wr.value, carry=ca.value, name=name.value.decode())

    def kernel_stats(self, which=0):
        ms, n = C.c_double(), C.c_int64()
        check(self.L.pic1dp_hip_kernel_stats(self._ctx, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- the reference driver, src/pic1dp.F90:64-109 ---------------------------------------
    def run(self, on_output=None, max_steps=None, fused=True):
        """particle_load must have been called.  Runs the initial deposit+solve
        and the RK2 time loop until check_termination, calling
        on_output(self) at step 0 and whenever the reference would call
        output_all.  Returns the number of steps taken."""
        if on_output:
            self.set_output_fusion(True)     # output steps take their diagnostics inside the step
        self.interaction_collect_charge()
        self.field_solve_electric()
        if on_output:
            on_output(self)
        steps = 0
        term = self.check_termination()
        while term == 0 and (max_steps is None or steps < max_steps):
            if fused:          # every step up to the next output_all in one call (src/pic1dp.F90:78-109 evaluated ahead)
                n = max(1, self.steps_to_output())
                if max_steps is not None:
                    n = min(n, max_steps - steps)
                self.step(n)
                steps += n - 1
            else:
                for irk in (1, 2):
                    self.interaction_push_particle(irk)
                    self.particle_optimize(irk)            # src/pic1dp.F90:82
                    self.interaction_collect_charge()
                    self.field_solve_electric()
                self.set_time(self.itime + 1, self.time + self.inp.dt)
            steps += 1
            term = self.check_termination()
            if on_output and self.output_due(term):
                on_output(self)
        return steps


def device_count():
    return _lib.load().pic1dp_hip_device_count()


def tuning_build():
    """True when the loaded library is a -DPIC1DP_TUNING build (it then reads the measurement knobs of tools/README.md)"""
    return _lib.load().pic1dp_hip_tuning_build() == 1
