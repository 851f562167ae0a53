"""A stand-in for the pic1dp_amd package WITHOUT a GPU (test support, tests/test_host_logic.py): just enough of the engine's
Python surface for bench.py's CONTROL PLANE to run as N processes on the CPU -- the launcher, gloo rendez-vous, the bootstrap
of both charge sums, the choice by rehearsal, headline / second workload / exchange-beside / extra configurations, the
max-over-ranks timing and the one JSON line -- at the target's rank count, eight, which the GPU box (six processes on its
card at most) cannot host.  No physics: a step advances a counter, the "field energy" is a function of it."""
import importlib.util
import os
import time

_REAL = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "pic1dp_amd")


def _load(name):
    spec = importlib.util.spec_from_file_location("pic1dp_amd." + name, os.path.join(_REAL, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Pic1dpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class _Inp(dict):
    __getattr__ = dict.__getitem__


def make_input(**kw):
    d = dict(nx=192, nmode=1, dt=0.05, nparticle_max=6400000)
    d.update(kw)
    return _Inp(d)


def device_count():
    return int(os.environ.get("FAKE_DEVICES", "8"))


def tuning_build():
    return False


class Pic1dp:
    def __init__(self, inp, rank=0, nranks=1, device=0, npe=0):
        self.inp, self.rank, self.nranks, self.device = inp, rank, nranks, device
        self.steps = 0
        self.kind = 0
        self.xchg = 0
        self.timers = False
        self.np = inp["nparticle_max"] // nranks + (1 if inp["nparticle_max"] % nranks > rank else 0)
        self.itime, self.time = 0, 0.0

    # -- charge sums
    def comm_unique_id(self):
        return bytes((i * 7 + 3) % 256 for i in range(128))

    def comm_available(self):
        pass

    def comm_init(self, uid):
        assert uid == self.comm_unique_id()

    def xchg_create(self):
        return bytes((self.rank * 64 + i) % 256 for i in range(64))

    def xchg_connect(self, handles):
        assert len(handles) == 64 * self.nranks and handles[64 * self.rank:64 * self.rank + 64] == self.xchg_create()

    def xchg_info(self):
        return 1, self.xchg

    def xchg_time(self, reset=False):
        return 0.01 * self.xchg, self.xchg

    def set_allreduce(self, kind):
        self.kind = kind

    # -- the engine
    def set_launch(self, threads, bpc):
        pass

    def set_step_mode(self, mode):
        pass

    def particle_load(self):
        pass

    def interaction_collect_charge(self):
        pass

    def field_solve_electric(self):
        pass

    def interaction_push_particle(self, irk):
        if irk == 2:
            self.steps += 1

    def particle_optimize(self, irk):
        return 0

    def charge_local(self):
        import numpy as np
        return np.full(self.inp["nx"], float(self.rank + 1))

    def charge_reduced(self, c):
        assert abs(c[0] - self.nranks * (self.nranks + 1) / 2) < 1e-9      # the host-staged sum saw every rank

    def step(self, n):
        self.steps += n
        if self.kind == 2:
            self.xchg += n
        time.sleep(2e-5 * n * (1.0 + 0.3 * (self.kind == 1)))     # (the exchange "wins" the rehearsal)

    def sync(self):
        pass

    def field_energy(self):
        return 1e-8 * (1.0 + 0.01 * self.steps)

    def local_sizes(self, isp=0):
        return self.np, self.np

    def kernel_stats_enable(self, on=True):
        self.k0 = self.steps

    def kernel_stats(self, which=0):
        n = self.steps - getattr(self, "k0", 0) if which == 6 else 0
        return 0.05 * n, n

    def kernel_bytes(self, which=6):
        return dict(read=32.0, written=24.0, carry=0.0, name="k_step_one<sums> (fake)" if which == 6 else "-")

    def timers_enable(self, on=True):
        self.timers = bool(on)

    def timers_reset(self):
        self.t0 = self.steps

    def timer_ms(self, which):
        return 0.04 * (self.steps - getattr(self, "t0", 0)) if which in (4, 7) else 0.0

    def close(self):
        pass


parallel = _load("parallel")


class _Probe:
    @staticmethod
    def stream(nr, nw, n, reps, device=0):
        return 6000.0

    @staticmethod
    def layout(n, tile_log2, reps, device=0):
        return 0.0, 56.0 * n / 6.3e12 * 1e3


probe = _Probe()
import sys as _sys  # noqa: E402
_sys.modules[__name__ + ".probe"] = probe
_sys.modules[__name__ + ".parallel"] = parallel
