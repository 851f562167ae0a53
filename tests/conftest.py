"""pytest configuration: the `gpu` marker, import paths and shared helpers."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """the CPU oracle (test infrastructure); built on demand with gcc"""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


def _check_state_after_every_call(cls):
    """every public method of the engine is followed by pic1dp_hip_check_state (host side only: no device work, no
    synchronisation): the relations between the flags of the library's state machine are asserted at
    every API boundary of every test, not only where a wrong result would show"""
    import functools
    import inspect

    def wrap(fn):
        @functools.wraps(fn)
        def call(self, *a, **k):
            out = fn(self, *a, **k)
            if getattr(self, "_ctx", None):
                self.check_state(False)
            return out
        return call

    for name, fn in list(vars(cls).items()):
        if name.startswith("_") or name in ("check_state", "close") or not inspect.isfunction(fn):
            continue
        setattr(cls, name, wrap(fn))
    cls._state_checked = True


@pytest.fixture(scope="session")
def amd():
    """the product package; importing it loads libpic1dp_hip.so or fails loudly"""
    import pic1dp_amd
    if not getattr(pic1dp_amd.Pic1dp, "_state_checked", False):
        _check_state_after_every_call(pic1dp_amd.Pic1dp)
    return pic1dp_amd


@pytest.fixture
def tuning(amd):
    """tests that vary a MEASUREMENT knob (launch shapes, thresholds, schedules: tools/README.md) need a -DPIC1DP_TUNING
    build of the library -- the product build compiles none of them (round 6).  Build one with
    `PIC1DP_EXTRA_FLAGS=-DPIC1DP_TUNING PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/v_tuning.so python pic1dp_amd/build.py --force`
    and run the suite with PIC1DP_LIB pointing at it; with the product library these tests are skipped."""
    if not amd.tuning_build():
        pytest.skip("needs a -DPIC1DP_TUNING build of the library (PIC1DP_LIB=.../v_tuning.so): the product build has no such knob")
    return True


@pytest.fixture(scope="session")
def probe():
    """libpic1dp_probe.so (measurement / test support, include/pic1dp_probe.h): array evaluations of the
    device functions the marker kernels call, and the streaming probes"""
    from pic1dp_amd import probe as mod
    mod.load()
    return mod


# input keyword sets shared by CPU and GPU tests: (id, kwargs)
DIST_CASES = [
    ("bump_on_tail", dict()),
    ("maxwellian", dict(iptcldist=0, species_density=[1.0], species_v0=[0.0])),
    ("two_stream1", dict(iptcldist=1, species_density=[1.0])),
    ("two_stream2", dict(iptcldist=2, species_density=[1.0], species_v0=[3.0])),
    ("bump_nonpow2", dict(iptcldist=3, species_temperature=[1.3], species_temperature2=[0.7],
                          species_mass=[1.1], species_density=[0.85], species_v0=[4.5])),
    ("two_stream2_nonpow2", dict(iptcldist=2, species_density=[1.0], species_v0=[3.0],
                                 species_temperature=[0.9], species_mass=[1.2])),
    ("maxwellian_nonpow2", dict(iptcldist=0, species_density=[1.0], species_v0=[0.3],
                                species_temperature=[1.7], species_mass=[0.9])),
]
