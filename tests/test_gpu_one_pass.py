"""One pass over the markers per time step (kernels_step.hip k_step_one): the second sub-step's
kernel also deposits what the NEXT step's first sub-step would deposit, as coefficients of the
kept field modes (the half push is linear in the field it sees, src/pic1dp_interaction.F90:261,
268-329; the field is its kept modes times fixed tables, src/pic1dp_field.F90:251-257), so from
the second step on no first-sub-step pass runs.  Against the two-pass engine (PIC1DP_PREDICT=0)
and against the oracle."""
import numpy as np
import pytest

from conftest import DIST_CASES
from util import relerr

pytestmark = pytest.mark.gpu

N = 200_001


def engine(amd, monkeypatch, predict, kind=None, npe=1, **kw):
    """kind 1: the prediction as tiles (k_step_one) wherever they fit; kind 2: as six sums (k_step_one<PRIV> in
    thread-private LDS slots where E0, Eh and the tables fit the LDS, else k_step_sums) -- the library's own choice
    is the sums from nx = 512 up with one kept mode, the tiles below and with two kept modes"""
    monkeypatch.setenv("PIC1DP_PREDICT", "1" if predict else "0")
    if kind and predict:         # 1: the tiles wherever they fit; 2: the six sums (k_step_one<PRIV> / k_step_sums);
        monkeypatch.setenv("PIC1DP_PRED_KIND", str(kind))      # 3: the six sums in registers (k_step_sums) at any grid
    else:
        monkeypatch.delenv("PIC1DP_PRED_KIND", raising=False)
    e = amd.Pic1dp(amd.make_input(**kw), npe=npe)
    e.particle_load()
    e.interaction_collect_charge()
    e.field_solve_electric()
    return e


MODES = [("df_nonlinear", dict()), ("df_linear", dict(linear=1)), ("full_f", dict(deltaf=0))]


KINDS = [1, 2]


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
@pytest.mark.parametrize("name,kw", DIST_CASES, ids=[c[0] for c in DIST_CASES])
@pytest.mark.parametrize("mname,mkw", MODES, ids=[m[0] for m in MODES])
def test_one_pass_equals_two_passes(amd, monkeypatch, name, kw, mname, mkw, kind):
    if mkw.get("deltaf") == 0 and name != "maxwellian":
        pytest.skip("full-f evaluates no f0 derivative: one distribution covers it")
    kw = dict(kw, nparticle_max=N, nx=96, **mkw)
    a = engine(amd, monkeypatch, True, kind, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    a.kernel_stats_enable(True)
    b.kernel_stats_enable(True)
    nsteps = 12
    a.step(nsteps)
    b.step(nsteps)
    ea, eb = a.energy_history(), b.energy_history()
    assert np.max(np.abs(ea / eb - 1.0)) < 1e-11
    assert relerr(a.get_field_half(), b.get_field_half()) < 1e-11      # the predicted half-step field
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k
    # one first-sub-step pass (the very first step), then one kernel per step
    assert a.kernel_stats(3)[1] == 1 and a.kernel_stats(6)[1] == nsteps and a.kernel_stats(4)[1] == 0
    assert b.kernel_stats(3)[1] == nsteps and b.kernel_stats(4)[1] == nsteps and b.kernel_stats(6)[1] == 0


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_one_pass_against_oracle(oracle_mod, amd, monkeypatch, kind):
    """field energy at every step of a 300-step run within 1e-10 of the CPU arithmetic"""
    kw = dict(nparticle_max=N, nx=64)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    eng = engine(amd, monkeypatch, True, kind, **kw)
    eng.kernel_stats_enable(True)
    eo = []
    for _ in range(300):
        sim.step(1)
        eo.append(sim.field_energy())
    eng.step(100)
    eng.step(1)
    eng.step(199)
    assert np.max(np.abs(eng.energy_history() / np.array(eo) - 1.0)) < 1e-10
    assert eng.kernel_stats(3)[1] == 1 and eng.kernel_stats(6)[1] == 300


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
@pytest.mark.parametrize("name,kw", DIST_CASES, ids=[c[0] for c in DIST_CASES])
@pytest.mark.parametrize("mname,mkw", MODES, ids=[m[0] for m in MODES])
def test_one_pass_against_oracle_every_distribution(oracle_mod, amd, monkeypatch, name, kw, mname, mkw, kind):
    """the path the bench times against the CPU arithmetic DIRECTLY, for every distribution and mode (not through
    the two-pass engine): field energy at every one of 80 steps within 1e-10, markers to 1e-9 at the end"""
    if mkw.get("deltaf") == 0 and name != "maxwellian":
        pytest.skip("full-f evaluates no f0 derivative: one distribution covers it")
    kw = dict(kw, nparticle_max=N, nx=64, **mkw)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    eng = engine(amd, monkeypatch, True, kind, **kw)
    assert eng.predict_kind() == kind
    eng.kernel_stats_enable(True)
    nsteps = 80
    eo = []
    for _ in range(nsteps):
        sim.step(1)
        eo.append(sim.field_energy())
    eng.step(nsteps)
    assert np.max(np.abs(eng.energy_history() / np.array(eo) - 1.0)) < 1e-10
    assert eng.kernel_stats(3)[1] == 1 and eng.kernel_stats(6)[1] == nsteps
    g = eng.particles_download()
    assert np.max(np.abs(g["x"] - sim.gather("x"))) < 1e-9
    assert np.max(np.abs(g["v"] - sim.gather("v"))) < 1e-9
    if kw.get("deltaf", 1):
        assert np.max(np.abs(g["w"] - sim.gather("w"))) < 1e-9 * max(1.0, np.max(np.abs(sim.gather("w"))))


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_one_pass_through_the_call_sites(amd, monkeypatch, kind):
    """the reference's three call sites: collect_charge after push(1) combines the prediction
    (no marker pass), its chargeden is the eager deposit's to rounding -- with the six sums: the kept mode's
    content of it, all solve_field looks at --; looking at the markers in between still gives the eager state"""
    kw = dict(nparticle_max=N, nx=96)
    a = engine(amd, monkeypatch, True, kind, **kw)
    monkeypatch.setenv("PIC1DP_LAZY_CALLS", "0")
    b = engine(amd, monkeypatch, False, **kw)
    monkeypatch.delenv("PIC1DP_LAZY_CALLS")
    b.set_electric(a.get_field()["electric"])
    a.kernel_stats_enable(True)
    ang = 2.0 * np.pi * a.inp.modes[0] * np.arange(kw["nx"]) / kw["nx"]
    basis = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    for it in range(4):
        for irk in (1, 2):
            for e in (a, b):
                e.interaction_push_particle(irk)
                e.interaction_collect_charge()
                e.field_solve_electric()
            # with the six sums field_chargeden holds, between the sub-steps, the kept mode's content of the
            # half-step charge density -- all solve_field looks at; ASKING for it would rebuild the whole vector
            # by running the half push after all (test_chargeden_between_the_sub_steps_is_rebuilt_on_inspection),
            # so this test, which counts kernels, looks at the field only there
            filtered = kind == 2 and irk == 1 and it > 0
            fa, fb = a.get_field(chargeden=not filtered), b.get_field()
            if not filtered:
                assert relerr(fa["chargeden"], fb["chargeden"]) < 1e-11, (it, irk)
            assert relerr(fa["electric"], fb["electric"]) < 1e-11, (it, irk)
            b.set_electric(fa["electric"])
            a.set_electric(fa["electric"]) if False else None
        ga, gb = a.particles_download(), b.particles_download()
        for k in "xvw":
            assert np.array_equal(ga[k], gb[k]), (k, it)
    # step 1: half + one-pass kernel; after every look at the markers the prediction is still valid
    # (downloads do not change them), so steps 2-4 need no first-sub-step pass
    assert a.kernel_stats(3)[1] == 1 and a.kernel_stats(6)[1] == 4


def test_chargeden_between_the_sub_steps_is_rebuilt_on_inspection(amd, monkeypatch):
    """k_step_sums (grids beyond nx ~ 2400; forced here): the collect_charge after push(1) is served from six
    sums and leaves in field_chargeden the kept mode's content of the half-step charge density.  A host that
    LOOKS at field_chargeden there (a custom half-step diagnostic; the reference driver never does) gets the
    reference's full vector: the library pushes the half-step state into memory after all and deposits it --
    equal to the eager engine's deposit to rounding, not a filtered one --, and the run goes on correctly."""
    kw = dict(nparticle_max=N, nx=96, init_nmode=2, init_mode=[1, 3], init_mode_cos=[0.0, 0.0], init_mode_sin=[1e-3, 5e-4])
    a = engine(amd, monkeypatch, True, 2, **kw)
    monkeypatch.setenv("PIC1DP_LAZY_CALLS", "0")
    b = engine(amd, monkeypatch, False, **kw)
    monkeypatch.delenv("PIC1DP_LAZY_CALLS")
    a.kernel_stats_enable(True)
    for it in range(4):
        for irk in (1, 2):
            for e in (a, b):
                e.interaction_push_particle(irk)
                e.interaction_collect_charge()
                e.field_solve_electric()
            look = it == 2 and irk == 1          # once, in a step whose half-step charge was predicted
            fa, fb = a.get_field(chargeden=look), b.get_field()
            if look:
                cd = fb["chargeden"]
                # mode 3 was loaded too and is NOT kept by the field solve: it is in the full vector only
                spec = np.abs(np.fft.rfft(cd))
                assert spec[3] > 0.05 * spec[1]
                assert relerr(fa["chargeden"], cd) < 1e-11
            assert relerr(fa["electric"], fb["electric"]) < 1e-10, (it, irk)
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-10 * max(1.0, np.max(np.abs(gb[k]))), k
    # steps 0 (first step) and 2 (the inspection) took the two-pass kernels, 1 and 3 the one-pass kernel... the step
    # AFTER the inspection has no prediction to start from either
    assert a.kernel_stats(6)[1] >= 1 and a.kernel_stats(3)[1] >= 2


def test_chargeden_asked_for_between_push2_and_collect_charge(amd, monkeypatch):
    """push(2) has been NOTED (nothing launched) when the host asks for field_chargeden -- still the kept
    mode's content of the half-step charge density.  The library rebuilds the reference's vector (half push into
    memory, deposit) and then runs the noted push(2) at once, so that memory is what eager calls would hold; the flag
    that says 'kept mode only' is cleared because the vector WAS rebuilt, and pic1dp_hip_chargeden_state reports it."""
    kw = dict(nparticle_max=N, nx=96, init_nmode=2, init_mode=[1, 3], init_mode_cos=[0.0, 0.0], init_mode_sin=[1e-3, 5e-4])
    a = engine(amd, monkeypatch, True, 2, **kw)
    monkeypatch.setenv("PIC1DP_LAZY_CALLS", "0")
    b = engine(amd, monkeypatch, False, **kw)
    monkeypatch.delenv("PIC1DP_LAZY_CALLS")
    for it in range(4):
        for irk in (1, 2):
            for e in (a, b):
                e.interaction_push_particle(irk)
                if irk == 2 and it == 2:           # between push(2) and its collect_charge
                    if e is a:
                        assert a.chargeden_kept_mode_only()         # served from the six sums after push(1)
                    cd = e.get_field()["chargeden"]
                    if e is a:
                        assert not a.chargeden_kept_mode_only()     # rebuilt
                        cd_a = cd
                    else:
                        spec = np.abs(np.fft.rfft(cd))
                        assert spec[3] > 0.05 * spec[1]             # mode 3 is in the full vector only
                        assert relerr(cd_a, cd) < 1e-11
                e.interaction_collect_charge()
                e.field_solve_electric()
            assert relerr(a.get_field(chargeden=False)["electric"], b.get_field()["electric"]) < 1e-10, (it, irk)
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-10 * max(1.0, np.max(np.abs(gb[k]))), k


@pytest.mark.parametrize("kw", [dict(nx=96), dict(nx=1000), dict(nx=2050, deltaf=0),
                                dict(nx=64, nspecies=2, iptcldist=0, species_charge=[-1.0, 1.0], species_mass=[1.0, 25.0],
                                     species_temperature=[1.0, 0.5], species_temperature2=[1.0, 1.0],
                                     species_density=[1.0, 1.0], species_v0=[0.0, 0.0], lx=4 * np.pi)],
                         ids=["nx96", "nx1000", "nx2050_full_f", "two_species"])
@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
@pytest.mark.parametrize("npe", [1, 4, 9], ids=lambda v: "npe%d" % v)
def test_pair_solve_new_state_field_bit_identical_to_the_oracle(oracle_mod, amd, monkeypatch, kw, kind, npe):
    """k_field_solve_pair1 / k_field_solve_pair_sums1 (one kept mode: everything that does not wait for the serial sums in
    front of them, both fields of a one-pass step in one launch): the field of the new state from ITS charge density against
    the oracle's solve bit for bit, in the order of an npe-rank reference run; the predicted half-step field and the run
    against the two-pass engine to rounding.  (Until round 6 the comparison was with the unoptimised kernels of the same
    launch, PIC1DP_PAIR_PLAIN: retired with them.)"""
    kw = dict(kw, nparticle_max=N)
    if kind == 2 and kw["nx"] == 1000:
        kw["nx"] = 4096               # the grid the sums are for: 1024 threads, four cells each
    a = engine(amd, monkeypatch, False, None, npe, **kw)
    b = engine(amd, monkeypatch, True, kind, npe, **kw)
    assert a.predict_kind() == 0 and b.predict_kind() == kind
    field = oracle_mod.Field(oracle_mod.make_input(**kw))
    for it in range(4):
        a.step(1)
        b.step(1)
        fa, fb = a.get_field(), b.get_field()
        assert relerr(fa["electric"], fb["electric"]) < 1e-11, it
        for f in (fa, fb):
            E, re, im = field.solve(f["chargeden"], npe)   # in the order of an npe-rank reference run
            assert np.array_equal(f["electric"], E) and np.array_equal(f["mode_re"], re) and np.array_equal(f["mode_im"], im)
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_one_pass_with_the_hosts_own_reduction(amd, monkeypatch, kind):
    """a host that keeps its own MPI_Allreduce (charge_local / charge_reduced): after push(1) what it
    reduces is the prediction -- a charge vector (tiles) or the six sums at the head of one (sums)"""
    kw = dict(nparticle_max=N, nx=96)
    a = engine(amd, monkeypatch, True, kind, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    a.kernel_stats_enable(True)
    for it in range(5):
        for irk in (1, 2):
            a.interaction_push_particle(irk)
            c2 = a.charge_local()
            if kind == 2 and irk == 1 and it > 0:
                assert np.all(c2[6:] == 0.0) and np.any(c2[:6] != 0.0)
            a.charge_reduced(c2)
            a.field_solve_electric()
        b.step(1)
        assert abs(a.field_energy() / b.field_energy() - 1.0) < 1e-11, it
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k
    assert a.kernel_stats(3)[1] == 1 and a.kernel_stats(6)[1] == 5


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_prediction_is_dropped_when_it_no_longer_applies(amd, monkeypatch, kind):
    """a field set from outside, re-uploaded markers, another solver: the first-sub-step pass runs again"""
    kw = dict(nparticle_max=N, nx=96)
    a = engine(amd, monkeypatch, True, kind, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    a.kernel_stats_enable(True)
    for e in (a, b):
        e.step(3)
    E = b.get_field()["electric"] * 1.5            # not what the kept modes say
    for e in (a, b):
        e.set_electric(E)
        e.step(2)
    g = b.particles_download()
    for e in (a, b):
        e.particles_upload(g["x"], g["v"], g["p"], g["w"])
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.step(2)
    assert abs(a.field_energy() / b.field_energy() - 1.0) < 1e-11
    # first-sub-step passes: step 1, the step after set_electric, the step after the upload
    assert a.kernel_stats(3)[1] == 3 and a.kernel_stats(6)[1] == 7
    a.set_field_solver(1)                          # finite differences: E is not its kept modes
    a.step(2)
    assert a.kernel_stats(3)[1] == 5 and a.kernel_stats(4)[1] == 2


@pytest.mark.parametrize("kw,predicted", [
    (dict(nmode=2, modes=[1, 3], init_nmode=2, init_mode=[1, 3], init_mode_cos=[0.0, 2e-6], init_mode_sin=[1e-5, 0.0]), True),
    (dict(nmode=3, modes=[1, 2, 5], init_nmode=3, init_mode=[1, 2, 5], init_mode_cos=[0.0, 2e-6, 1e-6],
          init_mode_sin=[1e-5, 0.0, 3e-6]), True),              # three kept modes: tiles too (fixed-point sums, round 6)
    (dict(nmode=4, modes=[1, 2, 3, 7], init_nmode=2, init_mode=[1, 7], init_mode_cos=[0.0, 2e-6], init_mode_sin=[1e-5, 0.0]), False),
    (dict(nmode=5, modes=[1, 2, 3, 4, 5]), False),              # four and more kept modes: two passes
    (dict(nspecies=2, iptcldist=0, species_charge=[-1.0, 1.0], species_mass=[1.0, 25.0], species_temperature=[1.0, 0.5],
          species_temperature2=[1.0, 1.0], species_density=[1.0, 1.0], species_v0=[0.0, 0.0], lx=4 * np.pi), True),
    (dict(nx=4096), True),                                      # eight tiles of 32 KiB do not fit the LDS: six sums
    (dict(nx=4096, nmode=2, modes=[1, 3]), False),              # ... which are of ONE kept mode
    (dict(nx=4096, deltaf=0), True),
    (dict(nx=4096, nspecies=2, iptcldist=0, species_charge=[-1.0, 1.0], species_mass=[1.0, 25.0], species_temperature=[1.0, 0.5],
          species_temperature2=[1.0, 1.0], species_density=[1.0, 1.0], species_v0=[0.0, 0.0], lx=4 * np.pi), True),
    (dict(nx=33), True),                                        # odd grid: guard cell and tile padding
    (dict(nparticle_max=N + 1, species_nparticle_init=[N - 7]), True)],
    ids=["two_modes", "three_modes", "four_modes", "five_modes_fall_back", "two_species", "nx4096_sums", "nx4096_two_modes_fall_back", "nx4096_full_f", "nx4096_two_species", "odd_nx", "even_count_tail_slots"])
@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_one_pass_modes_species_fallbacks(amd, monkeypatch, kw, predicted, kind):
    kw = dict(dict(nparticle_max=N, nx=96), **kw)
    if kind == 2 and kw.get("nmode", 1) in (2, 3, 4) and kw["nx"] < 4096:
        pytest.skip("two to four kept modes: the tiles' case (or two passes)")
    a = engine(amd, monkeypatch, True, kind, **kw)
    assert a.predict_kind() == (0 if not predicted else 2 if (kind == 2 or kw["nx"] == 4096) else 1)
    b = engine(amd, monkeypatch, False, **kw)
    a.kernel_stats_enable(True)
    a.step(8)
    b.step(8)
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    assert relerr(a.get_field()["electric"], b.get_field()["electric"]) < 1e-11
    ns = kw.get("nspecies", 1)
    if predicted:
        assert a.kernel_stats(3)[1] == ns and a.kernel_stats(6)[1] == 8 * ns and a.kernel_stats(4)[1] == 0
    else:
        assert a.kernel_stats(3)[1] == 8 * ns and a.kernel_stats(4)[1] == 8 * ns and a.kernel_stats(6)[1] == 0


@pytest.mark.parametrize("kw,kind", [
    (dict(nx=2542), 1), (dict(nx=2543), 2),                                   # the last grid the tiles hold, the first for the sums
    (dict(nx=1694, nmode=2, modes=[1, 3]), 1), (dict(nx=1695, nmode=2, modes=[1, 3]), 0),
    (dict(nx=1270, nmode=3, modes=[1, 2, 5]), 1), (dict(nx=1271, nmode=3, modes=[1, 2, 5]), 0),
    (dict(nx=512, nmode=4, modes=[1, 2, 3, 5]), 0),                           # four and more kept modes: two passes
    (dict(nx=5063), 2), (dict(nx=5064), 0)],                                  # the last grid for the sums, then two passes
    ids=["tiles_last", "sums_first", "two_modes_last", "two_modes_beyond", "three_modes_last", "three_modes_beyond",
         "four_modes_two_passes", "sums_last", "beyond"])
def test_one_pass_at_the_lds_limits(amd, monkeypatch, kw, kind):
    """the grids at which the one-pass kernels' LDS tiles just fit and just do not (kernels.hpp step_one_lds_bytes,
    step_sums_lds_bytes against PARTICLE_LDS_CAP): the choice, and the run against the two-pass engine"""
    kw = dict(kw, nparticle_max=N)
    a = engine(amd, monkeypatch, True, None, **kw)
    assert a.predict_kind() == kind
    b = engine(amd, monkeypatch, False, **kw)
    a.step(5)
    b.step(5)
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    assert relerr(a.get_field()["electric"], b.get_field()["electric"]) < 1e-11
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k


@pytest.mark.parametrize("nx,force,kind,kernel", [
    (8, None, 2, "k_step_one<sums>"),         # the library's own choice for one kept mode from nx = 8 up (round 4) ...
    (7, None, 1, "k_step_one"),               # ... the tiles below (the sums travel in the head of an nx-vector)
    (192, None, 2, "k_step_one<sums>"),       # the reference's default grid
    (192, 1, 1, "k_step_one"),                # the tiles when insisted on
    (1096, None, 2, "k_step_one<sums>"),      # the last grid whose tiles and slots fit a CU twice
    (1097, None, 1, "k_step_one"),            # one cell more: the tiles again by default,
    (1097, 2, 2, "k_step_sums")],             # the register sums when the six sums are insisted on
    ids=["private_first", "below", "default_grid", "default_grid_forced_tiles", "private_last", "beyond_default",
         "beyond_forced_sums"])
def test_private_sums_at_their_limits(amd, monkeypatch, nx, force, kind, kernel):
    """k_step_one<PRIV> (six sums in thread-private LDS slots) between nx = 8 and the last grid at which E0, Eh, the
    tables, the rho tile and the slots of TWO workgroups fit a CU's LDS (kernels.hpp step_one_private_lds_bytes): the
    choice, the kernel that ran (as the library names it), and the run against the two-pass engine"""
    kw = dict(nx=nx, nparticle_max=N)
    a = engine(amd, monkeypatch, True, force, **kw)
    assert a.predict_kind() == kind
    b = engine(amd, monkeypatch, False, **kw)
    a.kernel_stats_enable(True)
    a.step(5)
    b.step(5)
    name = a.kernel_bytes(6)["name"]
    assert name.startswith(kernel) and (kernel != "k_step_one" or not name.startswith("k_step_one<sums>")), name
    assert a.kernel_stats(6)[1] == 5
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    assert relerr(a.get_field()["electric"], b.get_field()["electric"]) < 1e-11
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_prediction_of_markers_that_cross_several_boxes(amd, monkeypatch, kind):
    """markers fast enough to cross more than a box length in half a step: the prediction wraps the CELL of
    x + dt/2 v as an integer (its general path), the push wraps the position"""
    kw = dict(nparticle_max=N, nx=96, iptcldist=0, species_density=[1.0], species_v0=[0.0], lx=4 * np.pi)
    a = engine(amd, monkeypatch, True, kind, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    g = b.particles_download()
    rng = np.random.default_rng(5)
    fast = rng.random(g["v"].size) < 0.05
    g["v"][fast] = rng.choice([-1.0, 1.0], fast.sum()) * rng.uniform(600.0, 3000.0, fast.sum())
    for e in (a, b):
        e.particles_upload(g["x"], g["v"], g["p"], g["w"])
        e.interaction_collect_charge()
        e.field_solve_electric()
    a.kernel_stats_enable(True)
    a.step(5)
    b.step(5)
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    assert relerr(a.get_field_half(), b.get_field_half()) < 1e-11
    assert a.kernel_stats(6)[1] == 5



# ---------------------------------------------------------------------------
# One launch per time step: the marker kernel's prologue solves the field of the previous step (kernels_step.hip
# FUSED, kernels.hpp FusedSolve) with the device functions of the field kernels.  PIC1DP_FUSE_SOLVE=0 keeps the solve
# in a launch of its own.
# ---------------------------------------------------------------------------
def fused_pair(amd, monkeypatch, npe=1, env=None, **kw):
    """two engines on the same input: the solve inside the marker launches, and in launches of its own"""
    env = dict(env or {})
    kind = int(env.pop("PIC1DP_PRED_KIND", "2"))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "2")      # (1, the default, fuses where the serial sums are short enough)
    a = engine(amd, monkeypatch, True, kind, npe=npe, **kw)
    monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "0")
    b = engine(amd, monkeypatch, True, kind, npe=npe, **kw)
    monkeypatch.delenv("PIC1DP_FUSE_SOLVE")
    return a, b


FUSED_CASES = [
    ("bump_private_slots", dict(), 1, {}),
    ("bump_register_sums", dict(), 1, {"PIC1DP_PRED_KIND": "3"}),
    ("bump_four_rank_order", dict(), 4, {}),
    ("bump_reference_order_dlnf0_carry", dict(), 1, {"PIC1DP_DLNF0": "ref"}),
    ("two_stream2_nonpow2", dict(iptcldist=2, species_density=[1.0], species_v0=[3.0], species_temperature=[0.9],
                                 species_mass=[1.2]), 1, {}),
    ("maxwellian_linear", dict(iptcldist=0, species_density=[1.0], species_v0=[0.0], linear=1), 1, {}),
    ("maxwellian_full_f", dict(iptcldist=0, species_density=[1.0], species_v0=[0.0], deltaf=0), 2, {}),
    ("two_species", dict(nspecies=2, species_charge=[-1.0, 1.0], species_mass=[1.0, 4.0], species_temperature=[1.0, 1.0],
                         species_temperature2=[1.0, 1.0], species_density=[0.9, 0.9], species_v0=[5.0, 5.0],
                         species_nparticle_init=[96, 96]), 1, {}),
]


@pytest.mark.parametrize("name,kw,npe,env", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_fused_solve_bit_identical_to_the_field_kernel(amd, monkeypatch, name, kw, npe, env):
    """With at most one wave of markers per rank block the charge sums have ONE order (a wave's lanes, then program
    order), so both engines see the same bits in their accumulators at every step -- and must then agree bit for bit
    in everything the solve writes: E, charge density, the kept mode, the predicted half-step field, the field energy
    of every step, and the markers that were pushed with those fields."""
    kw = dict(kw, nparticle_max=96, nx=32)
    a, b = fused_pair(amd, monkeypatch, npe=npe, env=env, **kw)
    nsteps = 25
    a.step(nsteps)
    b.step(nsteps)
    assert a.kernel_stats(7)[1] == nsteps - 1 and b.kernel_stats(7)[1] == 0      # (the first step of a run takes two passes)
    fa, fb = a.get_field(), b.get_field()
    for k in ("electric", "chargeden", "mode_re", "mode_im"):
        assert np.array_equal(fa[k], fb[k]), k
    assert np.array_equal(a.get_field_half(), b.get_field_half())
    assert np.array_equal(a.energy_history(), b.energy_history())
    for isp in range(kw.get("nspecies", 1)):
        ga, gb = a.particles_download(isp), b.particles_download(isp)
        for k in "xvw":
            assert np.array_equal(ga[k], gb[k]), (isp, k)
    # and nothing is left behind in the accumulators: the next deposit starts from zero in both
    a.interaction_collect_charge()
    b.interaction_collect_charge()
    assert np.array_equal(a.get_field()["chargeden"], b.get_field()["chargeden"])


@pytest.mark.parametrize("threads,bpc,osub", [(64, 0, None), (128, 0, None), (256, 0, None), (1024, 0, None), (512, 2, None),
                                              (0, 0, "1"), (0, 0, "3")],
                         ids=["t64", "t128", "t256", "t1024", "t512x2", "osub1", "osub3"])
def test_fused_solve_over_launch_shapes(amd, monkeypatch, request, threads, bpc, osub):
    """pic1dp_hip_set_launch / PIC1DP_OSUB (a tuning build's knob) together with the solve in the marker launch's prologue.  A
    launch shape asked for by hand selects the register-sum kernel (k_step_sums<FUSED>) with that workgroup size; the
    prologue zeroes the rotated accumulator sets with strided stores whatever the size, and a workgroup of one wave --
    which has no first AND last wave for the prologue's two jobs -- must not fuse at all.  One wave of markers per
    block: one order of the charge sums, so fused and unfused agree bit for bit over enough steps for every
    accumulator set to have been read, deposited into and zeroed several times."""
    kw = dict(nparticle_max=96, nx=32)
    if osub:
        request.getfixturevalue("tuning")
        monkeypatch.setenv("PIC1DP_OSUB", osub)
    monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "2")
    a = engine(amd, monkeypatch, True, 2, **kw)
    monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "0")
    b = engine(amd, monkeypatch, True, 2, **kw)
    for e in (a, b):
        if threads or bpc:
            e.set_launch(threads, bpc)
    nsteps = 40
    a.step(nsteps)
    b.step(nsteps)
    fused = a.kernel_stats(7)[1]
    if threads == 64:
        assert fused == 0                      # a one-wave workgroup never carries the solve
    else:
        assert fused == nsteps - 1
    assert b.kernel_stats(7)[1] == 0
    assert np.array_equal(a.energy_history(), b.energy_history())
    fa, fb = a.get_field(), b.get_field()
    for k in ("electric", "chargeden", "mode_re", "mode_im"):
        assert np.array_equal(fa[k], fb[k]), k
    assert np.array_equal(a.get_field_half(), b.get_field_half())
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.array_equal(ga[k], gb[k]), k
    a.interaction_collect_charge()                # nothing stale in any accumulator set
    b.interaction_collect_charge()
    assert np.array_equal(a.get_field()["chargeden"], b.get_field()["chargeden"])


def test_fused_solve_many_markers_small_workgroups(oracle_mod, amd, monkeypatch):
    """the 64-thread case of ADVICE r04 at a marker count where every copy of the six sums is deposited into
    (more workgroups than copies): the run against the oracle, 1e-10 at every step"""
    kw = dict(nparticle_max=N, nx=96)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    eo = []
    for _ in range(40):
        sim.step(1)
        eo.append(sim.field_energy())
    for threads in (64, 128):
        monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "2")
        a = engine(amd, monkeypatch, True, 2, **kw)
        a.set_launch(threads, 0)
        a.step(40)
        assert np.max(np.abs(a.energy_history() / np.array(eo) - 1.0)) < 1e-10, threads


@pytest.mark.parametrize("nx,env", [(1024, {}), (96, {}), (2500, {}), (1024, {"PIC1DP_PRED_KIND": "3"})],
                         ids=["nx1024_private_slots", "nx96_private_slots", "nx2500_register_sums", "nx1024_register_sums"])
def test_fused_solve_many_markers(oracle_mod, amd, monkeypatch, nx, env):
    """at a realistic marker count (the order of the charge atomics differs between runs): every step's field energy
    against the oracle, 1e-10, and against the engine that solves in launches of its own, 1e-11; calls of one step
    never fuse (their field has to be in memory when they return), calls of several fuse all but the last"""
    kw = dict(nparticle_max=N, nx=nx)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    a, b = fused_pair(amd, monkeypatch, env=env, **kw)
    eo = []
    for _ in range(60):
        sim.step(1)
        eo.append(sim.field_energy())
    for e in (a, b):
        e.step(1)
        e.step(1)
        e.step(30)
        e.step(1)
        e.step(27)
    assert a.kernel_stats(7)[1] == 29 + 26 and b.kernel_stats(7)[1] == 0
    ea, eb = a.energy_history(), b.energy_history()
    assert len(ea) == 60 and np.max(np.abs(ea / np.array(eo) - 1.0)) < 1e-10
    assert np.max(np.abs(ea / eb - 1.0)) < 1e-11
    assert relerr(a.get_field_half(), b.get_field_half()) < 1e-11
    fa, fb = a.get_field(), b.get_field()
    for k in ("electric", "chargeden", "mode_re", "mode_im"):
        assert relerr(fa[k], fb[k]) < 1e-11, k
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k


def test_fused_solve_with_output_steps(oracle_mod, amd, monkeypatch):
    """the reference's driver loop with output fusion: the step before an output_all takes its diagnostics inside
    k_step_full<DIAG> and needs its fields in memory -- the step before it must not leave its solve pending"""
    kw = dict(nparticle_max=N, nx=1024, output_interval=0.35)
    a, b = fused_pair(amd, monkeypatch, **kw)
    outs = {id(a): [], id(b): []}
    for e in (a, b):
        e.set_output_fusion(True)
        t = 0
        for chunk in (7, 7, 1, 7, 14, 3):
            e.step(chunk)
            t += chunk
            outs[id(e)].append((e.output_scalars(), e.ptcldist()))
    assert a.kernel_stats(7)[1] > 20 and b.kernel_stats(7)[1] == 0
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    for (sa, da), (sb, db) in zip(outs[id(a)], outs[id(b)]):
        assert relerr(sa, sb) < 1e-11
        for k in da:
            assert relerr(da[k], db[k]) < 1e-9, k


@pytest.mark.parametrize("modes", [[1, 2, 3], [1, 2, 3, 5]], ids=["three_modes", "four_modes"])
@pytest.mark.parametrize("mname,mkw", MODES, ids=[m[0] for m in MODES])
def test_three_and_four_kept_modes_against_oracle(oracle_mod, amd, monkeypatch, modes, mname, mkw):
    """the reference allows any input_nmode (src/pic1dp_input.F90:75-80).  Three kept modes are ONE pass
    per step (prediction tiles R0, RA_m, RB_m as fixed-point sums: round 6, 1.20 against the two passes' 1.40 ms at 1e8
    markers, profiles/r06/experiments/ab_nm3.log), four the two passes (k_step_half + k_step_full) -- against the oracle
    directly: field energy at every one of 80 steps within 1e-10, the field and the markers at the end"""
    nm = len(modes)
    kw = dict(nparticle_max=N, nx=128, nmode=nm, modes=modes, init_nmode=nm, init_mode=modes,
              init_mode_cos=[0.0, 2e-6, 1e-6, 5e-7][:nm], init_mode_sin=[1e-5, 3e-6, 0.0, 2e-6][:nm], **mkw)
    if mkw.get("deltaf") == 0:
        kw.update(iptcldist=0, species_density=[1.0], species_v0=[0.0])
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    eng = engine(amd, monkeypatch, True, None, **kw)
    assert eng.predict_kind() == (1 if nm == 3 else 0)           # the library's own choice
    eng.kernel_stats_enable(True)
    nsteps = 80
    eo = []
    for _ in range(nsteps):
        sim.step(1)
        eo.append(sim.field_energy())
    eng.step(nsteps)
    assert np.max(np.abs(eng.energy_history() / np.array(eo) - 1.0)) < 1e-10
    if nm == 3:
        assert eng.kernel_stats(3)[1] == 1 and eng.kernel_stats(6)[1] == nsteps and eng.kernel_stats(4)[1] == 0
    else:
        assert eng.kernel_stats(3)[1] == nsteps and eng.kernel_stats(4)[1] == nsteps and eng.kernel_stats(6)[1] == 0
    f = eng.get_field()
    so = sim.get_field()
    assert relerr(f["electric"], so[0]) < 1e-10
    g = eng.particles_download()
    assert np.max(np.abs(g["x"] - sim.gather("x"))) < 1e-9 and np.max(np.abs(g["v"] - sim.gather("v"))) < 1e-9
    if kw.get("deltaf", 1):
        assert np.max(np.abs(g["w"] - sim.gather("w"))) < 1e-9 * max(1.0, np.max(np.abs(sim.gather("w"))))


@pytest.mark.parametrize("kind", KINDS, ids=["tiles", "sums"])
def test_call_sites_two_launches_per_step(oracle_mod, amd, monkeypatch, kind):
    """through the three call sites (src/pic1dp.F90:80-89), one rank, the solve_field behind the second
    collect_charge solves BOTH fields of the step in one launch -- as pic1dp_hip_step does -- and the next step's push(1),
    collect_charge, solve_field launch nothing: marker kernel + one field launch per time step (three launches and a copy
    before; PIC1DP_CALL_PAIR=0).  With one wave of markers per block (one order of the charge sums) the three ways of
    stepping -- call sites with the pair, call sites without, pic1dp_hip_step -- agree bit for bit; at a realistic count
    every step's field energy is the oracle's to 1e-10."""
    def calls(e, n):
        for _ in range(n):
            for irk in (1, 2):
                e.interaction_push_particle(irk)
                e.particle_optimize(irk)
                e.interaction_collect_charge()
                e.field_solve_electric()
    nsteps = 30
    kw = dict(nparticle_max=96, nx=32)
    monkeypatch.setenv("PIC1DP_CALL_PAIR", "1")
    a = engine(amd, monkeypatch, True, kind, **kw)
    monkeypatch.setenv("PIC1DP_CALL_PAIR", "0")
    b = engine(amd, monkeypatch, True, kind, **kw)
    monkeypatch.setenv("PIC1DP_FUSE_SOLVE", "0")
    c = engine(amd, monkeypatch, True, kind, **kw)
    a.kernel_stats_enable(True)
    calls(a, nsteps)
    calls(b, nsteps)
    c.step(nsteps)
    # (the pair serves the six sums of one kept mode -- the default; with the tiles collect_charge owes the host the
    # whole half-step charge density, which the lean pair kernel never forms: three launches as before)
    assert a.kernel_stats(11)[1] == (nsteps - 1 if kind == 2 else 0) and b.kernel_stats(11)[1] == 0
    assert a.kernel_stats(3)[1] == 1 and a.kernel_stats(6)[1] == nsteps       # one first-sub-step pass in the whole run
    fa, fb, fc = a.get_field(), b.get_field(), c.get_field()
    ga, gb, gc = a.particles_download(), b.particles_download(), c.particles_download()
    # with the pair the call sites ARE pic1dp_hip_step's launches: bit for bit.  Without it the half-step field comes out
    # of the kept mode's content of the predicted charge density, solved again (k_field_solve_pred_sums): the same field
    # to rounding (tiles: the pair is not used, and pic1dp_hip_step's lean pair kernel groups the predicted charge otherwise)
    for k in ("electric", "mode_re", "mode_im", "chargeden"):
        if kind == 2:
            assert np.array_equal(fa[k], fc[k]), k
        else:
            assert np.array_equal(fa[k], fb[k]), k
        assert relerr(fa[k], fb[k]) < 1e-12 and relerr(fa[k], fc[k]) < 1e-12, k
    for k in "xvw":
        if kind == 2:
            assert np.array_equal(ga[k], gc[k]), k
        else:
            assert np.array_equal(ga[k], gb[k]), k
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-12 * max(1.0, np.max(np.abs(gb[k]))), k
    # what the host sees between the sub-steps is the half-step field, whichever way it was solved
    for e in (a, b):
        e.interaction_push_particle(1)
        e.interaction_collect_charge()
    assert np.array_equal(a.get_field(chargeden=False)["electric"], fa["electric"])    # solve_field not yet called
    for e in (a, b):
        e.field_solve_electric()
    ha, hb = a.get_field(chargeden=False), b.get_field(chargeden=False)
    for k in ("electric", "mode_re", "mode_im"):
        assert relerr(ha[k], hb[k]) < 1e-12, k
    assert relerr(ha["electric"], fa["electric"]) > 1e-3         # (it IS another field)
    for e in (a, b):
        e.interaction_push_particle(2)
        e.interaction_collect_charge()
        e.field_solve_electric()
    assert relerr(a.get_field()["electric"], b.get_field()["electric"]) < 1e-12
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-12 * max(1.0, np.max(np.abs(gb[k]))), k
    # a realistic marker count against the oracle
    kw = dict(nparticle_max=N, nx=96)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    monkeypatch.setenv("PIC1DP_CALL_PAIR", "1")
    e = engine(amd, monkeypatch, True, kind, **kw)
    for it in range(40):
        sim.step(1)
        calls(e, 1)
        if it % 7 == 3:        # looked at only now and then: the fast path runs in between
            assert abs(e.field_energy() / sim.field_energy() - 1.0) < 1e-10, it
    assert abs(e.field_energy() / sim.field_energy() - 1.0) < 1e-10
    assert e.kernel_stats(11)[1] == (39 if kind == 2 else 0)


@pytest.mark.parametrize("path", ["two_passes", "tiles", "tiles_two_modes"])
@pytest.mark.parametrize("drawn", ["4", "16"])
def test_chunks_drawn_in_the_other_whole_step_kernels(amd, monkeypatch, tuning, drawn, path):
    """the drawn chunk tail in k_step_half / k_step_full (two passes per step) and in the tiles' k_step_one, and in the
    diagnostics pass (k_ptcldist): every pair exactly once whoever takes it -- markers, energies and histograms against the
    same engine with every chunk dealt, to the order of the charge atomics"""
    kw = dict(nparticle_max=1_200_001, nx=256)
    if path == "tiles_two_modes":
        kw.update(nmode=2, modes=[1, 3])
    predict, kind = (False, None) if path == "two_passes" else (True, 1)
    monkeypatch.setenv("PIC1DP_DYN_TAIL", drawn)
    a = engine(amd, monkeypatch, predict, kind, **kw)
    monkeypatch.setenv("PIC1DP_DYN_TAIL", "0")
    b = engine(amd, monkeypatch, predict, kind, **kw)
    assert a.predict_kind() == (1 if predict else 0)
    a.step(20)
    b.step(20)
    assert np.max(np.abs(a.energy_history() / b.energy_history() - 1.0)) < 1e-11
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k
    da, db = a.ptcldist(0), b.ptcldist(0)
    for k in da:
        assert np.allclose(da[k], db[k], rtol=1e-11, atol=1e-12 * np.max(np.abs(db[k]))), k
    assert np.allclose(a.energy_sums(), b.energy_sums(), rtol=1e-12, atol=0)


@pytest.mark.parametrize("drawn", ["4", "16"])
def test_chunks_drawn_from_the_lds_counter(oracle_mod, amd, monkeypatch, tuning, drawn):
    """tuning knob PIC1DP_DYN_TAIL: the last n/16 of a workgroup's 64-pair chunks are drawn by its
    waves from an LDS counter instead of dealt -- every pair exactly once whoever takes it: the run against the oracle
    (1e-10 at every step) and the markers against the default engine's (the same arithmetic per marker: only the order of
    the charge atomics moves)"""
    kw = dict(nparticle_max=1_500_001, nx=512)
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    eo = []
    for _ in range(25):
        sim.step(1)
        eo.append(sim.field_energy())
    monkeypatch.setenv("PIC1DP_DYN_TAIL", drawn)
    a = engine(amd, monkeypatch, True, 2, **kw)
    monkeypatch.setenv("PIC1DP_DYN_TAIL", "0")
    b = engine(amd, monkeypatch, True, 2, **kw)
    a.step(25)
    b.step(25)
    assert np.max(np.abs(a.energy_history() / np.array(eo) - 1.0)) < 1e-10
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-11 * max(1.0, np.max(np.abs(gb[k]))), k


@pytest.mark.parametrize("case", ["two_stream1_singular", "field_jump", "no_perturbation", "full_f"])
def test_fixed_point_prediction_tiles_and_their_double_path(oracle_mod, amd, monkeypatch, case):
    """round 6: the prediction tiles (two kept modes) are 64-bit fixed-point sums in the LDS, scaled from
    per-species bounds on |w| and |c| that the host seeds from the markers it loads and the kernels raise; a marker beyond 16x
    its bound adds its terms straight into the global accumulators in doubles.  Cases that take that path: two-stream1's
    -f0'/f0 = v - 2/v near v = 0; a field set by the host that makes the weights jump by orders of magnitude within a step;
    a species without any perturbation (no bound at all: zeros are skipped); and full-f (q = p, one slice).  Every step's
    field energy against the two-pass engine 1e-11 and against the oracle 1e-10."""
    kw = dict(nparticle_max=N, nx=96, nmode=2, modes=[1, 3], init_nmode=2, init_mode=[1, 3], init_mode_cos=[0.0, 2e-6],
              init_mode_sin=[1e-5, 0.0])
    if case == "two_stream1_singular":
        kw.update(iptcldist=1, species_density=[1.0])
    elif case == "field_jump":
        kw.update(linear=1)           # (the linearised equations: the jump scales the weights, the orbits stay what they are)
    elif case == "no_perturbation":
        kw.update(init_mode_cos=[0.0, 0.0], init_mode_sin=[0.0, 0.0])
    elif case == "full_f":
        kw.update(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0])
    a = engine(amd, monkeypatch, True, 1, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    assert a.predict_kind() == 1 and b.predict_kind() == 0
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    a.kernel_stats_enable(True)
    nsteps = 30
    eo = []
    for it in range(nsteps):
        if case == "field_jump" and it == 10:     # the weights of every marker grow by ~1e7 in this step
            E = 1e3 * np.sin(2 * np.pi * np.arange(96) / 96)
            for e in (a, b):
                e.set_electric(E)
            sim.set_field(E)
        a.step(1)
        b.step(1)
        sim.step(1)
        eo.append(sim.field_energy())
    ea, eb, eo = a.energy_history(), b.energy_history(), np.array(eo)
    scale = max(np.max(eo), 1e-300)
    assert np.max(np.abs(ea - eb)) <= 1e-11 * scale
    assert np.max(np.abs(ea - eo)) <= 1e-10 * scale
    assert a.kernel_stats(6)[1] >= nsteps - 3        # one pass per step all along (the jump costs the prediction one step)
    ga, gb = a.particles_download(), b.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) <= 1e-10 * max(1.0, np.max(np.abs(gb[k]))), k


@pytest.mark.gpu
def test_fixed_point_bounds_follow_a_growing_mode(amd, monkeypatch):
    """The tiles' bounds have to follow weights of the size a run has -- ~1e-10 here (p ~ lx 16 f0 / N, w = 1e-7 p), 1e-11 at
    1e8 markers -- as an unstable mode multiplies them: a bound that stays behind sends more and more markers through the
    double path (global atomics: correct and slow).  (Found by reading: the workgroup's maximum travelled as float(mx + 3),
    which is 3 for every weight below 1e-7.)  kernel_stats 13 counts the terms that went the double path."""
    kw = dict(nparticle_max=N, nx=64, nmode=2, modes=[1, 2], init_nmode=1, init_mode=[1], init_mode_cos=[0.0],
              init_mode_sin=[1e-7], linear=1, dt=0.2)
    a = engine(amd, monkeypatch, True, 1, **kw)
    b = engine(amd, monkeypatch, False, **kw)
    assert a.predict_kind() == 1
    w0 = np.max(np.abs(a.particles_download()["w"]))
    bound0, slow0 = a.kernel_stats(13)
    assert slow0 == 0 and w0 <= bound0 <= 1.01 * w0 and w0 < 1e-8
    a.kernel_stats_enable(True)
    nsteps = 300
    a.step(nsteps)
    b.step(nsteps)
    ea, eb = a.energy_history(), b.energy_history()
    assert np.max(np.abs(ea - eb)) <= 1e-11 * np.max(eb)
    assert a.kernel_stats(6)[1] >= nsteps - 2
    w1 = np.max(np.abs(a.particles_download()["w"]))
    bound1, slow1 = a.kernel_stats(13)
    assert w1 >= 20.0 * w0, (w0, w1)                       # the mode grew ...
    assert 0.9 * w1 <= bound1 <= 1.5 * w1, (bound1, w1)    # ... the bound with it (as of the last launches: a step behind) ...
    assert slow1 <= 1e-6 * N * nsteps, slow1               # ... and the double path stayed the exception
