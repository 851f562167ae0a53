"""Marker optimisation: particle_merge / particle_remove /
particle_split (src/pic1dp_particle.F90:356-813) through the engine against the
oracle.  The routines are sequential and run on the host inside the library; the
tests feed both sides identical markers right before an optimisation event and
require identical results (positions, velocities, weights, counts), then check
whole runs statistically."""
import os

import numpy as np
import pytest

from util import both_inputs, relerr

pytestmark = pytest.mark.gpu


def synced_pair(oracle, amd, npe, **kw):
    o, g = both_inputs(oracle, amd, **kw)
    sim = oracle.Sim(o, npe=npe)
    assert sim.load() == 0
    eng = amd.Pic1dp(g, npe=npe)
    eng.particle_load()
    sim.collect_charge()
    sim.solve_field()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    return sim, eng


def compare_blocks(sim, eng, npe, isp=0):
    """valid markers of every block, in block order, bit for bit"""
    nalloc, npv = eng.local_sizes(isp)
    assert npv == sum(sim.rank_np(r, isp) for r in range(npe))
    got = eng.particles_download(isp)
    for k in "xvpw":
        assert np.array_equal(got[k][:npv], sim.gather(k, isp)), k
    return npv


def drive_to_event(sim, eng, nsteps):
    """advance both sides; after every step put the oracle's markers on the
    engine's values so that the next optimisation sees identical inputs"""
    for _ in range(nsteps):
        sim.step(1)
        eng.step(1)


@pytest.mark.parametrize("kind,kw", [
    ("merge", dict(nmerge=1, tmerge=[0.3], thshmerge=[0.5])),
    ("merge_all", dict(nmerge=2, tmerge=[0.3, 0.4], thshmerge=[2.0, 2.0])),
    ("remove_profile", dict(nremove=1, tremove=[0.3], typeremove=2)),
    ("remove_threshold", dict(nremove=1, tremove=[0.3], typeremove=1, thshremove=[0.4], remove_frac=0.7)),
    ("split", dict(nsplit=1, tsplit=[0.3], thshsplit=[0.3], split_ngroup=3)),
    ("split_until_full", dict(nsplit=1, tsplit=[0.3], thshsplit=[0.01], split_ngroup=5)),
    ("all_three", dict(nmerge=1, tmerge=[0.3], thshmerge=[0.3], nremove=1, tremove=[0.3], typeremove=2,
                       nsplit=1, tsplit=[0.3], thshsplit=[0.6])),
], ids=lambda v: v if isinstance(v, str) else "")
@pytest.mark.parametrize("npe", [1, 3])
def test_optimisation_event_is_bit_identical(oracle_mod, amd, kind, kw, npe):
    base = dict(nparticle_max=60000, species_nparticle_init=[36000], nx=32, nv=64)
    n_after = aligned_event(oracle_mod, amd, npe, dict(base, **kw))
    if kind.startswith("merge") or kind.startswith("remove"):
        assert n_after < 36000
    if kind.startswith("split"):
        assert n_after > 36000


def aligned_event(oracle_mod, amd, npe, kw, steps_before=5):
    """both sides to the step in which the event fires (t = 0.25 + dt >= 0.3 with the default dt), the oracle's markers
    put on the engine's values, the event step by hand on both sides; returns the valid markers after the event"""
    sim, eng = synced_pair(oracle_mod, amd, npe, **kw)
    # identical steps (the initial steps agree bit for bit in x, v, and w to rounding),
    # then align the markers exactly and run the event step by hand on both sides
    sim.step(steps_before)
    eng.step(steps_before)
    got = eng.particles_download()
    off = 0
    for r in range(npe):
        n = sim.rank_np(r)
        for k in "xvpw":
            sim.array(r, 0, k)[:n] = got[k][off:off + n]
        off += n
    sim.set_field(eng.get_field()["electric"])
    n_after = None
    for irk in (1, 2):
        sim.push(irk)
        eng.interaction_push_particle(irk)
        # weights may differ in the last bits (exp): re-align after either push
        g = eng.particles_download()
        off = 0
        for r in range(npe):
            n = sim.rank_np(r)
            sim.array(r, 0, "w")[:n] = g["w"][off:off + n]
            off += n
        did_o = sim.optimize(irk)
        did_g = eng.particle_optimize(irk)
        assert did_o == did_g == (irk == 2)
        if irk == 2:
            n_after = compare_blocks(sim, eng, npe)
        sim.collect_charge()
        eng.interaction_collect_charge()
        sim.solve_field()
        eng.field_solve_electric()
        sim.set_field(eng.get_field()["electric"])
    assert relerr(eng.get_field()["chargeden"], sim.get_field()[1]) < 1e-11
    # nothing further is due
    assert not eng.particle_optimize(2)
    return n_after


@pytest.mark.parametrize("threads", ["1", "2", "7"])
def test_event_walks_on_host_threads_change_nothing(oracle_mod, amd, monkeypatch, threads):
    """the blocks of an event are walked side by side on host threads (one random stream, one set of
    keys per block: the reference runs them as separate ranks, src/pic1dp_particle.F90:411-746) -- whatever the number
    of threads, every block bit for bit what the oracle's routines leave (seven blocks; one, two, seven workers)"""
    monkeypatch.setenv("PIC1DP_OPT_THREADS", threads)
    kw = dict(nparticle_max=70000, species_nparticle_init=[42000], nx=32, nv=64, nmerge=1, tmerge=[0.3], thshmerge=[0.3],
              nremove=1, tremove=[0.3], typeremove=2, nsplit=1, tsplit=[0.3], thshsplit=[0.6])
    aligned_event(oracle_mod, amd, 7, kw)


def random_event_case(rng):
    nmax = int(rng.integers(3000, 70000))
    ninit = int(nmax * rng.uniform(0.35, 1.0))
    kw = dict(nparticle_max=nmax, species_nparticle_init=[ninit], nx=int(rng.choice([4, 8, 32, 100])),
              nv=int(rng.choice([8, 16, 64, 128])), iptcldist=int(rng.choice([0, 2, 3, 3])))
    kinds = rng.permutation(["merge", "remove", "split"])[:int(rng.integers(1, 4))]
    for k in kinds:
        if k == "merge":
            kw.update(nmerge=1, tmerge=[0.3], thshmerge=[float(rng.choice([0.05, 0.3, 0.6, 2.0]))])
        elif k == "remove":
            kw.update(nremove=1, tremove=[0.3], typeremove=int(rng.choice([1, 2])),
                      thshremove=[float(rng.uniform(0.1, 0.8))], remove_frac=float(rng.uniform(0.2, 0.9)))
        else:
            kw.update(nsplit=1, tsplit=[0.3], thshsplit=[float(rng.choice([0.01, 0.2, 0.5, 0.9]))],
                      split_ngroup=int(rng.integers(2, 7)))
    return kw, int(rng.choice([1, 1, 2, 3, 5]))


_BASE = int(os.environ.get("PIC1DP_FUZZ_BASE", "0"))
_MULT = int(os.environ.get("PIC1DP_FUZZ_MULT", "1"))


@pytest.mark.parametrize("seed", range(_BASE, _BASE + 12 * _MULT))
def test_random_optimisation_events(oracle_mod, amd, seed):
    """seeded random events (kinds and their combinations, thresholds, both kinds of removal, group sizes, marker counts
    with and without free tail slots, grids, distributions, 1 ... 5 reference ranks) with the markers on the device,
    against the oracle on aligned inputs: every valid marker of every block bit for bit.  A longer campaign:
    PIC1DP_FUZZ_BASE=1000 PIC1DP_FUZZ_MULT=20 (as tests/test_gpu_fuzz.py)"""
    kw, npe = random_event_case(np.random.default_rng(77000 + seed))
    aligned_event(oracle_mod, amd, npe, kw)


def test_whole_step_path_runs_the_events(oracle_mod, amd):
    """pic1dp_hip_step notices due events and takes those steps through the
    sub-step kernels; marker counts follow the oracle's exactly as long as the
    thresholds are not borderline, energies statistically"""
    kw = dict(nparticle_max=200000, species_nparticle_init=[120000], nx=64, nv=64,
              nmerge=1, tmerge=[1.0], thshmerge=[0.2], nsplit=1, tsplit=[2.0], thshsplit=[0.5])
    sim, eng = synced_pair(oracle_mod, amd, 2, **kw)
    sim.step(60)
    eng.step(60)
    n_g = eng.local_sizes()[1]
    n_o = sim.rank_np(0) + sim.rank_np(1)
    assert n_g != 120000 and abs(n_g - n_o) <= 0.002 * n_o
    assert abs(eng.field_energy() / sim.field_energy() - 1.0) < 0.05
    assert relerr(eng.energy_sums(), sim.energy_sums()) < 1e-2


def test_optimisation_needs_loader_stream_and_delta_f(amd):
    inp = amd.make_input(nparticle_max=2000, nx=16, nv=32, nremove=1, tremove=[0.05])
    eng = amd.Pic1dp(inp)
    n = 2000
    rng = np.random.default_rng(0)
    eng.particles_upload(rng.uniform(0, inp.lx, n), rng.uniform(-8, 8, n), rng.uniform(0.5, 1, n),
                         rng.uniform(-1e-3, 1e-3, n))
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.interaction_push_particle(1)
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.interaction_push_particle(2)
    with pytest.raises(amd.Pic1dpError) as ei:      # remove needs the block's random stream
        eng.particle_optimize(2)
    assert ei.value.code == 4
    full_f = amd.Pic1dp(amd.make_input(nparticle_max=2000, nx=16, nv=32, nmerge=1, tmerge=[0.05], deltaf=0,
                                       iptcldist=0, species_density=[1.0], species_v0=[0.0]))
    full_f.particle_load()
    full_f.interaction_collect_charge()
    full_f.field_solve_electric()
    full_f.step(3)
    assert full_f.local_sizes()[1] == 2000            # "now only support optimization for delta f"


def test_call_site_path_needs_the_library_clock(amd):
    """a host that swaps only the call sites must hand its clock to the
    library after every step (pic1dp_hip_set_time) -- particle_optimize compares the event times
    with the library's time + dt like src/pic1dp_particle.F90:742-770 with global_time + dt.  With
    the clock the merge fires in the same step as under pic1dp_hip_step; without it, never."""
    kw = dict(nparticle_max=120000, species_nparticle_init=[90000], nx=32, nv=64, nmerge=1, tmerge=[0.2],
              thshmerge=[0.3])

    def fresh():
        e = amd.Pic1dp(amd.make_input(**kw))
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        return e

    def drive(e, nsteps, clock):
        fired = []
        for it in range(nsteps):
            for irk in (1, 2):
                e.interaction_push_particle(irk)
                if e.particle_optimize(irk):
                    fired.append(it)
                e.interaction_collect_charge()
                e.field_solve_electric()
            if clock:
                e.set_time(e.itime + 1, e.time + e.inp.dt)
        return fired

    ref = fresh()
    counts = []
    for it in range(6):
        ref.step(1)
        counts.append(ref.local_sizes()[1])
    step_fired = [it for it in range(6) if counts[it] != (counts[it - 1] if it else 90000)]
    assert step_fired == [3]                    # the step that starts at t = 0.15: 0.15 + dt >= 0.2
    with_clock = fresh()
    assert drive(with_clock, 6, True) == step_fired
    assert with_clock.local_sizes()[1] == counts[-1]
    without = fresh()
    assert drive(without, 6, False) == [] and without.local_sizes()[1] == 90000


@pytest.mark.parametrize("kind,kw", [
    ("merge", dict(nmerge=1, tmerge=[0.1], thshmerge=[0.5])),
    ("remove_profile", dict(nremove=1, tremove=[0.1], typeremove=2)),
    ("remove_threshold", dict(nremove=1, tremove=[0.1], typeremove=1, thshremove=[0.4], remove_frac=0.7)),
    ("split", dict(nsplit=1, tsplit=[0.1], thshsplit=[0.3], split_ngroup=3)),
], ids=lambda v: v if isinstance(v, str) else "")
def test_event_moves_keys_not_markers_over_pcie(amd, monkeypatch, kind, kw):
    """the markers stay on the device during an event -- |delta f|(v) is summed there (in the
    reference's order of additions), one small key per marker goes to the host for the sequential walk, and the
    walk's decisions come back: at most 9 B per marker between host and device (the host-side pass moves 64 per
    allocated slot), with the same markers afterwards as the pass on host copies (PIC1DP_OPT_HOST=1)."""
    n = 2_000_000
    base = dict(nparticle_max=n + n // 2, species_nparticle_init=[n], nx=64, nv=64, **kw)

    def run(host):
        monkeypatch.setenv("PIC1DP_OPT_HOST", "1" if host else "0")
        e = amd.Pic1dp(amd.make_input(**base), npe=2)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.step(1)                       # t = 0.05: the event fires in the next step (0.05 + dt >= 0.1)
        before = e.kernel_stats(8)[1]
        e.step(1)
        return e, e.kernel_stats(8)[1] - before

    dev, bytes_dev = run(False)
    ref, bytes_host = run(True)
    nalloc, npv = dev.local_sizes()
    assert npv != n and npv == ref.local_sizes()[1]
    assert bytes_host == 64 * nalloc
    print("%s: %.2f B per marker between host and device (host-side pass: %.0f)" % (kind, bytes_dev / n, bytes_host / n))
    assert 0 < bytes_dev <= 9 * n, bytes_dev / n
    # the two engines' fields differ in the last bits by then (order of the charge atomics), hence v and w too: the same
    # markers in the same slots to rounding (bit for bit against the oracle on aligned inputs: the tests above)
    a, b = dev.particles_download(), ref.particles_download()
    for k in "xvpw":
        assert np.allclose(a[k][:npv], b[k][:npv], rtol=1e-9, atol=1e-18), k
    assert relerr(dev.energy_sums(), ref.energy_sums()) < 1e-12      # (the whole local vector: tail slots too)
