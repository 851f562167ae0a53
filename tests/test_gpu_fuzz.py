"""Randomised differential test: seeded random configurations (grid size, kept
modes, distribution, linear / nonlinear / full-f, species, masses and
temperatures, virtual ranks, unloaded tails, odd marker counts) run for a few
time steps on the GPU and in the oracle; the field-energy series must agree to
1e-10 and positions to the trajectory-following tolerance."""
import numpy as np
import pytest

from util import both_inputs

pytestmark = pytest.mark.gpu


def random_case(rng):
    nsp = int(rng.integers(1, 4))
    nx = int(rng.choice([2, 3, 17, 64, 100, 192, 255, 512, 1000, 2048]))
    nmode_max = max(1, min(6, nx // 2))
    nmode = int(rng.integers(1, nmode_max + 1))
    modes = sorted(rng.choice(np.arange(1, max(2, nx // 2 + 1)), size=nmode, replace=False).tolist())
    dist = int(rng.integers(0, 4))
    deltaf = 1 if dist in (1, 2, 3) or rng.random() < 0.7 else 0
    linear = int(rng.random() < 0.3) if deltaf else 0
    nmax = int(rng.integers(2000, 30000))
    ninit = [int(nmax - rng.integers(0, nmax // 4)) if rng.random() < 0.3 else nmax for _ in range(nsp)]
    pow2 = rng.random() < 0.4
    def pick(vals_p2, lo, hi):
        return float(rng.choice(vals_p2)) if pow2 else float(rng.uniform(lo, hi))
    kw = dict(
        nspecies=nsp, nx=nx, nmode=nmode, modes=[int(m) for m in modes], iptcldist=dist, deltaf=deltaf,
        linear=linear, nparticle_max=nmax, species_nparticle_init=ninit,
        species_charge=[float(rng.choice([-1.0, 1.0, -2.0])) for _ in range(nsp)],
        species_mass=[pick([0.5, 1.0, 2.0, 4.0], 0.5, 30.0) for _ in range(nsp)],
        species_temperature=[pick([0.5, 1.0, 2.0], 0.5, 2.0) for _ in range(nsp)],
        species_temperature2=[pick([0.5, 1.0, 2.0], 0.5, 2.0) for _ in range(nsp)],
        species_density=[float(rng.uniform(0.6, 0.95)) for _ in range(nsp)],
        species_v0=[float(rng.uniform(2.0, 5.0)) for _ in range(nsp)],
        lx=float(rng.uniform(5.0, 40.0)), dt=float(rng.choice([0.05, 0.1, 0.02])),
        init_nmode=2, init_mode=[1, int(rng.integers(1, 4))],
        init_mode_cos=[float(rng.uniform(-1e-4, 1e-4)), 0.0], init_mode_sin=[1e-5, float(rng.uniform(-1e-5, 1e-5))],
        multirand_al_int=int(rng.choice([1, 2, 3])), multirand_warmup=int(rng.integers(0, 3)),
    )
    npe = int(rng.choice([1, 1, 2, 3]))
    return kw, npe


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration(oracle_mod, amd, seed):
    rng = np.random.default_rng(1000 + seed)
    kw, npe = random_case(rng)
    o, g = both_inputs(oracle_mod, amd, **kw)
    sim = oracle_mod.Sim(o, npe=npe)
    assert sim.load() == 0
    eng = amd.Pic1dp(g, npe=npe)
    eng.particle_load()
    for isp in range(kw["nspecies"]):
        got = eng.particles_download(isp)
        npv = eng.local_sizes(isp)[1]
        for k in "xvpw":
            assert np.array_equal(got[k][:npv], sim.gather(k, isp)), (k, isp)
    if seed % 2:
        eng.set_step_mode(1)
    sim.collect_charge()
    sim.solve_field()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    nsteps = 6
    eo = [sim.field_energy()]
    eg = [eng.field_energy()]
    for _ in range(nsteps):
        sim.step(1)
        eo.append(sim.field_energy())
    eng.step(nsteps)
    eg = np.concatenate([eg, eng.energy_history()])
    eo = np.array(eo)
    # energies can be tiny (or zero when no kept mode is excited): absolute floor
    scale = max(np.max(eo), 1e-300)
    assert np.max(np.abs(eg - eo)) <= 1e-10 * scale, (kw, npe)
    for isp in range(kw["nspecies"]):
        got = eng.particles_download(isp)
        npv = eng.local_sizes(isp)[1]
        assert np.max(np.abs(got["x"][:npv] - sim.gather("x", isp))) < 1e-8
        assert np.max(np.abs(got["v"][:npv] - sim.gather("v", isp))) < 1e-8
