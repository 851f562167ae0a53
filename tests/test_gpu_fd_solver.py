"""The opt-in finite-difference field solver (BASELINE north_star's
"tridiagonal / cyclic-reduction HIP kernel").  It is NOT the reference's solver
(the reference projects on the kept Fourier modes), so it is checked
against its own CPU statement (Thomas algorithm, oracle/) and against the
analytic response of the discrete operators -- never against the reference."""
import ctypes as C

import numpy as np
import pytest

from util import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nx", [3, 4, 17, 64, 192, 1000, 1024, 4096])
def test_fd_solver_matches_thomas_algorithm(oracle_mod, amd, nx):
    eng = amd.Pic1dp(amd.make_input(nparticle_max=16, nx=nx))
    eng.set_field_solver(1)
    rng = np.random.default_rng(nx)
    rho = rng.standard_normal(nx)
    eng.set_chargeden(rho)
    eng.field_solve_electric()
    E = eng.get_field()["electric"]
    want = np.empty(nx)
    oracle_mod.lib().orc_field_solve_fd(C.byref(oracle_mod.make_input(nx=nx)), rho, want)
    # two different direct methods on a system of condition number ~ (2 nx / pi)^2
    assert relerr(E, want) < 1e-16 * max(1.0, (2 * nx / np.pi) ** 2) * 50
    assert abs(E.sum()) < 1e-9 * np.abs(E).sum() + 1e-12            # zero-mean field
    # Gauss's law in its discrete form: (E[i+1]-E[i-1])/(2h) = avg of rho - <rho> over i-1, i+1 ... check
    # via the composed operator instead: -(phi'' ) = rho - mean  =>  D0 D0 phi != D2 phi, so test the potential
    # equation directly by rebuilding phi from E is not unique; the analytic test below covers the response.


@pytest.mark.parametrize("nx,m", [(64, 1), (192, 1), (192, 7), (1024, 3), (4096, 100)])
def test_fd_solver_analytic_response(amd, nx, m):
    """rho = cos(k x): phi = rho / K^2 with K^2 = (2/h)^2 sin^2(k h / 2) and
    E = sin(k h)/h * phi_amplitude * sin(k x)"""
    eng = amd.Pic1dp(amd.make_input(nparticle_max=16, nx=nx))
    eng.set_field_solver(1)
    lx = eng.inp.lx
    h = lx / nx
    k = 2 * np.pi * m / lx
    x = np.arange(nx) * h
    eng.set_chargeden(np.cos(k * x) + 0.37)          # the mean must drop out
    eng.field_solve_electric()
    E = eng.get_field()["electric"]
    K2 = (2 / h) ** 2 * np.sin(k * h / 2) ** 2
    want = np.sin(k * h) / h / K2 * np.sin(k * x)
    assert np.max(np.abs(E - want)) < 1e-9 * np.max(np.abs(want))
    # continuum limit: E -> sin(kx)/k
    if m * 20 < nx:
        assert abs(np.max(np.abs(E)) * k - 1.0) < (k * h) ** 2


def test_fd_solver_in_a_run(oracle_mod, amd):
    """a run with the alternative solver: for the default single kept mode the
    linear growth rate is the same physics (the unstable mode dominates), the
    field now also carries the other modes' noise"""
    kw = dict(nparticle_max=2_000_000, nx=128)
    a = amd.Pic1dp(amd.make_input(**kw))
    b = amd.Pic1dp(amd.make_input(**kw))
    b.set_field_solver(1)
    series = []
    for e in (a, b):
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.step(900)
        series.append(e.energy_history())
    t = (np.arange(900) + 1) * 0.05
    g = [oracle_mod.growthrate_energy_fit(t, s, 25.0, 42.0) for s in series]
    assert abs(g[0] / 0.16766 - 1.0) < 0.03
    assert abs(g[1] / g[0] - 1.0) < 0.08          # k_eff of the FD operators differs slightly; noise floor higher
    with pytest.raises(amd.Pic1dpError):
        amd.Pic1dp(amd.make_input(nparticle_max=16, nx=8192)).set_field_solver(1)
    with pytest.raises(amd.Pic1dpError):
        a.set_field_solver(2)
