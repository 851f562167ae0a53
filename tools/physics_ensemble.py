#!/usr/bin/env python3
"""physics_ensemble.py -- round 6: the linear growth / damping rates of the three BASELINE.md physics
anchors at 10^8 markers as an ENSEMBLE over RNG streams (pic1dp_hip_set_seed_offset: member m loads reference block b from
stream mype = b + 16 m; the reference compares runs with different seeds the same way, tools/runinfo.py:94-122,136-231),
mean +- sigma against the roots of the Vlasov dispersion relation (tools/dispersion.py:130-157):

    bump-on-tail  2 gamma = +0.16766      (omega = 1.1693765077 + 0.0838310511 i, k = 0.36)
    two-stream    2 gamma = +0.30505      (omega = 0 + 0.1525251736 i, v0 = 3, k = 0.36)
    Landau        2 gamma = -0.30672, omega_r = 1.41566   (omega = 1.4156618886 - 0.1533594669 i, k = 0.5)

and how the fitted rate moves with the time step: the RK2 (midpoint) scheme of src/pic1dp.F90:79-93 amplifies a mode
exp(z), z = (gamma - i omega) dt, by 1 + z + z^2/2 = exp(z) (1 - z^3/6 + ...), i.e. it adds -Re(z^3)/6 / dt =
+gamma omega^2 dt^2 / 2 (1 - gamma^2 / (3 omega^2)) to gamma: +0.17 % at dt = 0.05 for the bump-on-tail wave, a quarter of
that at dt = 0.025 -- a bias that only shows once the markers' noise is below it.

    python tools/physics_ensemble.py [--members 8] [--markers 1e8] [--dts 0.05,0.025] [--cases bump,two_stream,landau]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    "bump": dict(kw=dict(nx=1024), t_fit=(15.0, 45.0), two_gamma=2 * 0.0838310511, omega=1.1693765077),
    "two_stream": dict(kw=dict(nx=512, iptcldist=2, species_density=[1.0], species_v0=[3.0]), t_fit=(12.0, 28.0),
                       two_gamma=2 * 0.1525251736, omega=0.0),
    "landau": dict(kw=dict(nx=1024, iptcldist=0, species_density=[1.0], species_v0=[0.0], lx=4 * np.pi, linear=1),
                   t_fit=(3.0, 16.0), two_gamma=-2 * 0.1533594669, omega=1.4156618886),
}


def fit_rate(t, e, t1, t2):
    """least-squares slope of ln(int E^2 dx), tools/OutputData.py:153-170"""
    i1 = int(np.searchsorted(t, t1)) - 1
    i2 = int(np.searchsorted(t, t2))
    tt, ln = t[i1:i2], np.log(e[i1:i2])
    n = i2 - i1
    return (n * np.sum(tt * ln) - np.sum(tt) * np.sum(ln)) / (n * np.sum(tt * tt) - np.sum(tt) ** 2)


def fit_damped_wave(t, e, t1, t2, two_gamma0, omega0):
    """a standing wave that decays: int E^2 dx = A exp(2 gamma t) cos^2(omega t + phi); least squares on every sample of the
    window (scipy), start values from theory; returns 2 gamma, omega"""
    from scipy.optimize import least_squares
    m = (t >= t1) & (t <= t2)
    tt, ee = t[m], e[m]
    scale = ee[0] * np.exp(-two_gamma0 * tt[0])

    def resid(p):
        a, g2, om, ph = p
        model = a * scale * np.exp(g2 * tt) * np.cos(om * tt + ph) ** 2
        return (model - ee) / (scale * np.exp(two_gamma0 * tt))       # relative to the envelope: every period counts alike

    best = None
    for ph0 in np.linspace(0.0, np.pi, 8, endpoint=False):
        r = least_squares(resid, [1.0, two_gamma0, omega0, ph0], x_scale=[1.0, 0.01, 0.01, 0.1])
        if best is None or r.cost < best.cost:
            best = r
    return best.x[1], best.x[2]


def rk2_bias(two_gamma, omega, dt):
    """what the midpoint scheme adds to 2 gamma of a mode exp((gamma - i omega) t): -2 Re(z^3) / (6 dt), z = (gamma - i omega) dt"""
    z = (0.5 * two_gamma - 1j * omega) * dt
    return -2.0 * (z ** 3).real / (6.0 * dt)


def member(pic1dp_amd, case, markers, dt, m, npe=16):
    c = CASES[case]
    eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=int(markers), dt=dt, **c["kw"]), npe=npe)
    eng.set_seed_offset(npe * m)
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    e0 = eng.field_energy()
    nsteps = int(round((c["t_fit"][1] + 1.0) / dt))
    eng.step(nsteps)
    e = np.concatenate([[e0], eng.energy_history()])
    t = np.arange(nsteps + 1) * dt
    eng.close()
    if case == "landau":
        return fit_damped_wave(t, e, c["t_fit"][0], c["t_fit"][1], c["two_gamma"], c["omega"])
    return fit_rate(t, e, *c["t_fit"]), None


def ensemble(pic1dp_amd, case, markers, dt, members, log=print):
    c = CASES[case]
    rates, oms = [], []
    for m in range(members):
        g2, om = member(pic1dp_amd, case, markers, dt, m)
        rates.append(g2)
        if om is not None:
            oms.append(om)
    rates = np.array(rates)
    mean, sig = float(np.mean(rates)), float(np.std(rates, ddof=1))
    sem = sig / np.sqrt(members)
    th = c["two_gamma"]
    bias = rk2_bias(th, c["omega"], dt)
    out = dict(case=case, dt=dt, members=members, markers=int(markers), rates=rates.tolist(), mean=mean, sigma=sig, sem=sem,
               theory=th, rk2_bias=bias)
    log("%-10s dt %.4f  %d members x %.3g markers: 2 gamma = %+.6f +- %.6f (sigma; %.3f %% of |2 gamma|), mean +- %.6f | theory %+.6f: "
        "mean - theory = %+.6f = %+.2f sigma_mean (%+.3f %%) | with the RK2 term %+.6f: %+.2f sigma_mean"
        % (case, dt, members, markers, mean, sig, 100 * sig / abs(th), sem, th, mean - th, (mean - th) / sem,
           100 * (mean - th) / abs(th), bias, (mean - th - bias) / sem))
    if oms:
        oms = np.array(oms)
        out.update(omega_mean=float(np.mean(oms)), omega_sigma=float(np.std(oms, ddof=1)), omega_theory=c["omega"])
        log("           omega_r = %.6f +- %.6f (sigma) | theory %.6f: %+.3f %%"
            % (out["omega_mean"], out["omega_sigma"], c["omega"], 100 * (out["omega_mean"] / c["omega"] - 1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--markers", type=float, default=1e8)
    ap.add_argument("--dts", default="0.05,0.025")
    ap.add_argument("--cases", default="bump,two_stream,landau")
    a = ap.parse_args()
    import pic1dp_amd
    for case in a.cases.split(","):
        for dt in [float(x) for x in a.dts.split(",")]:
            t0 = time.time()
            ensemble(pic1dp_amd, case, a.markers, dt, a.members)
            print("           (%.1f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
