#!/usr/bin/env python3
"""diag_bench.py -- cost of the output_all diagnostics (N1) next to a time step.
The marker diagnostics are cached until the markers change, so every sample
takes a time step first; the step's own time is subtracted."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=1024))
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
eng.step(5)
eng.sync()


def sample(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


t_step = sample(lambda: (eng.step(1), eng.sync()))
print("%-40s %.3f ms" % ("step", t_step), flush=True)
for name, fn in (("ptcldist (first use after a step)", lambda: eng.ptcldist(0)),
                 ("output_scalars (first use after a step)", eng.output_scalars),
                 ("output_scalars + ptcldist = output_all", lambda: (eng.output_scalars(), eng.ptcldist(0)))):
    t = sample(lambda: (eng.step(1), eng.sync(), fn()))
    print("%-40s %.3f ms" % (name, t - t_step), flush=True)

# the reference's cadence: ten steps, then output_all -- with the diagnostics taken inside the
# tenth step (set_output_fusion) and as a separate pass
eng.step(40)
for fuse in (False, True, False, True):
    eng.set_output_fusion(fuse)
    eng.set_time(0, 0.0)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.step(10)
        eng.output_scalars()
        eng.ptcldist(0)
    dt = (time.perf_counter() - t0) / 10 * 1e3
    print("10 steps + output_all, fusion %-5s %.3f ms  (%.3f ms over 10 bare steps)" % (fuse, dt, dt - 10 * t_step), flush=True)
