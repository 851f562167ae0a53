#!/usr/bin/env python3
"""ramp_probe.py -- time per step in consecutive batches right after the load:
does the GPU need a while to reach its streaming rate?"""
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
eng.sync()
out = []
for b in range(60):
    t0 = time.perf_counter()
    eng.step(10)
    eng.sync()
    out.append((time.perf_counter() - t0) / 10 * 1e3)
print(" ".join("%.3f" % x for x in out))
