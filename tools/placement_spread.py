#!/usr/bin/env python3
"""placement_spread.py -- does kernel time depend on where hipMalloc put the marker
arrays?  Creates the engine several times in one process and times the whole-step
kernels each time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
from pic1dp_amd import probe  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
keep = []
for r in range(reps):
    eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=1024))
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.step(3)
    eng.sync()
    eng.kernel_stats_enable(True)
    eng.step(20)
    eng.sync()
    (hm, hn), (fm, fn) = eng.kernel_stats(3), eng.kernel_stats(4)
    print("create %d: step_half %.4f ms  step_full %.4f ms   probe 4r3w %.0f GB/s" % (
        r, hm / hn, fm / fn, probe.stream(4, 3, n, 10)), flush=True)
    if "--keep" in sys.argv:
        keep.append(eng)          # keep the allocation alive: the next engine lands elsewhere
    else:
        eng.close()
