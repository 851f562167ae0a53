#!/usr/bin/env python3
"""ab_launch.py -- launch shapes of the whole-step kernels compared inside one
process (same arrays; between processes the placement of the arrays moves kernel
times by several percent)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
eng.step(40)
shapes = [(0, 0), (256, 8), (512, 4), (512, 3), (640, 3), (768, 2), (896, 2), (1024, 2), (1024, 1), (0, 0)]
for rnd in range(2):
    for th, bpc in shapes:
        eng.set_launch(th, bpc)
        eng.step(2)
        eng.sync()
        eng.kernel_stats_enable(True)
        eng.timers_reset()
        eng.step(steps)
        eng.sync()
        (hm, hn), (fm, fn) = eng.kernel_stats(3), eng.kernel_stats(4)
        print("round %d threads %4d x %d/CU: step_half %.4f ms  step_full %.4f ms" % (rnd, th, bpc, hm / hn, fm / fn), flush=True)
