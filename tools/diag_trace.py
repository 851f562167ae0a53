#!/usr/bin/env python3
"""diag_trace.py -- forty steps at the reference's output cadence with the diagnostics taken inside the steps (to put
under rocprofv3 --kernel-trace --stats: which kernels an output costs)
    python tools/diag_trace.py [markers] [nx]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
eng.set_output_fusion(int(os.environ.get("FUSE", "1")))     # 0 own pass, 1 where it pays, 2 always inside the step
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
for _ in range(4):
    eng.step(10)
    eng.output_scalars()
    eng.ptcldist(0)
eng.sync()
