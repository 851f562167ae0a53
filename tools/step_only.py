#!/usr/bin/env python3
"""step_only.py -- nothing but pic1dp_hip_step on the default physics (for kernel traces):
    python tools/step_only.py [particles] [nx] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
import json  # noqa: E402
extra = json.loads(os.environ.get("PIC1DP_INPUT", "{}"))
# PIC1DP_NPE: reference ranks reproduced as virtual ranks of this one process (the field solve then sums in the
# npe-rank order: npe partial chains side by side)
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx, **extra), npe=int(os.environ.get("PIC1DP_NPE", "1")))
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
eng.step(steps)
eng.sync()
