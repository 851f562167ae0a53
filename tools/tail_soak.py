#!/usr/bin/env python3
"""tail_soak.py -- the marker launch's tail hand-off (agent atomics + arrival ticket + sc1 loads, device_xchg.hpp step_tail)
under the load it meets on a full GPU: 1e8 markers (2048 workgroups finishing at uneven times), a one-rank RCCL communicator,
N steps with the charge packed by the tail against the same run with the separate packing launch (PIC1DP_TAIL=0).  A stale
or missing contribution of ONE workgroup in ONE step would move the field energy of that step by parts in 1e3; the two
histories must agree to the order of the charge atomics (1e-12).   python tools/tail_soak.py [markers] [nx] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
hist = {}
for tail in ("1", "0"):
    os.environ["PIC1DP_TAIL"] = tail
    import pic1dp_amd
    e = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx), npe=8)
    e.particle_load()
    e.comm_init(e.comm_unique_id())
    e.interaction_collect_charge()
    e.field_solve_electric()
    e.step(steps)
    hist[tail] = e.energy_history().copy()
    print("PIC1DP_TAIL=%s: %d steps, tails %d, int E^2 dx first %.15e last %.15e" % (tail, steps, e.kernel_stats(10)[1], hist[tail][0], hist[tail][-1]), flush=True)
    e.close()
d = np.abs(hist["1"] / hist["0"] - 1.0)
print("max relative difference of the field energy over %d steps: %.3e (at step %d)" % (steps, d.max(), int(d.argmax()) + 1))
sys.exit(0 if d.max() < 1e-11 else 1)
