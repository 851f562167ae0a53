#!/usr/bin/env python3
"""output_host_cost.py -- what an output_all costs on the HOST side of the library: wall clock of each call of the
sequence output_scalars, get_field, ptcldist at a marker count whose kernels are negligible, after one step each
(so that nothing is cached).   python tools/output_host_cost.py [markers] [nx]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 192
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
eng.set_output_fusion(1)
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
eng.step(20)
REPS = 200
acc = {}
for _ in range(REPS):
    eng.step(1)
    eng.sync()
    for name, fn in (("output_scalars", eng.output_scalars), ("get_field", eng.get_field), ("ptcldist", lambda: eng.ptcldist(0))):
        t0 = time.perf_counter()
        fn()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
t_all = 0.0
for _ in range(REPS):
    eng.step(1)
    eng.sync()
    t0 = time.perf_counter()
    eng.output_all()
    t_all += time.perf_counter() - t0
tot = 0.0
for k, v in acc.items():
    print("%-16s %8.1f us per call" % (k, v / REPS * 1e6))
    tot += v
print("%-16s %8.1f us per output_all  (%d markers, nx %d)" % ("sum", tot / REPS * 1e6, n, nx))
print("%-16s %8.1f us: pic1dp_hip_output_all, the three in one call and one wait" % ("output_all", t_all / REPS * 1e6))
