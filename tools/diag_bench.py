#!/usr/bin/env python3
"""diag_bench.py -- cost of the output_all diagnostics (N1) next to a time step"""
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
for name, fn in (("ptcldist", lambda: eng.ptcldist(0)), ("output_scalars", eng.output_scalars),
                 ("energy_sums", eng.energy_sums), ("step", lambda: (eng.step(1), eng.sync()))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    print("%-15s %.3f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
