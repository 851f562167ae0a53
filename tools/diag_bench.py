#!/usr/bin/env python3
"""diag_bench.py -- what an output_all costs at the reference's cadence (ten steps, then output_all;
src/pic1dp.F90:98-108, src/pic1dp_input.F90:109,250): wall clock of blocks of {10 steps + output_scalars + ptcldist}
against blocks of {10 steps}, for the three ways the diagnostics can be taken:
    fusion 0  a pass of their own (k_ptcldist) after an ordinary one-pass step; the prediction survives it
    fusion 1  where it pays: as fusion 0 on a predicted one-pass step, inside k_step_full<DIAG> on a two-pass step
    fusion 2  always inside the step: k_step_full<DIAG> -- rounds 2-4's form, which costs a predicted step its
              prediction (the step after the output runs a first-sub-step pass again)
(round 5 also measured the diagnostics inside the predicted kernel itself, k_step_sums<DIAG>: 2.10 ms against 1.03 +
1.14 ms for k_step_one and k_ptcldist apart under rocprofv3 -- no gain, not kept: profiles/r05/experiments/diag_trace.log)
    python tools/diag_bench.py [markers] [nx]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
eng.step(60)
eng.sync()
REPS = 10


def block(with_output):
    eng.set_time(0, 0.0)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(REPS):
        eng.step(10)
        if with_output:
            eng.output_scalars()
            eng.ptcldist(0)
        else:
            eng.sync()
    eng.sync()
    return (time.perf_counter() - t0) / REPS * 1e3


print("%d markers, nx %d" % (n, nx), flush=True)
eng.set_output_fusion(0)
block(False)
bare = sorted(block(False) for _ in range(3))[1]
print("10 steps                                   %.3f ms" % bare, flush=True)
eng.kernel_stats_enable(True)
for fuse in (0, 1, 2, 0, 1, 2):
    eng.set_output_fusion(fuse)
    block(True)
    k0 = [eng.kernel_stats(k)[1] for k in (3, 4, 5, 6)]
    t = sorted(block(True) for _ in range(3))[1]
    k1 = [eng.kernel_stats(k)[1] for k in (3, 4, 5, 6)]
    d = [(b - a) / (3 * REPS) for a, b in zip(k0, k1)]
    print("10 steps + output_all, fusion %d            %.3f ms  = 10 steps %+.3f ms | per block: k_step_half %.1f  k_step_full %.1f  "
          "k_ptcldist %.1f  k_step_one / k_step_sums %.1f" % (fuse, t, t - bare, d[0], d[1], d[2], d[3]), flush=True)
