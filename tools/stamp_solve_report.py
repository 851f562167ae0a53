#!/usr/bin/env python3
"""stamp_solve_report.py -- the fused prologue's solve split into its dependent stages, from a build with
-DPIC1DP_TUNE_STAMPS -DPIC1DP_TUNE_STAMPS_SOLVE (stamps 2, 3, 4 taken inside fused_solve: kernels_step.hip):
    entry -> charge read, products staged (first barrier) -> serial forward sums done -> prediction's sums combined -> tiles staged
    python tools/stamp_solve_report.py <stamp file> [label]"""
import sys

import numpy as np

fn = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else fn
a = np.loadtxt(fn, dtype=np.uint64, comments="#").reshape(-1, 7)
t = a[:, :6].astype(np.int64)
us = (t - t[:, 0].min()) / 100.0


def q(x):
    return "min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (x.min(), np.percentile(x, 50), np.percentile(x, 90), x.max())


print("== %s: %d workgroups; the solve in the prologue, us" % (label, len(us)))
print("  entry -> charge and products staged     : %s" % q(us[:, 2] - us[:, 0]))
print("  -> serial forward sums done             : %s" % q(us[:, 3] - us[:, 2]))
print("  -> the prediction's sums combined       : %s" % q(us[:, 4] - us[:, 3]))
print("  -> both inverse transforms, tiles staged: %s" % q(us[:, 1] - us[:, 4]))
print("  prologue altogether                     : %s" % q(us[:, 1] - us[:, 0]))
