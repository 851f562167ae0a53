#!/usr/bin/env python3
"""stamp_report.py -- where the time of ONE launch of a whole-step marker kernel goes, from the per-workgroup wall-clock
stamps a -DPIC1DP_TUNE_STAMPS build writes (tools/stamp_probe.sh): launch skew, prologue (tile staging), marker loop,
tail imbalance, rho flush, the six sums' reduction.
    python tools/stamp_report.py <stamp file> [label]"""
import sys

import numpy as np

fn = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else fn
head = open(fn).readline().split()
blocks, threads = int(head[2]), int(head[4])
a = np.loadtxt(fn, dtype=np.uint64, comments="#").reshape(-1, 7)
t = a[:, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0  # 100 MHz wall clock
hw = a[:, 6]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
hwid = hw & np.uint64(0xFFFFFFFF)
cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
cukey = ((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu
ncu = len(np.unique(cukey))


def q(x):
    return "min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (x.min(), np.percentile(x, 50), np.percentile(x, 90), x.max())


end = us[:, 5].max()
print("== %s: %d workgroups x %d threads on %d CUs, kernel (first entry -> last exit) %.2f us" % (label, blocks, threads, ncu, end))
first = us[:, 0] < us[:, 3].min()          # workgroups that entered before any finished its loop: the resident wave
print("  entry (launch skew), first wave %4d WGs : %s" % (first.sum(), q(us[first, 0])))
if (~first).any():
    print("  entry, later workgroups   %4d WGs      : %s" % ((~first).sum(), q(us[~first, 0])))
print("  prologue (entry -> tiles staged)         : %s" % q(us[:, 1] - us[:, 0]))
print("  marker loop (staged -> workgroup done)   : %s" % q(us[:, 3] - us[:, 1]))
print("  thread 0 done -> workgroup done          : %s" % q(us[:, 3] - us[:, 2]))
print("  rho flush                                : %s" % q(us[:, 4] - us[:, 3]))
print("  sums / prediction flush                  : %s" % q(us[:, 5] - us[:, 4]))
print("  exit time                                : %s" % q(us[:, 5]))
# per CU: when its last workgroup left; the kernel ends with the slowest CU
last_by_cu = {}
busy_by_cu = {}
for k, e, b0, b1 in zip(cukey, us[:, 5], us[:, 1], us[:, 3]):
    last_by_cu[k] = max(last_by_cu.get(k, 0.0), e)
    busy_by_cu[k] = busy_by_cu.get(k, 0.0) + (b1 - b0)
lb = np.array(list(last_by_cu.values()))
print("  last exit per CU                         : %s   (idle at the end, mean over CUs: %.2f us)" % (q(lb), (end - lb).mean()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("    XCC %d: %4d WGs  loop mean %7.2f  last exit %7.2f" % (x, m.sum(), (us[m, 3] - us[m, 1]).mean(), us[m, 5].max()))
