#!/usr/bin/env python3
"""trace_gaps.py -- kernel durations and the idle time between consecutive kernels from a
rocprofv3 --kernel-trace CSV (the last `tail` dispatches): where a short step's time goes.
    python tools/trace_gaps.py <kernel_trace.csv> [tail]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-tail:]
dur, gap_after = defaultdict(list), defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    m = re.search(r"(k_\w+|__amd\w+)", n)
    short = m.group(1) if m else n[:40]
    dur[short].append(e - s)
    if i + 1 < len(rows):
        gap_after[short].append(rows[i + 1][0] - e)
for k in dur:
    d, g = dur[k], gap_after.get(k, [0])
    print("%-42s n %4d  dur %8.2f us  gap after %6.2f us" % (k, len(d), sum(d) / len(d) / 1e3, sum(g) / len(g) / 1e3))
span = rows[-1][1] - rows[0][0]
print("span %.1f us, busy %.1f us" % (span / 1e3, sum(e - s for s, e, _ in rows) / 1e3))
