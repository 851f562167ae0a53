#!/usr/bin/env python3
"""probe_sweep.py -- streaming-bandwidth probe over launch shapes and variants"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pic1dp_amd import probe  # noqa: E402  (libpic1dp_probe.so)
n = 10**8
NUM_CU = 256
for variant in (0, 1, 2):
    for thr, bpc in ((256, 1), (256, 2), (256, 4), (256, 8), (512, 1), (512, 2), (512, 4), (1024, 1), (1024, 2)):
        r = ["%d/%d: %5.0f" % (nr, nw, probe.stream(nr, nw, n, 5, blocks=NUM_CU * bpc, threads=thr, variant=variant))
             for nr, nw in ((1, 1), (4, 0), (4, 3), (7, 3))]
        print("variant %d threads %4d bpc %d  GB/s  %s" % (variant, thr, bpc, "  ".join(r)), flush=True)
