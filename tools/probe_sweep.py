#!/usr/bin/env python3
"""probe_sweep.py -- streaming-bandwidth probe over launch shapes and variants"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402
n = 10**8
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=16, nx=1024))
for variant in (0, 1, 2):
    os.environ["PIC1DP_PROBE_VARIANT"] = str(variant)
    for thr, bpc in ((256, 1), (256, 2), (256, 4), (256, 8), (512, 1), (512, 2), (512, 4), (1024, 1), (1024, 2)):
        eng.set_launch(thr, bpc)
        r = ["%d/%d: %5.0f" % (nr, nw, eng.stream_probe(nr, nw, n, 5)) for nr, nw in ((1, 1), (4, 0), (4, 3), (7, 3))]
        print("variant %d threads %4d bpc %d  GB/s  %s" % (variant, thr, bpc, "  ".join(r)), flush=True)
