#!/usr/bin/env python3
"""stream_vs_size.py -- the pure-stream rates of the probe library against the size of the working set (64 MB ... 3.2 GB
of marker state): does a state that fits the 256 MB Infinity Cache stream faster than one that does not?
    python tools/stream_vs_size.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pic1dp_amd import probe
for n in (2e6, 4e6, 6.4e6, 1e7, 1.25e7, 2.5e7, 1e8):
    n = int(n)
    r = [probe.stream(4, 3, n, 20) for _ in range(3)]
    ro = [probe.stream(4, 0, n, 20) for _ in range(3)]
    t = [probe.layout(n, 12, 20)[1] for _ in range(3)]
    print("n %9d state %6.0f MB: 4r3w SoA %6.0f GB/s  4r %6.0f GB/s  tiled in-place 4r3w %.4f ms = %6.0f GB/s" % (n, n * 32 / 1e6, max(r), max(ro), min(t), n * 56 / min(t) / 1e6), flush=True)
