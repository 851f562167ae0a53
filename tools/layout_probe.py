#!/usr/bin/env python3
"""layout_probe.py -- is the placement effect on k_step_full an interference between
its seven concurrent streams (4 read, 3 written in place)?  Times that traffic shape
over fresh slabs, arrays apart (SoA) against interleaved in tiles, while earlier slabs
stay allocated so that every slab lands in other physical memory."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PIC1DP_PLACEMENT_TRIES"] = "1"
import pic1dp_amd  # noqa: E402
from pic1dp_amd._lib import check  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=1000, nx=1024))
ms = (C.c_double * 2)()


def probe(lt, stagger, keep):
    check(eng.L.pic1dp_hip_debug_layout_probe(eng._ctx, n, lt, stagger, 20, keep, ms))
    return ms[0], ms[1]


probe(10, 0, 0)
for keep in (0, 1):
    for rnd in range(8 if keep else 3):
        for lt in (8, 10, 12, 14):
            a, b = probe(lt, 0, keep if lt == 14 else 0)
            gb = 56.0 * n / 1e6
            print("LAYOUT keep %d round %d tile 2^%-2d : SoA %.4f ms (%.0f GB/s)  tiled %.4f ms (%.0f GB/s)"
                  % (keep, rnd, lt, a, gb / a, b, gb / b), flush=True)
eng.close()
