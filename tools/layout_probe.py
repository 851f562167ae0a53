#!/usr/bin/env python3
"""layout_probe.py -- is the placement effect on k_step_full an interference between
its seven concurrent streams (4 read, 3 written in place)?  Times that traffic shape
over fresh slabs, arrays apart (SoA) against interleaved in tiles, while earlier slabs
stay allocated so that every slab lands in other physical memory."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pic1dp_amd import probe as probe_lib  # noqa: E402  (libpic1dp_probe.so: measurement code, not the product)

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8


def probe(lt, stagger, keep):
    return probe_lib.layout(n, lt, 20, stagger_bytes=stagger, keep=keep)


probe(10, 0, 0)
rw, ro = 56.0 * n / 1e6, 32.0 * n / 1e6
for keep in (0, 1):
    for rnd in range(4 if keep else 2):
        for lt in (10, 11, 12, 13, 14):
            m = probe(lt, 0, keep if lt == 14 else 0)
            print("LAYOUT keep %d round %d tile 2^%-2d : r/w  SoA %.4f ms (%4.0f GB/s) tiled %.4f (%4.0f) tiled-wg %.4f (%4.0f)"
                  " | read-only  SoA %.4f (%4.0f) tiled %.4f (%4.0f) tiled-wg %.4f (%4.0f)"
                  % (keep, rnd, lt, m[0], rw / m[0], m[1], rw / m[1], m[4], rw / m[4], m[2], ro / m[2], m[3], ro / m[3],
                     m[5], ro / m[5]), flush=True)
probe_lib.release()
