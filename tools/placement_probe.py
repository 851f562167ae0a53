#!/usr/bin/env python3
"""placement_probe.py -- what makes the whole-step kernels' time depend on where the
marker arrays lie?  (Needs the UNTILED build, -DPIC1DP_TILE_LOG2=0: PIC1DP_SLAB_STAGGER sets the
array-to-array distance there; layout 0 = four separate hipMallocs existed only in the commit that
ran experiment C.  Logs: profiles/r02/experiments/placement_probe{1,2}.log.)  All engines live in ONE process (placement effects only compare
inside one), each is loaded, warmed up and timed on the same physics.

  A. one slab, arrays k*stride apart with stride = 2 MiB multiple + stagger, for a
     list of staggers (twice, second time in reverse order: drift vs placement)
  B. the same stagger re-created several times while the previous slabs stay
     allocated (same relative offsets, different physical memory)
  C. four separate hipMallocs (the round-1 layout), addresses printed
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
nx = 1024


def measure(layout, stagger, keep=None, label=""):
    os.environ["PIC1DP_MARKER_LAYOUT"] = str(layout)
    os.environ["PIC1DP_SLAB_STAGGER"] = str(stagger)
    eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.step(40)
    eng.sync()
    eng.kernel_stats_enable(True)
    eng.step(30)
    eng.sync()
    (hm, hn), (fm, fn) = eng.kernel_stats(3), eng.kernel_stats(4)
    print("RESULT %s layout %d stagger %8d : half %.4f ms  full %.4f ms" % (label, layout, stagger, hm / hn, fm / fn), flush=True)
    if keep is not None:
        keep.append(eng)
    else:
        eng.close()
    return hm / hn, fm / fn


M2 = 2 << 20
if "--big" in sys.argv:     # array-to-array distance = (382 + k) * 2 MiB
    staggers = [k * M2 for k in (0, 1, 2, 3, 4, 5, 7, 8, 16, 17, 32, 64, 128, 130, 256, 650)]
else:
    staggers = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, 4096 + 256, (1 << 20) + 4096 + 256,
                (1 << 16) + (1 << 12) + (1 << 8)]
print("== A: slab staggers", flush=True)
for s in staggers:
    measure(1, s, label="A1")
for s in reversed(staggers):
    measure(1, s, label="A2")
if "--big" in sys.argv:
    sys.exit(0)
print("== B: same stagger, fresh physical memory each time (previous kept)", flush=True)
keep = []
for r in range(5):
    measure(1, 0, keep=keep, label="B0")
for e in keep:
    e.close()
keep = []
for r in range(5):
    measure(1, 4096 + 256, keep=keep, label="B1")
for e in keep:
    e.close()
print("== C: four hipMallocs", flush=True)
keep = []
for r in range(6):
    measure(0, 0, keep=keep, label="C")
for e in keep:
    e.close()
