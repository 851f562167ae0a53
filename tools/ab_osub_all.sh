#!/bin/bash
# grid size of the one-pass kernel in units of the resident grid, all configurations, three runs each
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//; s/mode 0: //'; }
for r in 1 2 3; do
for o in 4 6 8; do echo "run $r C3 osub $o    : $(PIC1DP_OSUB=$o q 1e8 1024 60)"; done
for o in 1 2 3 4; do echo "run $r C2 osub $o    : $(PIC1DP_OSUB=$o q 1e7 256 200)"; done
for o in 1 2 3; do echo "run $r C1 osub $o    : $(PIC1DP_OSUB=$o q 6.4e6 192 200)"; done
for o in 1 2; do echo "run $r share osub $o : $(PIC1DP_OSUB=$o q 1.25e7 1024 200)"; done
done
