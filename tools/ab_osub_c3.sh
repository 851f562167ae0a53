#!/bin/bash
# grid size of the one-pass kernel in units of the resident grid, C3 and C5
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1 PIC1DP_FUSE_SOLVE=0
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2; do
for o in 1 2 3 4 6 8 12 16; do
echo "run $r C3 osub $o : $(PIC1DP_OSUB=$o q 1e8 1024 60)"
done
for o in 1 2 4; do
echo "run $r C5 osub $o : $(PIC1DP_OSUB=$o PIC1DP_INPUT=$LANDAU q 1e8 4096 60)"
done
done
