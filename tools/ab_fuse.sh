#!/bin/bash
# the field solve inside the marker launch (one launch per step: PIC1DP_FUSE_SOLVE=2 fuses whatever the grid) against a
# launch of its own (0), with the serial sums through the matrix unit
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//; s/mode 0: //'; }
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2 3; do
for f in 0 2; do
echo "run $r fuse $f share 1.25e7/1024        : $(PIC1DP_FUSE_SOLVE=$f q 1.25e7 1024 200)"
echo "run $r fuse $f share 1.25e7/1024 npe 8  : $(PIC1DP_NPE=8 PIC1DP_FUSE_SOLVE=$f q 1.25e7 1024 200)"
echo "run $r fuse $f C1 6.4e6/192             : $(PIC1DP_FUSE_SOLVE=$f q 6.4e6 192 200)"
echo "run $r fuse $f C2 1e7/256 osub 1        : $(PIC1DP_OSUB=1 PIC1DP_FUSE_SOLVE=$f q 1e7 256 200)"
echo "run $r fuse $f C5 1e8/4096              : $(PIC1DP_INPUT=$LANDAU PIC1DP_FUSE_SOLVE=$f q 1e8 4096 60)"
echo "run $r fuse $f 2.5e7/512 (C4 share)     : $(PIC1DP_FUSE_SOLVE=$f q 2.5e7 512 100)"
done; done
