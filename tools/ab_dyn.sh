#!/bin/bash
# the one-pass kernels' chunk schedule: everything dealt round-robin (PIC1DP_DYN_FRAC=0) against a drawn last share
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2; do
for f in ${FRACS:-0 0.1 0.2 0.3}; do
export PIC1DP_DYN_FRAC=$f
echo "run $r frac $f C3 osub auto : $(q 1e8 1024 60)"
echo "run $r frac $f C3 osub 1    : $(PIC1DP_OSUB=1 q 1e8 1024 60)"
echo "run $r frac $f C3 osub 2    : $(PIC1DP_OSUB=2 q 1e8 1024 60)"
echo "run $r frac $f share        : $(q 1.25e7 1024 200)"
echo "run $r frac $f C1           : $(q 6.4e6 192 200)"
echo "run $r frac $f C2 osub auto : $(q 1e7 256 200)"
echo "run $r frac $f C2 osub 1    : $(PIC1DP_OSUB=1 q 1e7 256 200)"
echo "run $r frac $f C5           : $(PIC1DP_INPUT=$LANDAU q 1e8 4096 60)"
done; done
