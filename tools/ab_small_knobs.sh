#!/bin/bash
# the strong-scaling share (1.25e7 markers, nx 1024) and C2 (1e7, nx 256) under the launch / access knobs the library has
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
for r in 1 2; do
for cfg in "1.25e7 1024" "1e7 256"; do
echo "run $r $cfg default            : $(q $cfg 200)"
echo "run $r $cfg unfused            : $(PIC1DP_FUSE_SOLVE=0 q $cfg 200)"
echo "run $r $cfg nt off             : $(PIC1DP_NT_FORCE=0 q $cfg 200)"
echo "run $r $cfg nt on              : $(PIC1DP_NT_FORCE=1 q $cfg 200)"
echo "run $r $cfg rho global copies 8: $(PIC1DP_RHO_GLOBAL_COPIES=8 q $cfg 200)"
echo "run $r $cfg rho global copies 1: $(PIC1DP_RHO_GLOBAL_COPIES=1 q $cfg 200)"
echo "run $r $cfg osub 2             : $(PIC1DP_OSUB=2 q $cfg 200)"
echo "run $r $cfg osub 1             : $(PIC1DP_OSUB=1 q $cfg 200)"
echo "run $r $cfg tiles              : $(PIC1DP_PRED_KIND=1 q $cfg 200)"
echo "run $r $cfg register sums      : $(PIC1DP_PRED_KIND=3 q $cfg 200)"
done; done
