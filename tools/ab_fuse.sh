#!/bin/bash
# the field solve inside the marker launch (one launch per step) against a launch of its own; sums against tiles at small grids
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0'; }
for r in 1 2; do
echo "C1 sums fused           : $(PIC1DP_PRED_KIND=2 q 6.4e6 192 400)"
echo "C1 sums unfused         : $(PIC1DP_PRED_KIND=2 PIC1DP_FUSE_SOLVE=0 q 6.4e6 192 400)"
echo "n2e5 sums fused         : $(PIC1DP_PRED_KIND=2 q 2e5 192 400)"
echo "n2e5 sums unfused       : $(PIC1DP_PRED_KIND=2 PIC1DP_FUSE_SOLVE=0 q 2e5 192 400)"
done
