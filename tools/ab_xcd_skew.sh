#!/bin/bash
# (the tuning code this drives was taken out of the kernel after the experiment: check out commit d5b7589 to run it again)
# is the kernel held up by the slower (odd) XCDs?  The tuning build -DPIC1DP_TUNE_XCD_SKEW splits the pairs in eight
# pools, one per XCD, of 1 +- skew of an equal share (PIC1DP_XCD_SKEW; odd XCDs take less):
#   PIC1DP_EXTRA_FLAGS=-DPIC1DP_TUNE_XCD_SKEW PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/v_skew.so python pic1dp_amd/build.py --force
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1 PIC1DP_FUSE_SOLVE=0
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
for r in 1 2; do
echo "run $r default build            : $(q 1e8 1024 60)"
for s in 0 0.01 0.02 0.03 0.04 -0.02; do
echo "run $r pools, skew $s osub auto : $(PIC1DP_LIB=$L/v_skew.so PIC1DP_XCD_SKEW=$s q 1e8 1024 60)"
done
for s in 0 0.02 0.03; do
echo "run $r pools, skew $s osub 8    : $(PIC1DP_LIB=$L/v_skew.so PIC1DP_OSUB=8 PIC1DP_XCD_SKEW=$s q 1e8 1024 60)"
echo "run $r pools, skew $s osub 1    : $(PIC1DP_LIB=$L/v_skew.so PIC1DP_OSUB=1 PIC1DP_XCD_SKEW=$s q 1e8 1024 60)"
done
done
