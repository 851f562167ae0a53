#!/bin/bash
# tiled (default build) against untiled (PIC1DP_TILE_LOG2=0 build) marker storage, alternating
# processes: kernels of the whole-step path at [markers] [nx].  Build the untiled variant first:
#   PIC1DP_EXTRA_FLAGS=-DPIC1DP_TILE_LOG2=0 PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/libpic1dp_hip_untiled.so python pic1dp_amd/build.py --force
N=${1:-1e8}; NX=${2:-1024}
export PIC1DP_QB_WARMUP=40
for r in 1 2 3; do
  echo "== tiled   run $r"; python tools/quick_bench.py $N $NX 40 | grep -E "mode 0|calls"
  echo "== untiled run $r"; PIC1DP_LIB=$PWD/pic1dp_amd/lib/libpic1dp_hip_untiled.so python tools/quick_bench.py $N $NX 40 | grep -E "mode 0|calls"
done
