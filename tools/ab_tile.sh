#!/bin/bash
# tile size of the marker storage (kernels.hpp TILE_LOG2): whole-step kernels at [markers] [nx]
# for the variants built with
#   PIC1DP_EXTRA_FLAGS=-DPIC1DP_TILE_LOG2=$lt PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/libpic1dp_hip_t$lt.so python pic1dp_amd/build.py --force
N=${1:-1e8}; NX=${2:-1024}
export PIC1DP_QB_WARMUP=40
for r in 1 2; do
  for lt in 10 11 12 13 14; do
    lib=$PWD/pic1dp_amd/lib/libpic1dp_hip_t$lt.so
    [ $lt = 12 ] && lib=$PWD/pic1dp_amd/lib/libpic1dp_hip.so
    echo "== tile 2^$lt run $r: $(PIC1DP_LIB=$lib python tools/quick_bench.py $N $NX 40 | grep -E 'mode 0')"
  done
done
