#!/bin/bash
# register budget of the whole-step kernels (-DPIC1DP_WAVES_PER_EU=n builds) against launch shapes
export PIC1DP_QB_WARMUP=40
L=$PWD/pic1dp_amd/lib
for r in 1 2; do
  echo "== w6 768x2  run $r: $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== w5 640x2  run $r: $(PIC1DP_LIB=$L/libpic1dp_hip_w5.so PIC1DP_THREADS=640 PIC1DP_BPC=2 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== w5 1024x1 run $r: $(PIC1DP_LIB=$L/libpic1dp_hip_w5.so PIC1DP_THREADS=1024 PIC1DP_BPC=1 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== w5 512x2  run $r: $(PIC1DP_LIB=$L/libpic1dp_hip_w5.so PIC1DP_THREADS=512 PIC1DP_BPC=2 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== w6 512x3  run $r: $(PIC1DP_THREADS=512 PIC1DP_BPC=3 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
