#!/bin/bash
# ds_add_f64-only deposit against the wave-level pre-reduction (-DPIC1DP_DEPOSIT_PREREDUCE=1 build:
#   PIC1DP_EXTRA_FLAGS=-DPIC1DP_DEPOSIT_PREREDUCE=1 PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/libpic1dp_hip_prereduce.so python pic1dp_amd/build.py --force)
# at the grid sizes of C1, C2, C3
export PIC1DP_QB_WARMUP=10
L=$PWD/pic1dp_amd/lib
for cfg in "6.4e6 192" "1e7 256" "1e8 1024"; do
  echo "== $cfg ds_add_f64 only : $(python tools/quick_bench.py $cfg 20 | grep 'mode 0')"
  echo "== $cfg pre-reduction   : $(PIC1DP_LIB=$L/libpic1dp_hip_prereduce.so python tools/quick_bench.py $cfg 20 | grep 'mode 0')"
done
