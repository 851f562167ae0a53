#!/bin/bash
# deposit collisions: the per-workgroup LDS rho tile in 1, 2, 4, 8 copies
# (lane l deposits into copy l % copies, device_math.hpp my_rho_copy) at the grid sizes of C1, C2, C3
export PIC1DP_QB_WARMUP=40
for cfg in "6.4e6 192" "1e7 256" "1e8 1024"; do
  for r in 1 2; do
    for k in 1 2 4 8; do
      echo "== markers/nx $cfg copies $k run $r: $(PIC1DP_RHO_COPIES=$k python tools/quick_bench.py $cfg 60 | grep -E 'mode 0')"
    done
  done
done
