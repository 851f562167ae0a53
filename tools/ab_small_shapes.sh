#!/bin/bash
# launch shapes of k_step_one where a workgroup's one-off costs (staging, flush atomics) weigh: the per-GPU
# share of a strong-scaled 1e8 run, C2, C1
export PIC1DP_QB_WARMUP=40
for cfg in "1.25e7 1024" "1e7 256" "6.4e6 192"; do
  for sh in "0 0" "1024 1" "768 1" "512 2" "512 1" "0 0"; do set -- $sh
    echo "== $cfg $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 python tools/quick_bench.py $cfg 200 | grep 'mode 0')"
  done
done
