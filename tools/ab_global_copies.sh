#!/bin/bash
# copies of the species charge accumulators in memory (GridConst::gcopies): flush contention
export PIC1DP_QB_WARMUP=40
for cfg in "6.4e6 192" "1.25e7 1024" "1e7 256" "1e8 1024"; do for r in 1 2; do for g in 1 8 32; do
  echo "== $cfg global copies $g run $r: $(PIC1DP_RHO_GLOBAL_COPIES=$g python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
done; done; done
