#!/bin/bash
# launch shapes of k_step_one on the headline (threads x workgroups per CU; more workgroups than the two
# resident ones = oversubscribed grid: the CUs that finish early take more of them)
export PIC1DP_QB_WARMUP=40
for sh in "0 0" "768 4" "768 6" "768 8" "768 12" "768 16" "768 32" "768 64" "0 0"; do set -- $sh
  echo "== bump nx1024 $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
for sh in "0 0" "768 4" "768 8" "768 16"; do set -- $sh
  echo "== bump 1.25e7 nx1024 $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 python tools/quick_bench.py 1.25e7 1024 100 | grep 'mode 0')"
  echo "== bump 1e7 nx256 $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 python tools/quick_bench.py 1e7 256 100 | grep 'mode 0')"
done
