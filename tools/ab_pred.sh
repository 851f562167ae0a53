#!/bin/bash
# HISTORICAL (the commit that introduced k_step_one): cost of k_step_one against k_step_full through a
# measurement hook of that commit (PIC1DP_DEBUG_PRED: 1 no carry, 2 carry written, 3 carry read + written),
# log in profiles/r02/experiments/ab_pred1.log.  Today: PIC1DP_PREDICT=0/1 and PIC1DP_CARRY=0/1, see
# tools/ab_carry_onepass.sh.
export PIC1DP_QB_WARMUP=40
for cfg in "1e8 1024" "1e7 256" "6.4e6 192"; do for r in 1 2; do for d in 0 1 3; do
  echo "== $cfg debug_pred $d run $r: $(PIC1DP_DEBUG_PRED=$d python tools/quick_bench.py $cfg 40 | grep 'mode 0')"
done; done; done
