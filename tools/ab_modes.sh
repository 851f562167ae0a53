#!/bin/bash
# one pass per step with 2, 3 and 4 kept modes (prediction tiles) against the two-pass path (PIC1DP_PREDICT=0)
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
for r in 1 2; do
for nm in 2 3 4; do
  M='{"nmode":'$nm',"modes":['$(seq -s, 1 $nm)']}'
  for cfg in "1e8 1024" "1e8 512" "1e7 256"; do
    echo "run $r nmode $nm $cfg one pass : $(PIC1DP_INPUT=$M q $cfg 40)"
    echo "run $r nmode $nm $cfg two pass : $(PIC1DP_INPUT=$M PIC1DP_PREDICT=0 q $cfg 40)"
  done
done; done
