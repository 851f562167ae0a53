#!/bin/bash
# ab_dyn_tail_other.sh -- the drawn chunk tail (PIC1DP_DYN_TAIL sixteenths, 0 = all dealt) in the OTHER whole-step kernels:
# k_step_half + k_step_full (two passes per step: PIC1DP_PREDICT=0) and k_step_one with the prediction as tiles (two kept
# modes; one kept mode on request), alternating fresh processes.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
M2='{"nmode":2,"modes":[1,2]}'
for r in 1 2; do
  for v in 0 8 16; do
    export PIC1DP_DYN_TAIL=$v
    echo "== run $r drawn $v/16 two passes C3 1e8/1024 : $(PIC1DP_PREDICT=0 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r drawn $v/16 two passes C2 1e7/256  : $(PIC1DP_PREDICT=0 python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r drawn $v/16 two passes 1.25e7/1024 : $(PIC1DP_PREDICT=0 python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r drawn $v/16 tiles nm 2 1e8/1024    : $(PIC1DP_INPUT=$M2 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r drawn $v/16 tiles nm 2 1e7/256     : $(PIC1DP_INPUT=$M2 python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r drawn $v/16 tiles nm 1 1e8/1024    : $(PIC1DP_PRED_KIND=1 python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-120)"
  done
done
