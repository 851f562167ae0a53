#!/bin/bash
# ab_dyn_tail.sh -- VERDICT r04 item 4: the last n/16 of a workgroup's 64-pair chunks DRAWN by its waves from an LDS counter
# (PIC1DP_DYN_TAIL=n, k_step_one<PRIV>) against all of them dealt (0), alternating fresh processes, at the small
# configurations, the strong-scaling share and C3
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for r in 1 2; do
  for v in 0 4 8 16; do
    export PIC1DP_DYN_TAIL=$v
    echo "== run $r drawn $v/16  C1 6.4e6/192 : $(python tools/quick_bench.py 6.4e6 192 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r drawn $v/16  C2 1e7/256   : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r drawn $v/16  1.25e7/1024  : $(python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r drawn $v/16  C3 1e8/1024  : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-110)"
  done
done
