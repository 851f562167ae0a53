#!/bin/bash
# ab_osub_dyn.sh -- with the drawn chunk tail, what is the oversubscribed grid still worth at C3?  PIC1DP_OSUB (grid in units
# of the resident one) x PIC1DP_DYN_TAIL (sixteenths drawn) x the solve in the prologue (possible on a resident grid only).
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for r in 1 2; do
  for o in 1 2 4; do
    for d in 8 12 16; do
      for f in 1 0; do
        [ $o != 1 ] && [ $f = 0 ] && continue
        echo "== run $r osub $o drawn $d/16 fuse $f : $(PIC1DP_OSUB=$o PIC1DP_DYN_TAIL=$d PIC1DP_FUSE_SOLVE=$f python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-110)"
      done
    done
  done
done
