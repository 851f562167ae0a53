#!/bin/bash
# the drawn tail in k_step_sums: C5 (Landau, nx 4096) and the default input at nx 2048 (bump-on-tail: two spilled registers)
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2; do
  for v in 0 8 16; do
    export PIC1DP_DYN_TAIL=$v
    echo "== run $r drawn $v/16  C5 1e8/4096 Landau : $(PIC1DP_INPUT=$LANDAU python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r drawn $v/16  bump 1e8/2048      : $(python tools/quick_bench.py 1e8 2048 60 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r drawn $v/16  Landau 1.25e7/4096 : $(PIC1DP_INPUT=$LANDAU python tools/quick_bench.py 1.25e7 4096 300 | grep 'mode 0' | cut -c1-110)"
  done
done
