#!/bin/bash
# ab_wave_prio.sh -- PIC1DP_WAVE_PRIO: the one-pass kernels' waves lower their issue priority (s_setprio 3 .. 0) with their
# progress, so that the two workgroups of a CU stay together (profiles/r05/experiments/stamps_drawn_tail.log: one leaves at
# ~85 us, the other at ~125 at the 8-way share).  Alternating fresh processes.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
C5='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}'
for r in 1 2 3; do
  for v in 0 1; do
    export PIC1DP_WAVE_PRIO=$v
    echo "== run $r prio $v  C1 6.4e6/192 : $(python tools/quick_bench.py 6.4e6 192 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r prio $v  C2 1e7/256   : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r prio $v  1.25e7/1024  : $(python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r prio $v  C3 1e8/1024  : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r prio $v  C5 1e8/4096  : $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0' | cut -c1-110)"
  done
done
