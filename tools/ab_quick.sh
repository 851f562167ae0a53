#!/bin/bash
# the default path at C3 / C2 / C1 / the strong-scaling share / C5, two runs each (whole-step path only)
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2; do
  echo "== run $r C3 1e8/1024   : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== run $r C2 1e7/256    : $(python tools/quick_bench.py 1e7 256 200 | grep 'mode 0')"
  echo "== run $r C1 6.4e6/192  : $(python tools/quick_bench.py 6.4e6 192 200 | grep 'mode 0')"
  echo "== run $r 1.25e7/1024   : $(python tools/quick_bench.py 1.25e7 1024 200 | grep 'mode 0')"
  echo "== run $r C5 1e8/4096   : $(PIC1DP_INPUT=$LANDAU python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
done
