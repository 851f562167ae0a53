#!/bin/bash
# the working tree's library against pic1dp_amd/lib/libpic1dp_hip_prev.so (built from the commit before) at every configuration
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$PWD/pic1dp_amd/lib
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//; s/mode 0: //'; }
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for r in 1 2 3; do
for cfg in "1e8 1024 60" "1e7 256 200" "6.4e6 192 200" "1.25e7 1024 200"; do
  echo "run $r $cfg prev: $(PIC1DP_LIB=$L/libpic1dp_hip_prev.so q $cfg)"
  echo "run $r $cfg new : $(q $cfg)"
done
done
