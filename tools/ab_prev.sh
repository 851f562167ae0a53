#!/bin/bash
# the working tree's library against pic1dp_amd/lib/libpic1dp_hip_prev.so (built from the previous commit)
export PIC1DP_QB_WARMUP=40
L=$PWD/pic1dp_amd/lib
for cfg in "1e8 1024" "1e7 256" "6.4e6 192"; do for r in 1 2; do
  echo "== $cfg prev run $r: $(PIC1DP_LIB=$L/libpic1dp_hip_prev.so python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
  echo "== $cfg new  run $r: $(python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
done; done
NU='{"iptcldist":3,"species_temperature":[1.3],"species_temperature2":[0.7],"species_mass":[1.1],"species_density":[0.85],"species_v0":[4.5]}'
echo "== bump general prev: $(PIC1DP_LIB=$L/libpic1dp_hip_prev.so PIC1DP_INPUT=$NU python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"
echo "== bump general new : $(PIC1DP_INPUT=$NU python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"
