#!/bin/bash
# (a) marker-loop variants of the whole-step kernels (PIC1DP_STEP_PIPE builds), default input
# (b) carry of -f0'/f0 between the kernels on/off for species with general divisor constants
export PIC1DP_QB_WARMUP=40
L=$PWD/pic1dp_amd/lib
for r in 1 2; do
  for v in "" _pipe1 _pipe2; do
    echo "== pipe '$v' run $r: $(PIC1DP_LIB=$L/libpic1dp_hip$v.so python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"
  done
done
for r in 1 2; do
  for v in "" _pipe2; do
    echo "== 6.4e6/192 pipe '$v' run $r: $(PIC1DP_LIB=$L/libpic1dp_hip$v.so python tools/quick_bench.py 6.4e6 192 100 | grep 'mode 0')"
  done
done
NU='{"iptcldist":3,"species_temperature":[1.3],"species_temperature2":[0.7],"species_mass":[1.1],"species_density":[0.85],"species_v0":[4.5]}'
TS='{"iptcldist":2,"species_density":[1.0],"species_v0":[3.0],"species_temperature":[0.9],"species_mass":[1.2]}'
for r in 1 2; do
  for c in 0 1; do
    echo "== bump non-unit carry $c run $r: $(PIC1DP_CARRY=$c PIC1DP_INPUT=$NU python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"
    echo "== two-stream non-unit carry $c run $r: $(PIC1DP_CARRY=$c PIC1DP_INPUT=$TS python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"
  done
done
