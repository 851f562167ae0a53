#!/bin/bash
# ab_chain_now.sh -- the serial forward sums through the matrix unit with the next batch's LDS reads under way while a batch's
# dependent instructions run (device_field.hpp chain_rows_mfma, round 5) against the read-then-run form before it
# (pic1dp_amd/lib/v_prevchain.so: the library built from the commit before), alternating fresh processes.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
C5='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}'
for r in 1 2 3; do
  for v in prev new; do
    if [ $v = new ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$L/v_prevchain.so; fi
    echo "== run $r $v share 1.25e7/1024 : $(python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v C3 1e8/1024       : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v C5 1e8/4096       : $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v 2e5/1024          : $(python tools/quick_bench.py 2e5 1024 1000 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v landau 1.25e7/4096: $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1.25e7 4096 300 | grep 'mode 0' | cut -c1-110)"
  done
done
