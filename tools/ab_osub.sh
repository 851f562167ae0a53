#!/bin/bash
# grid size of the marker kernels in units of the resident grid (PIC1DP_OSUB; 0 = the library's rule)
export PIC1DP_QB_WARMUP=40
for cfg in "1e8 1024" "2.5e7 512" "1.25e7 1024" "1e7 256" "6.4e6 192"; do
  for o in 0 1 2 3 4 6; do
    echo "== one-pass $cfg osub $o: $(PIC1DP_OSUB=$o python tools/quick_bench.py $cfg 100 | grep 'mode 0')"
  done
done
for cfg in "1e8 1024" "1e7 256"; do
  for o in 0 1 2 4; do
    echo "== two-pass $cfg osub $o: $(PIC1DP_PREDICT=0 PIC1DP_OSUB=$o python tools/quick_bench.py $cfg 100 | grep 'mode 0')"
  done
done
C5='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}'
for o in 0 1 2 4; do
  echo "== landau 1e8 4096 osub $o: $(PIC1DP_OSUB=$o PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== landau 1e8 1024 osub $o: $(PIC1DP_OSUB=$o PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
