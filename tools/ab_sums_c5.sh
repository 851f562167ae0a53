#!/bin/bash
# BASELINE configs[4] per GPU (Landau damping: Maxwellian, 1e8 markers, nx = 4096): two passes per step
# against the one pass with the prediction as six sums (k_step_sums), and launch shapes of the latter;
# then bump-on-tail at the same grid
export PIC1DP_QB_WARMUP=30
C5='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}'
for r in 1 2; do
  echo "== landau two-pass        run $r: $(PIC1DP_PREDICT=0 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== landau sums default    run $r: $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== landau sums 768x1      run $r: $(PIC1DP_THREADS=768 PIC1DP_BPC=1 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== landau sums 512x1      run $r: $(PIC1DP_THREADS=512 PIC1DP_BPC=1 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== bump   two-pass        run $r: $(PIC1DP_PREDICT=0 python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
  echo "== bump   sums default    run $r: $(python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
done
# where both kernels can run (nx = 1024): tiles against sums, Maxwellian
for r in 1 2; do
  echo "== landau nx1024 tiles    run $r: $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== landau nx1024 sums     run $r: $(PIC1DP_PRED_KIND=2 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
