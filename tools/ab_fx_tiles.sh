#!/bin/bash
# ab_fx_tiles.sh -- round 6: two kept modes, the prediction tiles summed as 64-bit fixed-point numbers in
# the LDS (ds_add_u64: 4.5 ns per wave-instruction at random words against ds_add_f64's 8.7, tools/lds_atomic_rate.hip) --
# a TIMING experiment: tuning build -DPIC1DP_TUNING -DPIC1DP_TUNE_FXTILES (fixed scales, no overflow handling) as
# pic1dp_amd/lib/v_fx.so against the product library, alternating fresh processes.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
M='{"nmode":2,"modes":[1,2]}'
for r in 1 2 3; do
  for v in double fixed; do
    if [ $v = double ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$L/v_fx.so; fi
    for cfg in "1e8 1024" "1e8 512" "1e7 256"; do
      echo "run $r $v nmode 2 $cfg : $(PIC1DP_INPUT=$M q $cfg 40)"
    done
    echo "run $r $v nmode 1 tiles 1e8 1024 : $(PIC1DP_PRED_KIND=1 q 1e8 1024 40)"
  done
done
