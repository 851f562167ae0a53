#!/bin/bash
# ab_fx_tiles.sh -- round 6: the prediction tiles (two kept modes; one kept mode when asked for) summed as
# 64-bit fixed-point numbers in the LDS (kernels_step.hip FxTiles) against the library before them (double sums:
# pic1dp_amd/lib/v_prev.so, built from a worktree of the commit before with PIC1DP_LIB_OUT), alternating fresh processes.
# The first version of this script timed a tuning build with fixed scales and no overflow handling
# (profiles/r06/experiments/ab_fx_tiles_timing_experiment.log): 1.235 -> 1.08 ms for two kept modes at 1e8 markers.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
M='{"nmode":2,"modes":[1,2]}'
for r in 1 2 3; do
  for v in double fixed; do
    if [ $v = fixed ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$L/v_prev.so; fi
    for cfg in "1e8 1024" "1e8 512" "1e7 256"; do
      echo "run $r $v nmode 2 $cfg : $(PIC1DP_INPUT=$M q $cfg 40)"
    done
    echo "run $r $v nmode 1 tiles 1e8 1024 : $(PIC1DP_PRED_KIND=1 q 1e8 1024 40)"
    echo "run $r $v nmode 1 sums (headline kernel) 1e8 1024 : $(q 1e8 1024 40)"
  done
done
