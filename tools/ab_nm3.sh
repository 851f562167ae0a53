#!/bin/bash
# ab_nm3.sh -- round 6: THREE kept modes as one pass with fixed-point prediction tiles (a build with -DPIC1DP_PRED_MAX_MODES=3,
# pic1dp_amd/lib/v_nm3.so) against the product's two passes; round 4 had measured 1.49 (tiles in doubles) against 1.45 ms.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
q() { python tools/quick_bench.py "$@" | grep 'mode 0' | sed 's/| with the events.*//'; }
M='{"nmode":3,"modes":[1,2,3]}'
PIC1DP_LIB=$L/v_nm3.so python - <<'PY'
import numpy as np, os, pic1dp_amd
kw = dict(nparticle_max=400001, nx=128, nmode=3, modes=[1, 2, 5], init_nmode=3, init_mode=[1, 2, 5], init_mode_cos=[0.0, 2e-6, 1e-6], init_mode_sin=[1e-5, 0.0, 3e-6])
def run(predict):
    os.environ["PIC1DP_PREDICT"] = predict
    e = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(**kw)); e.particle_load(); e.interaction_collect_charge(); e.field_solve_electric(); e.step(60)
    return e.predict_kind(), e.energy_history()
ka, ea = run("1"); kb, eb = run("0")
print("check: three kept modes, one pass (kind %d) against two passes (kind %d): max relative difference of int E^2 dx over 60 steps %.3g" % (ka, kb, np.max(np.abs(ea / eb - 1))))
PY
for r in 1 2 3; do
  for v in two_passes one_pass_fx; do
    if [ $v = two_passes ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$L/v_nm3.so; fi
    for cfg in "1e8 1024" "1e8 512" "1e7 256"; do
      echo "run $r $v nmode 3 $cfg : $(PIC1DP_INPUT=$M q $cfg 40)"
    done
  done
done
