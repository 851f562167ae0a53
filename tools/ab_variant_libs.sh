#!/bin/bash
# A/B of tuning builds (PIC1DP_EXTRA_FLAGS ... PIC1DP_LIB_OUT=pic1dp_amd/lib/v_<name>.so python pic1dp_amd/build.py --force)
# against the default library at C3, alternating fresh processes:  tools/ab_variant_libs.sh name1 name2 ...
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
for r in 1 2 3; do
  echo "run $r default : $(python tools/quick_bench.py ${N:-1e8} ${NX:-1024} 60 | grep 'mode 0')"
  for v in "$@"; do
    echo "run $r $v : $(PIC1DP_LIB=$L/v_$v.so python tools/quick_bench.py ${N:-1e8} ${NX:-1024} 60 | grep 'mode 0')"
  done
done
