#!/bin/bash
# stamp_solve_probe.sh [out dir] -- the solve in the prologue, stage by stage (tools/stamp_solve_report.py), on a build
#   PIC1DP_EXTRA_FLAGS="-DPIC1DP_TUNING -DPIC1DP_TUNE_STAMPS -DPIC1DP_TUNE_STAMPS_SOLVE" PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/v_stamps.so python pic1dp_amd/build.py --force
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/stamps_solve}
mkdir -p "$OUT"
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for cfg in "2e5 1024 n2e5_nx1024" "1.25e7 1024 share_1.25e7_nx1024" "6.4e6 192 c1_6.4e6_nx192" "1e7 256 c2_1e7_nx256"; do
  set -- $cfg
  PIC1DP_OSUB=1 PIC1DP_LIB=$R/pic1dp_amd/lib/v_stamps.so PIC1DP_STAMP_AT=150 PIC1DP_STAMP_FILE=$OUT/$3.txt \
    python $R/tools/quick_bench.py $1 $2 200 | grep 'mode 0' | sed "s/^/## $3 : /"
  python $R/tools/stamp_solve_report.py $OUT/$3.txt $3
done
