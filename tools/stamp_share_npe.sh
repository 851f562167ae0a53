#!/bin/bash
# stamp_share_npe.sh [out dir] -- round 6: where the 8-way strong-scaling share (1.25e7 markers, nx 1024)
# spends its launch, with the solve in the prologue summed in the ONE-rank order (1024 dependent additions through the matrix
# unit) and in the EIGHT-rank order the 8-GPU run reproduces (eight partial chains of 128 side by side): stamps of the
# prologue's stages and of the whole launch.  Needs pic1dp_amd/lib/v_stamps.so
# (-DPIC1DP_TUNING -DPIC1DP_TUNE_STAMPS -DPIC1DP_TUNE_STAMPS_SOLVE).
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/stamps_share}
mkdir -p "$OUT"
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for npe in 1 8; do
  tag=share_1.25e7_nx1024_npe$npe
  echo "## $tag product build : $(PIC1DP_NPE=$npe python $R/tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0')"
  PIC1DP_NPE=$npe PIC1DP_LIB=$R/pic1dp_amd/lib/v_stamps.so PIC1DP_STAMP_AT=150 PIC1DP_STAMP_FILE=$OUT/$tag.txt \
    python $R/tools/quick_bench.py 1.25e7 1024 200 | grep 'mode 0' | sed "s/^/## $tag stamps build  : /"
  python $R/tools/stamp_solve_report.py $OUT/$tag.txt $tag
done
