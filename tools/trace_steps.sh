#!/bin/bash
# kernel durations and gaps of nothing but pic1dp_hip_step (rocprofv3 --kernel-trace of tools/step_only.py), the solve
# in a launch of its own and inside the marker launch:  bash tools/trace_steps.sh [out dir]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/trace_steps}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for fuse in 0 1; do
for cfg in "6.4e6 192" "1e7 256" "1.25e7 1024"; do
  set -- $cfg
  export PIC1DP_FUSE_SOLVE=$fuse
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/step_only.py $1 $2 300 > /dev/null 2>&1
  echo "== $1 markers, nx $2, PIC1DP_FUSE_SOLVE=$fuse"
  python3 $R/tools/trace_gaps.py $(find $OUT/t -name "*kernel_trace.csv") 400
  rm -rf $OUT/t
done; done
