#!/bin/bash
# stamp_probe.sh [out dir] -- the fixed cost of the one-pass marker kernel, decomposed: per-workgroup wall-clock stamps of
# ONE launch (a -DPIC1DP_TUNING -DPIC1DP_TUNE_STAMPS build, pic1dp_amd/lib/v_stamps.so) at the small configurations and the strong-scaling
# share, next to the launch's duration by HIP events in the default build.
#   PIC1DP_EXTRA_FLAGS="-DPIC1DP_TUNING -DPIC1DP_TUNE_STAMPS" PIC1DP_LIB_OUT=pic1dp_amd/lib/v_stamps.so python pic1dp_amd/build.py --force
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/stamps}
mkdir -p "$OUT"
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
run() {  # markers nx tag [input json]
  local tag=$3
  echo "## $tag default build : $(PIC1DP_INPUT=${4:-'{}'} python $R/tools/quick_bench.py $1 $2 200 | grep 'mode 0')"
  PIC1DP_INPUT=${4:-'{}'} PIC1DP_LIB=$R/pic1dp_amd/lib/v_stamps.so PIC1DP_STAMP_AT=150 PIC1DP_STAMP_FILE=$OUT/$tag.txt \
    python $R/tools/quick_bench.py $1 $2 200 | grep 'mode 0' | sed "s/^/## $tag stamps build  : /"
  python $R/tools/stamp_report.py $OUT/$tag.txt $tag
}
run 4096 1024 n4096_nx1024
run 2e5 1024 n2e5_nx1024
run 1e6 1024 n1e6_nx1024
run 1.25e7 1024 share_1.25e7_nx1024
run 6.4e6 192 c1_6.4e6_nx192
run 1e7 256 c2_1e7_nx256
run 1e8 1024 c3_1e8_nx1024
run 1.25e7 4096 landau_1.25e7_nx4096 "$LANDAU"
