#!/bin/bash
# collect_profiles.sh -- the rocprofv3 evidence of profiles/ in one go (run on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r03 [c3|c5]'
# Leaves everything under gpurun_out/<tag>/; copy the summaries into profiles/<tag>/ afterwards
# (cp).  Counter passes are separate runs with --kernel-trace only.
set -e -o pipefail
TAG=${1:-r03}
CFG=${2:-c3}
NX=1024
[ "$CFG" = c5 ] && NX=4096
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

echo "[1/4] default bench line" && date
python3 "$R/bench.py" --config $CFG > "$OUT/bench_${CFG}_wholestep.json" 2> "$OUT/bench_${CFG}_wholestep.err"
cat "$OUT/bench_${CFG}_wholestep.json"

echo "[2/4] kernel trace + stats" && date
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
  python3 "$R/bench.py" --config $CFG --steps 300 --warmup 40 --no-cpu-baseline --no-traffic-pass \
  > "$OUT/bench_${CFG}_wholestep_under_rocprof.json" 2> "$OUT/stats.err"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -n 1)" "$OUT/bench_${CFG}_wholestep_kernel_stats.csv"

for C in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $C | tr 'A-Z' 'a-z')
  echo "[pmc] $C" && date
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$lc" -- \
    python3 "$R/bench.py" --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-traffic-pass > "$OUT/bench_pmc_$lc.json" 2> "$OUT/pmc_$lc.err"
  # keep the particle and field kernels only (torch's own fill kernels are noise)
  f=$(find "$OUT/pmc_$lc" -name '*counter_collection.csv' | head -n 1)
  (head -n 1 "$f"; grep -E 'k_step|k_push|k_deposit|k_field|k_charge|k_pred' "$f" || true) > "$OUT/bench_${CFG}_wholestep_pmc_$lc.csv"
done
python3 "$R/profiles/summarize_pmc.py" "$OUT/bench_${CFG}_wholestep_pmc_fetch_size.csv" \
  "$OUT/bench_${CFG}_wholestep_pmc_write_size.csv" 1e8 $NX "$OUT/traffic_${CFG}_wholestep.json"
rm -rf "$OUT/stats" "$OUT/pmc_fetch_size" "$OUT/pmc_write_size"
echo done && date
