#!/bin/bash
# diag_trace.sh [out dir] -- kernel durations (rocprofv3 --kernel-trace --stats) of forty steps at the reference's
# output cadence, the diagnostics taken each of the three ways (tools/diag_bench.py names them)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/diag_trace}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # tag FUSE
  export FUSE=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/diag_trace.py 1e8 1024 > /dev/null 2>&1
  echo "== $1"
  python3 - "$(find $OUT/t -name '*kernel_stats.csv' | head -n 1)" <<'PY'
import csv, sys, re
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if not re.search(r"k_step|k_ptcldist|k_field", n):
        continue
    short = re.sub(r"pic1dp::\(anonymous namespace\)::|void ", "", n)
    short = re.sub(r"\(.*", "", short)
    print("   %-58s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (short[:58], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                                    float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
  rm -rf $OUT/t
}
run "fusion 0: a pass of their own (k_ptcldist)" 0
run "fusion 2: k_step_full<DIAG> (rounds 2-4)" 2
