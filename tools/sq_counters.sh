#!/bin/bash
# sq_counters.sh <tag> [PIC1DP_INPUT json] -- SQ counters of the whole-step kernels (rocprofv3 --pmc, two
# passes of <= 8 counters, --kernel-trace only) on tools/quick_bench.py 1e8 1024 4; means per dispatch
# into gpurun_out/<tag>_sq.json
set -e -o pipefail
TAG=${1:-sq}
export PIC1DP_INPUT=${2:-'{}'}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
i=0
export PIC1DP_QB_ONLY_STEP=1
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/quick_bench.py" 1e8 1024 4 > "$OUT/pass$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for name in ("k_step_half", "k_step_full", "k_step_one"):
            if name in k:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + "_sq.json", "w"), indent=1)
for k, d in res.items():
    np_ = 1e8
    print(k, "VALU wave-instr per marker %.1f" % (d.get("SQ_INSTS_VALU", 0) * 64 / np_ / 1.0),
          "VALU busy quad-cycles %.3g" % d.get("SQ_ACTIVE_INST_VALU", 0), {c: "%.4g" % v for c, v in d.items()})
PY
