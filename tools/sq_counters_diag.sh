#!/bin/bash
# sq_counters_diag.sh <tag> -- SQ counters of the output_all diagnostics kernels (k_ptcldist, k_step_full<DIAG>)
# at 1e8 markers / nx 1024: rocprofv3 --pmc in two passes of <= 8 counters, --kernel-trace only, on
# tools/diag_bench.py; means per dispatch and per marker into gpurun_out/<tag>_sq_diag.json
set -e -o pipefail
TAG=${1:-sqdiag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/diag_bench.py" 1e8 > "$OUT/pass$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_ptcldist" in k:
            name = "k_ptcldist"
        elif "k_step_full" in k:
            name = "k_step_full<DIAG>" if re.search(r"true\s*>\s*\(", k.replace(" ", "")) or k.rstrip(") ").rstrip().endswith("true>") else "k_step_full"
        elif "k_step_one" in k:
            name = "k_step_one"
        elif "k_step_half" in k:
            name = "k_step_half"
        else:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    d["dispatches_seen"] = max(len(v) for v in acc[k].values())
json.dump(res, open(out + "_sq_diag.json", "w"), indent=1)
for k, d in res.items():
    n = 1e8
    print(k, "| per marker: VALU wave-instr %.1f  LDS wave-instr %.2f | LDS active cycles %.4g  bank-conflict cycles %.4g (%.2f per active)"
          % (d.get("SQ_INSTS_VALU", 0) * 64 / n, d.get("SQ_INSTS_LDS", 0) * 64 / n, d.get("SQ_ACTIVE_INST_LDS", 0),
             d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_ACTIVE_INST_LDS", 1), 1)))
PY
rm -rf "$OUT/pass1" "$OUT/pass2"
