#!/bin/bash
# duration of the field launch of a one-pass step (k_field_solve_pair1 / _pair_sums1) against the number of
# reference ranks whose summation order it reproduces (PIC1DP_NPE): rocprofv3 --kernel-trace --stats on
# tools/step_only.py at the per-GPU share of a strong-scaled 1e8 run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
LANDAU='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0],"lx":12.566370614359172}'
for nx in 1024 4096; do
  cfg='{}'; [ $nx = 4096 ] && cfg=$LANDAU
  for npe in 1 2 4 8; do
    rm -rf /tmp/flt
    PIC1DP_NPE=$npe PIC1DP_INPUT="$cfg" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/flt -- python3 $R/tools/step_only.py 1.25e7 $nx 300 > /tmp/flt.log 2>&1
    python3 - $nx $npe <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob("/tmp/flt/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
out = []
for r in rows:
    m = re.search(r"(k_field_solve\w*|k_step_one|k_step_sums|k_step_half)", r["Name"])
    if m:
        out.append("%s x%s %.2f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
print("nx %s npe %s: %s" % (sys.argv[1], sys.argv[2], " | ".join(sorted(out)) or "no kernel stats (see /tmp/flt.log)"))
PY
  done
done
