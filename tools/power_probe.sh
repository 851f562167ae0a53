#!/bin/bash
# power_probe.sh -- clocks and socket power while a kernel runs back to back: is it power-limited?
#   the headline (bump-on-tail, k_step_one), the Landau configuration (k_step_sums, no exp), a pure stream
sample() {
  sleep 6
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '
    echo
    sleep 1
  done
}
echo "== k_step_one, bump-on-tail, 1e8 markers, nx 1024"
python tools/step_only.py 1e8 1024 6000 & P=$!; sample; wait $P
echo "== k_step_sums, Landau, 1e8 markers, nx 4096"
PIC1DP_INPUT='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}' python tools/step_only.py 1e8 4096 8000 & P=$!; sample; wait $P
echo "== pure stream, 4 arrays read + 3 written"
python - <<'PY' & P=$!
import os, sys
sys.path.insert(0, os.getcwd())
from pic1dp_amd import probe
for _ in range(9):
    probe.stream(4, 3, 10**8, 1000)
PY
sample; wait $P
echo "== idle"
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
