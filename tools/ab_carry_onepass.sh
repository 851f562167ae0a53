#!/bin/bash
# k_step_one with and without the carry of -f0'/f0 through memory, and the no-exp distributions
export PIC1DP_QB_WARMUP=40
for r in 1 2; do for c in 1 0; do
  echo "== default carry $c run $r: $(PIC1DP_CARRY=$c python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done; done
MX='{"iptcldist":0,"species_density":[1.0],"species_v0":[0.0]}'
TS='{"iptcldist":2,"species_density":[1.0],"species_v0":[3.0]}'
LIN='{"linear":1}'
FF='{"deltaf":0}'
for cfg in "$MX" "$TS" "$LIN" "$FF"; do
  echo "== $cfg one pass : $(PIC1DP_INPUT=$cfg python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
  echo "== $cfg two pass : $(PIC1DP_PREDICT=0 PIC1DP_INPUT=$cfg python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
