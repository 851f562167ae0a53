#!/bin/bash
# ab_variant.sh <variant.so>: the default library against a tuning build (PIC1DP_EXTRA_FLAGS=... PIC1DP_LIB_OUT=...)
export PIC1DP_QB_WARMUP=40
V=$1
for cfg in "1e8 1024" "1e7 256"; do for r in 1 2; do
  echo "== $cfg base    run $r: $(python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
  echo "== $cfg variant run $r: $(PIC1DP_LIB=$V python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
done; done
