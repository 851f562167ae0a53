export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for r in 1 2 3; do
  for g in 8 4 2 1; do
    export PIC1DP_RHO_GLOBAL_COPIES=$g
    echo "== run $r copies $g C1 6.4e6/192 : $(python tools/quick_bench.py 6.4e6 192 400 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r copies $g C2 1e7/256   : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
  done
done
