export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1 PIC1DP_PREDICT=0
for r in 1 2 3; do
  for v in 8 16; do
    export PIC1DP_DYN_TAIL_FULL=$v
    echo "== run $r k_step_full drawn $v/16 C3 1e8/1024 : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r k_step_full drawn $v/16 C2 1e7/256  : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r k_step_full drawn $v/16 1.25e7/1024 : $(python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-120)"
    echo "== run $r k_step_full drawn $v/16 C1 6.4e6/192: $(python tools/quick_bench.py 6.4e6 192 300 | grep 'mode 0' | cut -c1-120)"
  done
done
