export PIC1DP_QB_WARMUP=40
for i in 1 2 3 4 5 6; do echo "fresh process $i: $(python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"; done
for cfg in "1e8 1024" "1e7 256" "6.4e6 192"; do for r in 1 2; do
  echo "flush    $cfg: $(python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
  echo "NO flush $cfg: $(PIC1DP_DEBUG_NOFLUSH=1 python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
done; done
