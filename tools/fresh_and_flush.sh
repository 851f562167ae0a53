export PIC1DP_QB_WARMUP=40
for i in 1 2 3 4 5 6; do echo "fresh process $i: $(python tools/quick_bench.py 1e8 1024 40 | grep 'mode 0')"; done
for cfg in "1e8 1024" "1e7 256" "6.4e6 192"; do for r in 1 2; do
  echo "flush    $cfg: $(python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
  # a tuning build: PIC1DP_EXTRA_FLAGS=-DPIC1DP_TUNE_NOFLUSH PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/v_noflush.so python pic1dp_amd/build.py --force
  echo "NO flush $cfg: $(PIC1DP_LIB=$PWD/pic1dp_amd/lib/v_noflush.so python tools/quick_bench.py $cfg 60 | grep 'mode 0')"
done; done
