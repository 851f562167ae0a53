export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
for r in 1 2 3; do
  echo "== run $r quick_bench C3          : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-140)"
  echo "== run $r bench.py events no fence : $(python bench.py --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["drop_in_call_sites"]["ms_per_step"])')"
  echo "== run $r bench.py events fenced   : $(PIC1DP_EVENT_FENCE=1 python bench.py --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["drop_in_call_sites"]["ms_per_step"])')"
done
