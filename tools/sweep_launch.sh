mkdir -p gpurun_out
for t in 256 512 1024; do for b in 1 2 4 8; do
  if [ $((t*b)) -le 2048 ]; then
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --threads $t --blocks-per-cu $b 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('threads=$t bpc=$b  value=%.3e  ms/step=%.3f  kernel_ms=%.4f frac=%.3f'%(d['value'],d['ms_per_step'],r['avg_launch_ms'],r['frac']))"
  fi
done; done
