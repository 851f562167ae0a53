#!/bin/bash
# ab_priv_threads.sh -- k_step_one<PRIV> as ONE workgroup of 1024 threads per CU (tuning builds -DPIC1DP_PRIV_THREADS=1024,
# with 6 and with 4 waves per SIMD as the register budget: v_p1024.so, v_p1024w4.so) against the default two workgroups of
# 768: with the drawn chunk tail a single workgroup's sixteen waves finish together, where two workgroups of a CU do not
# (profiles/r05/experiments/stamps_drawn_tail.log).  Alternating fresh processes.
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
L=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib
for r in 1 2; do
  for v in default p1024 p1024w4; do
    if [ $v = default ]; then unset PIC1DP_LIB; else export PIC1DP_LIB=$L/v_$v.so; fi
    echo "== run $r $v  C1 6.4e6/192 : $(python tools/quick_bench.py 6.4e6 192 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v  C2 1e7/256   : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v  1.25e7/1024  : $(python tools/quick_bench.py 1.25e7 1024 300 | grep 'mode 0' | cut -c1-110)"
    echo "== run $r $v  C3 1e8/1024  : $(python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0' | cut -c1-110)"
  done
done
