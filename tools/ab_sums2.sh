#!/bin/bash
# ab_sums2.sh -- VERDICT r04 item 9: two kept modes, the prediction as TWENTY sums in thread-private LDS slots (one
# workgroup of 512 threads per CU: 80 KB of slots) against the tiles (k_step_one<NM = 2>: ten atomics at random cells).
# Needs the tuning build   PIC1DP_EXTRA_FLAGS=-DPIC1DP_TUNE_SUMS2 PIC1DP_LIB_OUT=$PWD/pic1dp_amd/lib/v_sums2.so python pic1dp_amd/build.py --force
# in which PIC1DP_SUMS2=1 launches the sums' MARKER KERNEL in the tiles' place (no solve reads its sums: the fields of such a
# run mean nothing, the kernel's duration -- the markers' arithmetic and traffic are the same whatever the field -- is the figure).
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1 PIC1DP_INPUT='{"nmode":2,"modes":[1,2]}'
export PIC1DP_LIB=$(cd "$(dirname "$0")/.." && pwd)/pic1dp_amd/lib/v_sums2.so
for r in 1 2; do
  for nx in 1024 512; do
    unset PIC1DP_SUMS2
    echo "== run $r nmode 2 1e8/$nx tiles        : $(python tools/quick_bench.py 1e8 $nx 60 | grep 'mode 0' | cut -c1-110)"
    export PIC1DP_SUMS2=1
    echo "== run $r nmode 2 1e8/$nx twenty sums  : $(python tools/quick_bench.py 1e8 $nx 60 | grep 'mode 0' | cut -c1-110)"
  done
  unset PIC1DP_SUMS2
  echo "== run $r nmode 2 1e7/256 tiles        : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
  export PIC1DP_SUMS2=1
  echo "== run $r nmode 2 1e7/256 twenty sums  : $(python tools/quick_bench.py 1e7 256 300 | grep 'mode 0' | cut -c1-110)"
done
