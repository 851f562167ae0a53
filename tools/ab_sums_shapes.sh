#!/bin/bash
# launch shapes of k_step_sums (threads x workgroups per CU)
export PIC1DP_QB_WARMUP=30
C5='{"iptcldist": 0, "species_density": [1.0], "species_v0": [0.0], "lx": 12.566370614359172}'
for sh in "256 1" "384 1" "512 1" "640 1"; do set -- $sh
  echo "== landau nx4096 sums $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
done
for sh in "512 1" "768 1" "1024 1"; do set -- $sh
  echo "== bump   nx4096 sums $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 python tools/quick_bench.py 1e8 4096 60 | grep 'mode 0')"
done
for sh in "512 2" "768 2" "1024 2" "512 3" "512 4" "1024 1"; do set -- $sh
  echo "== landau nx1024 sums $1x$2: $(PIC1DP_PRED_KIND=2 PIC1DP_THREADS=$1 PIC1DP_BPC=$2 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
for sh in "512 2" "768 2" "512 3" "1024 1"; do set -- $sh
  echo "== landau nx1024 tiles $1x$2: $(PIC1DP_THREADS=$1 PIC1DP_BPC=$2 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 1024 60 | grep 'mode 0')"
done
for sh in "512 1" "768 1" "1024 1" "512 2"; do set -- $sh
  echo "== landau nx2048 sums $1x$2: $(PIC1DP_PRED_KIND=2 PIC1DP_THREADS=$1 PIC1DP_BPC=$2 PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 2048 60 | grep 'mode 0')"
done
echo "== landau nx2048 tiles default: $(PIC1DP_INPUT="$C5" python tools/quick_bench.py 1e8 2048 60 | grep 'mode 0')"
