#!/bin/bash
# -f0'/f0 in the one-exp form (default) against the reference's operation order (PIC1DP_DLNF0=ref), each with and
# without the carry through memory (PIC1DP_CARRY), at C3 (1e8 markers, nx 1024): default input, a bump-on-tail
# species with general constants, two-stream2.  Alternating fresh processes; log -> profiles/r03/experiments/
export PIC1DP_QB_WARMUP=40 PIC1DP_QB_ONLY_STEP=1
N=${1:-1e8}; NX=${2:-1024}
GEN='{"species_temperature":[1.3],"species_temperature2":[0.7],"species_mass":[1.1],"species_density":[0.85],"species_v0":[4.5]}'
TS='{"iptcldist":2,"species_density":[1.0],"species_v0":[3.0]}'
for r in 1 2; do
  for cfg in '{}' "$GEN" "$TS"; do
    for form in one_exp ref; do for c in 1 0; do
      echo "== run $r $cfg form $form carry $c: $(PIC1DP_DLNF0=$form PIC1DP_CARRY=$c PIC1DP_INPUT=$cfg python tools/quick_bench.py $N $NX 60 | grep 'mode 0')"
    done; done
  done
done
