#!/usr/bin/env python3
"""Generate tests/golden/multirand_reference.json from the REFERENCE's own
multirand module (src/multirand.F90 compiled with flang into
oracle/_ref/libmultirand_ref.so by oracle/Makefile; only possible where
/root/reference exists).  The fixture holds data only: inputs (engine, seed
type, rank, warm-up) and the numbers the reference produced.

    python tests/golden/gen_multirand_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def hex64(a):
    return ["%016X" % (int(x) & 0xFFFFFFFFFFFFFFFF) for x in a]


def main():
    oracle.build()
    if not oracle.RefMultirand.available():
        sys.exit("oracle/_ref/libmultirand_ref.so missing: needs /root/reference and flang")
    ref = oracle.RefMultirand()
    cases = []
    for al in (1, 2, 3):
        for mype in range(8):
            ref.init(al, 1, mype, 5, True)
            ints = ref.int_array(10**6)
            u = ints.view(np.uint64)
            reals = ref.real_array(4)
            gauss = ref.gaussian_array(6)      # even count: leaves no buffered value behind
            cases.append(dict(
                al_int=al, seed_type=1, mype=mype, warmup=5, selftest=True,
                first_int64=hex64(ints[:8]),
                int64_at_20630=hex64(ints[20630:20640]),
                xor_1e6="%016X" % int(np.bitwise_xor.reduce(u)),
                sum_1e6="%016X" % int(u.sum(dtype=np.uint64)),
                next_real64_bits=hex64(reals.view(np.int64)),
                next_gaussian64_bits=hex64(gauss.view(np.int64)),
            ))
    # warm-up 0 and the other engines' first reals
    extra = []
    for al, warm in ((1, 0), (2, 0), (3, 0), (3, 1)):
        ref.init(al, 1, 0, warm, True)
        extra.append(dict(al_int=al, seed_type=1, mype=0, warmup=warm, selftest=True,
                          first_real64_bits=hex64(ref.real_array(4).view(np.int64))))
    out = dict(
        source="reference src/multirand.F90 compiled with AMD flang (oracle/Makefile target ref), "
               "called through oracle/ref_multirand_shim.F90",
        generator="tests/golden/gen_multirand_golden.py",
        note="draw order per case: 1e6 int64, then 4 real64, then 6 gaussian64",
        cases=cases, warmup_cases=extra)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multirand_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(cases), "cases")


if __name__ == "__main__":
    main()
