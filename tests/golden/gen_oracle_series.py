#!/usr/bin/env python3
"""Generate tests/golden/oracle_series.json: field-energy time series of small
runs of the CPU oracle.  This pins the ORACLE against regressions between
rounds; it is not a reference output (the reference cannot be built here)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

CASES = [
    dict(input=dict(nparticle_max=20000, nx=32), npe=1, steps=40),
    dict(input=dict(nparticle_max=20001, nx=48, iptcldist=2, species_density=[1.0], species_v0=[3.0]), npe=3, steps=30),
    dict(input=dict(nparticle_max=15000, nx=32, iptcldist=0, species_density=[1.0], species_v0=[0.0],
                    lx=12.566370614359172, linear=1), npe=2, steps=30),
]


def main():
    out = []
    for c in CASES:
        sim = oracle.Sim(oracle.make_input(**c["input"]), npe=c["npe"])
        sim.load()
        sim.collect_charge()
        sim.solve_field()
        e = [sim.field_energy()]
        for _ in range(c["steps"]):
            sim.step(1)
            e.append(sim.field_energy())
        out.append(dict(c, energy_hex=[float(x).hex() for x in e]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_series.json")
    with open(path, "w") as f:
        json.dump(dict(source="oracle/pic1dp_oracle.c (NOT the reference)", generator="tests/golden/gen_oracle_series.py",
                       cases=out), f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
