#!/usr/bin/env python3
"""default_run.py -- BASELINE configs[0]'s workload WHOLE through the Fortran host: the reference's default input
(src/pic1dp_input.F90:35,109,113,128,250: bump-on-tail, 6.4e6 markers, nx 192, dt 0.05, time_max 500 = 10 000 steps,
output_all every 0.5 = 1 001 records), seed_type 1, run by pic1dp_amd/fortran/pic1dp_host as the reference driver
runs it (src/pic1dp.F90:78-109).

    python tools/default_run.py [--modes 0,2,3] [--time-max 500] [--oracle-steps 100] [--keep DIR]

For every mode of the host (PIC1DP_FUSED: 0 the three call sites, 2 pic1dp_hip_step(1) per iteration, 3 all the steps
up to the next output_all in one call) it prints: wall clock of the process, the time loop's wall clock as the host takes
it (PIC1DP_HOST_PROFILE=2: nothing added to the loop; --profile 1: the split steps | output_all in the library | file
writes, which costs a device synchronisation per iteration), the size and record count of pic1dp.out against the layout of
SURVEY 5.5 (103 000 B per record), 2 gamma fitted by the rule of tools/OutputData.py:153-170 against the analytic
0.16766, and the records' int E^2 dx of the first --oracle-steps steps against the CPU oracle (test infrastructure;
this tool is a measurement harness, not the product)."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,2,3")
    ap.add_argument("--time-max", type=float, default=500.0)
    ap.add_argument("--oracle-steps", type=int, default=100)
    ap.add_argument("--keep", default="")
    ap.add_argument("--profile", default="2", help="PIC1DP_HOST_PROFILE: 1 the split (a device sync per iteration), 2 the time loop's wall clock only")
    ap.add_argument("--timers", default="17", help="PIC1DP_TIMERS: the host's timer mode (0 off, 1 HIP events around every launch, n every n-th block of 64 launches: the host's default 17)")
    args = ap.parse_args()
    from pic1dp_amd import output
    import pic1dp_amd
    exe = os.path.join(ROOT, "pic1dp_amd", "fortran", "pic1dp_host")
    inp = pic1dp_amd.make_input(time_max=args.time_max)
    nsteps = int(round(args.time_max / inp.dt))
    nrec = nsteps // 10 + 1
    want_bytes = output.header_bytes(inp) + nrec * output.record_bytes(inp)
    print("workload: %d markers, nx %d, %d steps, %d records, pic1dp.out must be %d B (header %d + %d x %d)"
          % (inp.nparticle_max, inp.nx, nsteps, nrec, want_bytes, output.header_bytes(inp), nrec, output.record_bytes(inp)),
          flush=True)
    e_ref = None
    if args.oracle_steps > 0:
        import oracle
        t0 = time.perf_counter()
        sim = oracle.Sim(oracle.make_input(time_max=args.time_max))
        sim.load()
        sim.collect_charge()
        sim.solve_field()
        e_ref = [sim.field_energy()]
        for _ in range(args.oracle_steps // 10):
            sim.step(10)
            e_ref.append(sim.field_energy())
        e_ref = np.array(e_ref)
        print("oracle: %d steps on one thread in %.1f s" % (args.oracle_steps, time.perf_counter() - t0), flush=True)
    base = tempfile.mkdtemp(prefix="default_run_", dir=args.keep or None)
    first = None
    for mode in args.modes.split(","):
        wd = os.path.join(base, "fused" + mode)
        os.makedirs(wd)
        env = dict(os.environ, PIC1DP_FUSED=mode, PIC1DP_HOST_PROFILE=args.profile, PIC1DP_TIMERS=args.timers, PIC1DP_TIME_MAX=repr(args.time_max))
        t0 = time.perf_counter()
        r = subprocess.run([exe], cwd=wd, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        tail = [ln for ln in r.stdout.splitlines() if not ln.startswith(("i", "t")) or "%" not in ln]
        print("== PIC1DP_FUSED=%s  rc %d  process wall clock %.2f s" % (mode, r.returncode, wall), flush=True)
        print("\n".join(tail[-8:]), flush=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            continue
        path = os.path.join(wd, "pic1dp.out")
        size = os.path.getsize(path)
        d = output.OutputData(path)
        g2 = d.growthrate_energy_fit(15.0, 45.0) if args.time_max >= 45.0 else float("nan")
        print("   pic1dp.out %d B (%s), %d records (%s); 2 gamma over t in [15, 45] = %.5f (analytic 0.16766: %+.2f %%)"
              % (size, "ok" if size == want_bytes else "WRONG, want %d" % want_bytes, d.ntime,
                 "ok" if d.ntime == nrec else "WRONG, want %d" % nrec, g2, (g2 / 0.16766 - 1.0) * 100), flush=True)
        print("   int E^2 dx: first %.6e  max %.6e at t = %.1f  last %.6e"
              % (d.scalars[0, 1], d.scalars[:, 1].max(), d.scalars[int(np.argmax(d.scalars[:, 1])), 0], d.scalars[-1, 1]))
        if e_ref is not None:
            got = d.scalars[: len(e_ref), 1]
            print("   int E^2 dx of records 0 .. %d against the oracle: max relative difference %.3g"
                  % (len(e_ref) - 1, float(np.max(np.abs(got / e_ref - 1.0)))), flush=True)
        if first is None:
            first = d
        else:
            n = min(first.ntime, d.ntime, 60)        # the linear phase: modes agree to rounding
            print("   against the first mode, records 0 .. %d: max relative difference of int E^2 dx %.3g"
                  % (n - 1, float(np.max(np.abs(d.scalars[:n, 1] / first.scalars[:n, 1] - 1.0)))), flush=True)
        if not args.keep:
            os.remove(path)
    if not args.keep:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
