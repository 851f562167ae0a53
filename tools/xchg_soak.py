#!/usr/bin/env python3
"""xchg_soak.py -- the one-hop exchange at the target's rank count, for thousands of steps instead of the suite's six:
eight ranks on ONE GPU (four processes of two contexts: a GPU box admits six processes on its card), every step's charge
summed through eight slots and flags, posted from the marker launches' tails (step) or by the exchange launch (calls).

    python tools/xchg_soak.py [--steps 4000] [--calls-steps 1500] [--nx 1024] [--markers 80003]

Passes when no rank reports a time-out, every rank counts the exchanges the sequence has, and E / chargeden / the whole
field-energy history are bit-identical on all eight ranks.  (tests/test_gpu_exchange.py holds the same run, six steps long,
against the virtual-rank engine.)"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(out, nproc, per, kw, steps, mode, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PIC1DP_XCHG_TIMEOUT_MS="60000",
               PIC1DP_RANKS_PER_PROC=str(per))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "xchg_worker.py"),
           out, json.dumps(kw), str(steps), mode]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        sys.exit("ranks failed (%s, %d steps):\n%s\n%s" % (mode, steps, r.stdout[-3000:], r.stderr[-3000:]))
    return [np.load(out + ".rank%d.npz" % k) for k in range(nproc * per)], dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--calls-steps", type=int, default=1500)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--markers", type=int, default=80003)
    ap.add_argument("--nproc", type=int, default=4)
    ap.add_argument("--per", type=int, default=2)
    a = ap.parse_args()
    kw = dict(nparticle_max=a.markers, nx=a.nx)
    world = a.nproc * a.per
    with tempfile.TemporaryDirectory() as d:
        for mode, steps, port in (("step", a.steps, 29871), ("calls", a.calls_steps, 29873)):
            ranks, dt = run(os.path.join(d, mode), a.nproc, a.per, kw, steps, mode, port)
            expect = 2 + steps if mode == "step" else 1 + 2 * steps
            same = all(np.array_equal(r[k], ranks[0][k]) for r in ranks[1:] for k in ("E", "cd", "hist"))
            counts = sorted({int(r["exchanges"]) for r in ranks})
            tails = sorted({int(r["tails"]) for r in ranks})
            ok = same and counts == [expect] and np.all(np.isfinite(ranks[0]["hist"]))
            print("%d ranks (%d processes x %d), %s, %d markers, nx %d, %d steps: exchanges per rank %s (expected %d), tail posts %s, "
                  "E / chargeden / history bit-identical on all ranks: %s, int E^2 dx %.6e -> %.6e, %.1f s  %s"
                  % (world, a.nproc, a.per, mode, a.markers, a.nx, steps, counts, expect, tails, same,
                     float(ranks[0]["e0"]), float(ranks[0]["energy"]), dt, "OK" if ok else "FAILED"), flush=True)
            if not ok:
                sys.exit(1)


if __name__ == "__main__":
    main()
