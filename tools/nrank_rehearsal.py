#!/usr/bin/env python3
"""nrank_rehearsal.py -- what ONE GPU can measure of an N-rank step: the per-GPU share of a
strong-scaled 1e8-marker run over 8 GPUs (1.25e7 markers, nx 1024, the 8-rank summation order) stepped

  A. by one process that owns a 1-rank RCCL communicator: marker launch (its tail packs the charge) -> ncclAllReduce of
     nx + 8 doubles -> paired solve -- every launch of the 8-GPU RCCL step, with the all-reduce's time a LOWER BOUND
     (one rank: no peer to wait for, no xGMI hop), against PIC1DP_TAIL=0 (the separate packing launch of rounds 3-4)
     and against the plain one-rank step (no charge sum at all);
  B. by two processes sharing the GPU through the one-hop exchange (bench.py --gpus 2 --allreduce p2p; the two ranks'
     kernels share the CUs, so absolute times are about twice a GPU's own -- the on / off comparison and the
     attribution are what this part is for).

    python tools/nrank_rehearsal.py [--steps 300] [--no-two-ranks]

Device times per step under the reference's timer ids (src/pic1dp_global.F90:38-50) come from the library's HIP-event
timers in a pass of their own; the ms-per-step figures from wall clock around un-instrumented steps."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IWT_PUSH, IWT_COLLECT, IWT_FIELD, IWT_ALLREDUCE = 4, 6, 7, 21


def one_process(n, nx, npe, steps, comm, tail):
    os.environ["PIC1DP_TAIL"] = "1" if tail else "0"
    import pic1dp_amd
    eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx), npe=npe)
    eng.particle_load()
    if comm:
        eng.comm_init(eng.comm_unique_id())
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.step(60)
    eng.sync()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.step(steps)
        eng.sync()
        best.append((time.perf_counter() - t0) / steps * 1e3)
    eng.timers_enable(True)
    eng.step(3)
    eng.sync()
    eng.timers_reset()
    eng.step(50)
    eng.sync()
    attr = {k: eng.timer_ms(i) / 50 * 1e3 for k, i in (("marker_us", IWT_PUSH), ("pack_us", IWT_COLLECT),
                                                      ("allreduce_us", IWT_ALLREDUCE), ("field_us", IWT_FIELD))}
    eng.timers_enable(False)
    tails = eng.kernel_stats(10)[1]
    energy = eng.field_energy()
    eng.close()
    return sorted(best), attr, tails, energy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1.25e7)
    ap.add_argument("--whole", type=float, default=1e8)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--npe", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--no-two-ranks", action="store_true")
    a = ap.parse_args()
    n = int(a.particles)
    # the denominator of the strong-scaling figure: the whole 1e8 markers on ONE GPU, same box, same run
    whole, _, _, _ = one_process(int(a.whole), a.nx, a.npe, 100, False, True)
    print("0. the whole %d markers on this one GPU (npe-%d order, one launch per step where the rule allows): %s ms per step"
          % (int(a.whole), a.npe, " ".join("%.4f" % b for b in whole)), flush=True)
    print("A. one process, %d markers, nx %d, %d-rank summation order; ms per step: five blocks of %d steps, sorted"
          % (n, a.nx, a.npe, a.steps), flush=True)
    rows = []
    for name, comm, tail in (("plain one-rank step (no charge sum; one launch per step)", False, True),
                             ("1-rank RCCL, charge packed in the marker launch's tail", True, True),
                             ("1-rank RCCL, separate packing launch (PIC1DP_TAIL=0)", True, False),
                             ("1-rank RCCL, tail again", True, True),
                             ("1-rank RCCL, separate launch again", True, False)):
        blocks, attr, tails, energy = one_process(n, a.nx, a.npe, a.steps, comm, tail)
        rows.append((name, blocks, attr))
        print("   %-62s %s  | device us per step: marker %.1f  pack %.1f  all-reduce %.1f  field %.1f | tails %d | int E^2 dx %.15e"
              % (name, " ".join("%.4f" % b for b in blocks), attr["marker_us"], attr["pack_us"], attr["allreduce_us"],
                 attr["field_us"], tails, energy), flush=True)
    t1 = whole[len(whole) // 2]
    print("   strong-scaling budget 1 -> 8 GPUs from these medians (whole / share):", flush=True)
    for name, blocks, attr in rows:
        ts = blocks[len(blocks) // 2]
        extra = ""
        if "RCCL" in name:
            ar = attr["allreduce_us"]
            # >= 6x needs share <= whole / 6: how long may the 8-GPU all-reduce take beyond the one-rank one measured here?
            room = (t1 / 6.0 - ts) * 1e3
            extra = "  | 6x allows the all-reduce %.1f us more than the one-rank communicator's %.1f us" % (room, ar)
        print("      %-62s %.4f / %.4f = %.2fx%s" % (name, t1, ts, t1 / ts, extra), flush=True)
    if a.no_two_ranks:
        return
    print("B. two processes share the GPU, one-hop exchange (bench.py --gpus 2 --allreduce p2p), %d markers each" % n, flush=True)
    for tail in ("1", "0", "1", "0"):
        env = dict(os.environ, PIC1DP_TAIL=tail, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
                   PIC1DP_XCHG_TIMEOUT_MS="60000")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c3", "--particles", str(n),
               "--nx", str(a.nx), "--strong-total", str(2 * n), "--allreduce", "p2p", "--no-cpu-baseline",
               "--steps", "100", "--warmup", "20"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("   PIC1DP_TAIL=%s: bench.py failed\n%s" % (tail, (r.stdout + r.stderr)[-1500:]), flush=True)
            continue
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        at = d["attribution"]
        print("   PIC1DP_TAIL=%s: %.4f ms per step (blocks %.4f .. %.4f) | device ms per step: marker %.4f  field launch %.4f "
              "of which exchange %.4f | exchanges per step %.2f"
              % (tail, d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], at["particle_kernels_ms_per_step"],
                 at["field_solve_ms_per_step"], at["exchange_inside_field_launch_ms_per_step"], at["exchanges_per_step"]),
              flush=True)


if __name__ == "__main__":
    main()
