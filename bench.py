#!/usr/bin/env python3
"""bench.py -- particle-updates/sec of the PIC1D time-step hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json): configs[2] = bump-on-tail, 10^8 markers, 1024 grid
cells, the configuration the metric is quoted on; it fits one GPU.  For N > 1
every GPU holds 10^8 markers (weak scaling; configs[4] is this at N = 8) and the
per-GPU charge vector is summed by one RCCL all-reduce per sub-step.

A "step" is one time step = two Runge-Kutta sub-steps of push+gather, deposit,
(all-reduce,) field solve over all markers.  A particle-update is one marker
through one sub-step: value = markers_total * 2 * K / wall time.  Markers are
resident in HBM before the timed region (the native loader runs untimed).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : algorithmic bytes of the fused push+gather+deposit kernel
                 (80 B per particle-update) / its mean launch
                 duration from HIP events on the engine's stream, vs 8 TB/s
  cpu_baseline : the CPU oracle (line-faithful restatement of the reference,
                 "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pic1dp_amd  # noqa: E402  (loads libpic1dp_hip.so first: one HIP runtime per process)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_UPDATE = 80.0    # fused push+gather(+deposit), delta-f FP64

CONFIGS = {
    # BASELINE.json configs[1] and configs[2]
    "c2": dict(nparticle_max=10**7, nx=256),
    "c3": dict(nparticle_max=10**8, nx=1024),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--particles", type=int, default=0, help="markers per GPU (override)")
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="workgroup size of the particle kernels")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--unfused", action="store_true",
                    help="time the three reference call sites per sub-step instead of step() "
                         "(PIC1DP_LAZY_CALLS=0 in the environment: one kernel per call)")
    ap.add_argument("--step-mode", type=int, default=0, choices=[0, 1],
                    help="0: whole-step kernels (half-step state recomputed); 1: two fused sub-steps")
    ap.add_argument("--force-host-allreduce", action="store_true",
                    help="testing only: skip RCCL, reduce the charge on the host with gloo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles-per-core", type=int, default=2 * 10**6)
    ap.add_argument("--cpu-steps", type=int, default=40)
    return ap.parse_args()


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, per_core, steps):
    """the oracle (CPU restatement of the reference path) on the host cores:
    T threads, each owning one reference rank block with a private charge array
    (= a T-rank reference run), plus the same on 1 thread (= 1 MPI rank)."""
    import oracle  # test infrastructure, used here only as the timed CPU baseline
    T = host_cores()
    out = {}
    for label, threads in (("all", T), ("one", 1)):
        n = per_core * threads
        inp = oracle.make_input(nparticle_max=n, nx=cfg["nx"])
        sim = oracle.Sim(inp, npe=threads, nthreads=threads)
        sim.load()
        sim.collect_charge()
        sim.solve_field()
        sim.step(1)                      # warm-up
        t0 = time.perf_counter()
        sim.step(steps)
        dt = time.perf_counter() - t0
        out[label] = (n * 2 * steps / dt, n, dt)
        del sim
    v, n, dt = out["all"]
    return {
        "value": v, "unit": "updates/s", "cores": T, "kind": "port",
        "sample": "oracle (oracle/pic1dp_oracle.c, gcc -O3 -ffp-contract=off), bump-on-tail nx=%d, "
                  "%d markers (%d per core, one reference rank block per core), %d steps, %.1f s"
                  % (cfg["nx"], n, per_core, steps, dt),
        "value_1core": out["one"][0], "cpu_model": cpu_model(),
    }


def main():
    if os.environ.get("PIC1DP_BENCH_TRACE"):     # debugging aid: dump all stacks after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PIC1DP_BENCH_TRACE"]), exit=True)
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "(one process per GPU); WORLD_SIZE is 1" % a.gpus)
        a.gpus = world

    cfg = dict(CONFIGS[a.config])
    if a.particles:
        cfg["nparticle_max"] = a.particles
    if a.nx:
        cfg["nx"] = a.nx
    per_gpu = cfg["nparticle_max"]
    total = per_gpu * world

    dist = None
    if world > 1 or a.force_host_allreduce:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # control plane only (barrier, max over ranks, unique-id broadcast); the
        # data-path all-reduce is RCCL inside libpic1dp_hip.so, on the engine's stream
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    # every rank owns one reference block of the global array (PETSC_DECIDE split)
    inp = pic1dp_amd.make_input(nparticle_max=total, nx=cfg["nx"])
    # one GPU per rank; more ranks than visible GPUs (a rehearsal of the multi-rank
    # control flow on a one-GPU box, host-staged all-reduce only) share the devices
    ndev = pic1dp_amd.device_count()
    device = local_rank % max(ndev, 1)
    if world > ndev and not a.force_host_allreduce and not os.environ.get("PIC1DP_BENCH_ALLOW_SHARED_GPU"):
        sys.exit("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank; "
                 "--force-host-allreduce rehearses the control flow on fewer)" % (world, ndev))
    eng = pic1dp_amd.Pic1dp(inp, rank=rank, nranks=world, device=device)
    if a.threads or a.blocks_per_cu:
        eng.set_launch(a.threads, a.blocks_per_cu)
    eng.set_step_mode(a.step_mode)

    # charge all-reduce: RCCL on the engine's stream.  Safety net only: if the RCCL
    # communicator cannot be created on some rank, every rank switches to the
    # split-phase deposit with a host-staged gloo all-reduce (slow, and flagged in
    # the JSON line) instead of producing no number at all.
    allreduce_kind, comm_error = "rccl", None
    if dist is not None:
        import torch
        ok = 0 if a.force_host_allreduce else 1
        if ok:
            try:
                pic1dp_amd.parallel.bootstrap_comm(eng, dist)
            except Exception as e:          # noqa: BLE001
                ok, comm_error = 0, str(e)
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            allreduce_kind = "host-staged gloo (RCCL unavailable: %s)" % (comm_error or "forced / failed on another rank")
    host_staged = allreduce_kind != "rccl"

    def collect_charge():
        if not host_staged:
            eng.interaction_collect_charge()
            return
        import torch
        t = torch.from_numpy(eng.charge_local())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        eng.charge_reduced(t.numpy())

    t0 = time.perf_counter()
    eng.particle_load()
    load_s = time.perf_counter() - t0
    collect_charge()
    eng.field_solve_electric()
    eng.sync()

    def device_sync():
        eng.sync()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize(device)
                return "torch.cuda.synchronize"
        except Exception:
            pass
        return "hipStreamSynchronize"

    def barrier():
        if dist is not None:
            dist.barrier()

    def run(nsteps):
        if not a.unfused and not host_staged:
            eng.step(nsteps)
            return
        for _ in range(nsteps):
            for irk in (1, 2):
                eng.interaction_push_particle(irk)
                collect_charge()
                eng.field_solve_electric()

    # the first ~30 steps over freshly loaded markers run 3-13 % slower than the
    # steady state (tools/ramp_probe.py), whatever the caller's W: settle first,
    # untimed and reported, then do the W warm-up steps of the contract
    settle = max(0, 30 - a.warmup)
    device_sync()      # the first call initialises torch's device context (seconds): not between warm-up and timing
    run(settle)
    run(a.warmup)
    sync_kind = device_sync()
    eng.kernel_stats_enable(True)
    eng.timers_reset()
    barrier()
    device_sync()
    t0 = time.perf_counter()
    run(a.steps)
    device_sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fused_ms, fused_n = eng.kernel_stats(0)
    push_ms, push_n = eng.kernel_stats(1)
    dep_ms, dep_n = eng.kernel_stats(2)
    half_ms, half_n = eng.kernel_stats(3)
    full_ms, full_n = eng.kernel_stats(4)
    energy = eng.field_energy()
    _, np_local = eng.local_sizes()

    # the same work through the reference's own three call sites per sub-step
    # (push, collect_charge, solve_field -- src/pic1dp.F90:80-89), which the
    # library serves lazily with the same whole-step kernels; reported beside
    # `value`, never instead of it
    calls_elapsed = None
    if not a.unfused and not host_staged:
        eng.kernel_stats_enable(False)
        barrier()
        device_sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            for irk in (1, 2):
                eng.interaction_push_particle(irk)
                eng.particle_optimize(irk)
                eng.interaction_collect_charge()
                eng.field_solve_electric()
        device_sync()
        barrier()
        calls_elapsed = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([calls_elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            calls_elapsed = float(t.item())

    if rank == 0:
        value = total * 2.0 * a.steps / elapsed
        # dominant kernel of the path that ran; one launch = np_local particle-updates,
        # algorithmic bytes 80 B per update whatever the kernel really moves
        if full_n:
            kname, kms, kn = "k_step_full (2nd sub-step: recompute half-step state, push+gather, deposit)", full_ms, full_n
            path = "whole-step kernels k_step_half + k_step_full (half-step state recomputed, not stored)"
            if a.unfused:
                path += ", reached through the three reference call sites per sub-step (lazy call sites)"
        elif push_n:
            kname, kms, kn, path = "k_push (separate gather+push)", push_ms, push_n, "separate push / deposit kernels"
        else:
            kname, kms, kn, path = "k_push<fused push+gather+deposit>", fused_ms, fused_n, "two fused push+gather+deposit sub-steps"
        kbytes = ALG_BYTES_PER_UPDATE
        avg_ms = kms / max(kn, 1)
        achieved = kbytes * np_local / (avg_ms * 1e-3) / 1e9 if kn else 0.0
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes
        # (profiles/summarize_pmc.py; FETCH_SIZE doubled per MI355X_MICROARCH.md)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("particles_per_gpu") == per_gpu and tj.get("nx") == cfg["nx"]:
                    key = "k_step_full" if full_n else "k_push"
                    traffic = tj.get("hbm_bytes_per_launch_by_kernel", {}).get(key)
            except (OSError, ValueError):
                pass
        # the box's own streaming rates (second denominator): plain copy
        # and the dominant kernel's traffic shape (4 arrays read, 3 written)
        probe_n = int(min(np_local, 10**8))
        copy_gbs = max(eng.stream_probe(1, 1, probe_n, 10) for _ in range(3))
        shape_gbs = max(eng.stream_probe(4, 3, probe_n, 10) for _ in range(3))
        out = {
            "metric": "particle-updates/sec", "value": value, "unit": "updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "settle_steps_before_warmup": settle,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "bump-on-tail delta-f, %g markers per GPU, nx=%d (BASELINE configs[%s]); "
                            "multirand constant seeds (al_int=3, seed_type=1), one reference rank block per GPU"
                            % (per_gpu, cfg["nx"], {"c2": "1", "c3": "2"}[a.config]),
                "particles_total": total, "particles_per_gpu": per_gpu, "nx": cfg["nx"],
                "nmode": 1, "dt": 0.05,
                "parallelism": "particle shard x%d, replicated grid, RCCL all-reduce of the charge vector" % world,
                "path": path, "allreduce": allreduce_kind if world > 1 or host_staged else "none (1 GPU)",
                "sync": sync_kind, "load_seconds": load_s,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kname, "avg_launch_ms": avg_ms, "launches": kn,
                "algorithmic_bytes_per_update": kbytes, "updates_per_launch": np_local,
                "deposit_kernel_avg_ms": dep_ms / dep_n if dep_n else None,
                "step_half_kernel_avg_ms": half_ms / half_n if half_n else None,
                "step_half_kernel_algorithmic_GBs": (kbytes * np_local / (half_ms / half_n * 1e-3) / 1e9) if half_n else None,
                "measured_copy_GBs": copy_gbs, "measured_4read_3write_GBs": shape_gbs,
                "traffic_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and kn) else None,
            },
            "field_energy_end": energy,
        }
        if calls_elapsed:
            out["drop_in_call_sites"] = {
                "value": total * 2.0 * a.steps / calls_elapsed, "unit": "updates/s",
                "ms_per_step": calls_elapsed / a.steps * 1e3,
                "what": "the same %d steps through pic1dp_hip_push / particle_optimize / collect_charge / "
                        "solve_field, the three call sites of src/pic1dp.F90:80-89 (served lazily by the "
                        "whole-step kernels)" % a.steps,
            }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_particles_per_core, a.cpu_steps)
        print(json.dumps(out), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
