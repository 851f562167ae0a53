#!/usr/bin/env python3
"""bench.py -- particle-updates/sec of the PIC1D time-step hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json): configs[2] = bump-on-tail, 10^8 markers, 1024 grid
cells, the configuration the metric is quoted on; it fits one GPU.  For N > 1 `value`
is the SAME 10^8 markers split over the N GPUs ("scaling": "strong" -- the metric is
"at 10^8 particles, 1/2/4/8 MI355X" and north_star's ">= 6x from 1 -> 8 GPUs" is a
statement about exactly that curve), the per-GPU charge vector summed once per
sub-step (RCCL all-reduce on the engine's stream, or the library's one-hop exchange);
the weak reading (10^8 markers PER GPU) is measured in the same run and listed beside
it (`weak_per_gpu`; --headline weak swaps the two).  The other GPU configurations of
BASELINE.json are selectable, and the two multi-GPU ones are measured by the driver's
plain command as well: `--gpus 4` adds a `configs3_c4` object, `--gpus 8` a
`configs4_c5` object (--extra-configs):
    c2  bump-on-tail, 10^7 markers per GPU, nx 256                      (weak)
    c4  two-stream (iptcldist 2, v0 = 3), 10^8 markers IN TOTAL, nx 512 (strong)
    c5  Landau damping (Maxwellian, lx = 4 pi), 10^8 per GPU, nx 4096   (weak)

A "step" is one time step = two Runge-Kutta sub-steps of push+gather, deposit,
(charge sum over GPUs,) field solve over all markers.  A particle-update is one
marker through one sub-step: value = markers_total * 2 * K / wall time.  Markers are
resident in HBM before the timed region (the native loader runs untimed).

ONE JSON line on rank 0 (contract in the task description) with, besides the
contract keys:
  roofline         : the dominant kernel priced at the bytes it has to move (32 B read
                     + 24 B written per marker for the second sub-step's kernel) / its
                     mean launch duration from HIP events on the engine's stream, vs
                     8 TB/s; the SURVEY 8(d) price of 80 B per update is reported as
                     reference_priced_GBs, never as the fraction
  weak_per_gpu     : N > 1: 10^8 markers PER GPU (the other reading of "at 10^8 particles,
                     1/2/4/8 MI355X"), measured in the same run with the same charge sum
  strong_1e8_total : 10^8 markers IN TOTAL split over the N GPUs -- `value` itself for
                     N > 1 (kept as an object of its own for N = 1 and --headline weak)
  configs3_c4 / configs4_c5 : BASELINE configs[3] (N = 4) / configs[4] (N = 8), 20 steps
  exchange         : the same two workloads with the one-hop charge exchange instead of
                     the RCCL all-reduce (N > 1)
  attribution      : device time per step of the particle kernels, the charge sum and
                     the field solve (the reference's timer ids, HIP events)
  cpu_baseline     : the CPU oracle (line-faithful restatement of the reference,
                     "port") timed on this box's host cores on a bounded sample (N = 1)
"""
import argparse
import datetime
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pic1dp_amd = None   # imported by main() in a rank process only (it loads libpic1dp_hip.so): the launching parent of
                    # `python bench.py --gpus N` never maps the HIP library and never touches a GPU

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PRICED_BYTES_PER_UPDATE = 80.0  # what a store-and-reload push+gather sub-step would move
IWT_PUSH, IWT_COLLECT, IWT_FIELD, IWT_ALLREDUCE = 4, 6, 7, 21   # the reference's timer ids (src/pic1dp_global.F90:38-50)

# BASELINE.json configs[1..4]; physics per SURVEY 8(d).  "per_gpu": weak, "total": strong.
CONFIGS = {
    "c2": dict(index=1, what="bump-on-tail", per_gpu=10**7, inp=dict(nx=256)),
    "c3": dict(index=2, what="bump-on-tail", per_gpu=10**8, inp=dict(nx=1024)),
    "c4": dict(index=3, what="two-stream (iptcldist 2, v0 = 3, density 1)", total=10**8,
               inp=dict(nx=512, iptcldist=2, species_density=[1.0], species_v0=[3.0])),
    "c5": dict(index=4, what="Landau damping (Maxwellian, lx = 4 pi)", per_gpu=10**8,
               inp=dict(nx=4096, iptcldist=0, species_density=[1.0], species_v0=[0.0], lx=4.0 * math.pi)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed block of --steps steps is run this many times back to back (each bracketed by "
                         "barrier + device sync, max over ranks); value and ms_per_step are the MEDIAN block's, "
                         "ms_per_step_min / _max the spread")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--particles", type=int, default=0,
                    help="override the config's marker count (per GPU for c2/c3/c5, in total for c4)")
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--strong-total", type=int, default=10**8, help="markers of the strong_1e8_total object")
    ap.add_argument("--no-strong", action="store_true")
    ap.add_argument("--headline", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1, config c3: which reading of 'at 10^8 particles, N GPUs' is `value` -- strong (10^8 in total, "
                         "what north_star's >= 6x speed-up 1 -> 8 is a statement about; auto) or weak (10^8 per GPU); the "
                         "other one is measured beside it")
    ap.add_argument("--extra-configs", default="auto",
                    help="BASELINE multi-GPU configurations measured beside the headline (20 steps each): auto = c4 when "
                         "N = 4 (configs[3]), c5 when N = 8 (configs[4]); a comma list insists (tests); none: nothing")
    ap.add_argument("--extra-particles", type=int, default=0,
                    help="(tests) marker count of the extra configurations: per GPU for c5, in total for c4")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--rehearse-with-host", action="store_true",
                    help="(tests) --allreduce auto also rehearses the host-staged sum, so that the choice by rehearsal "
                         "can run where RCCL cannot (ranks sharing one GPU)")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "rccl", "p2p", "host"],
                    help="charge sum over GPUs: RCCL all-reduce, the one-hop exchange, or host-staged gloo "
                         "(testing); auto = RCCL for the headline, the exchange measured beside it")
    ap.add_argument("--threads", type=int, default=0, help="workgroup size of the particle kernels")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--unfused", action="store_true",
                    help="time the three reference call sites per sub-step instead of step() "
                         "(PIC1DP_LAZY_CALLS=0 in the environment: one kernel per call)")
    ap.add_argument("--step-mode", type=int, default=0, choices=[0, 1],
                    help="0: whole-step kernels (half-step state recomputed); 1: two fused sub-steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="roofline.traffic from the committed profile constant instead of two rocprofv3 --pmc "
                         "passes of this command run as child processes (N = 1 only)")
    ap.add_argument("--cpu-particles", type=int, default=10**7, help="markers of the CPU sample (BASELINE.md 3: C2)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of the all-core CPU sample")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------
def host_cpu_info():
    """physical cores / sockets of the box, the CPUs this process may run on, and the
    cgroup CPU quota (a GPU box hands a job a share of the host)"""
    cores, sockets, model = set(), set(), "unknown"
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                        sockets.add(phys)
                    phys = core = None
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
            if a != "max":
                quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    return dict(physical_cores=len(cores) or affinity, sockets=len(sockets) or 1, logical_cpus_allowed=affinity,
                cgroup_cpu_quota=quota, cpu_model=model)


def cpu_baseline(n, nx, target_seconds):
    """the oracle (CPU restatement of the reference path) on the host cores, BASELINE.md
    section 3: the C2 workload (bump-on-tail, 10^7 markers, nx 256), T threads each owning
    one reference rank block with a private charge array (= a T-rank reference run), T =
    all physical cores this job may use; plus the same markers on 1 thread (= 1 MPI rank)."""
    import oracle  # test infrastructure, used here only as the timed CPU baseline
    info = host_cpu_info()
    T = min(info["physical_cores"], info["logical_cpus_allowed"])
    if info["cgroup_cpu_quota"]:
        T = max(1, min(T, int(info["cgroup_cpu_quota"] + 0.5)))
    out = {}
    for label, threads, secs in (("all", T, target_seconds), ("one", 1, target_seconds * 0.6)):
        inp = oracle.make_input(nparticle_max=n, nx=nx)
        sim = oracle.Sim(inp, npe=threads, nthreads=threads)
        sim.load()
        sim.collect_charge()
        sim.solve_field()
        sim.step(1)                      # warm-up (first touch of the arrays)
        t0 = time.perf_counter()
        sim.step(2)                      # the pace, for sizing the sample
        pace = (time.perf_counter() - t0) / 2
        steps = int(max(20 if threads > 1 else 4, min(2000, secs / max(pace, 1e-6))))
        t0 = time.perf_counter()
        sim.step(steps)
        dt = time.perf_counter() - t0
        out[label] = (n * 2 * steps / dt, steps, dt)
        del sim
    v, steps, dt = out["all"]
    return {
        "value": v, "unit": "updates/s", "cores": T, "kind": "port",
        "sample": "oracle (oracle/pic1dp_oracle.c, gcc -O3 -ffp-contract=off), BASELINE.md section 3 workload: "
                  "bump-on-tail, %d markers, nx=%d, one reference rank block per thread, %d threads, %d steps, %.1f s"
                  % (n, nx, T, steps, dt),
        "value_per_core": v / T, "value_1core": out["one"][0],
        "sample_1core": "%d steps, %.1f s" % (out["one"][1], out["one"][2]),
        "host": info,
        "cores_note": "threads = physical cores the job may use: min(physical cores, allowed CPUs, cgroup CPU quota)",
    }


# ---------------------------------------------------------------------------
# one engine + how its charge is summed over ranks
# ---------------------------------------------------------------------------
class Job:
    def __init__(self, a, name, inp, rank, world, device, dist, shared_gpus):
        self.a, self.name, self.rank, self.world, self.dist = a, name, rank, world, dist
        self.eng = pic1dp_amd.Pic1dp(inp, rank=rank, nranks=world, device=device)
        if a.threads or a.blocks_per_cu:
            self.eng.set_launch(a.threads, a.blocks_per_cu)
        self.eng.set_step_mode(a.step_mode)
        self.kind = "none (1 GPU)"
        self.rccl_why = self.p2p_why = None
        self.have_rccl = self.have_p2p = False
        self.xchg_memkind = 0
        if world > 1:
            from pic1dp_amd import parallel
            if a.allreduce in ("auto", "rccl"):
                if shared_gpus:
                    self.rccl_why = "ranks share GPUs (RCCL needs one GPU per rank)"
                else:
                    # every rank gets the same answer (pic1dp_amd/parallel.py); a communicator that
                    # comes up on some ranks only ends the job at the process group's timeout
                    if rank == 0:
                        sys.stderr.write("bench.py: %d ranks entering ncclCommInitRank; should it come up on some ranks "
                                         "only, the others wait for the process group's 300 s timeout and the job ends "
                                         "without a line\n" % world)
                        sys.stderr.flush()
                    # RCCL prints a version banner on STDOUT when its first communicator comes up: stdout is for the one
                    # JSON line (as with gloo's note in main())
                    sys.stdout.flush()
                    saved_stdout = os.dup(1)
                    os.dup2(2, 1)
                    try:
                        self.rccl_why = parallel.bootstrap_comm(self.eng, dist)
                    finally:
                        sys.stdout.flush()
                        os.dup2(saved_stdout, 1)
                        os.close(saved_stdout)
                self.have_rccl = self.rccl_why is None
            if a.allreduce in ("auto", "p2p", "host"):   # "host": connected too, to be measured beside it
                self.p2p_why = parallel.bootstrap_exchange(self.eng, dist)
                if self.p2p_why is None:
                    # plain (coarse-grained) device memory is not coherent across GPUs inside a kernel: a stale
                    # slot would be summed silently.  The library only hands it out when PIC1DP_XCHG_MEM=3 asks for
                    # it; a measurement does not take it on any rank
                    self.xchg_memkind = self.eng.xchg_info()[0]
                    if not parallel.agree(dist, self.xchg_memkind in (1, 2)):
                        self.p2p_why = "the exchange area of some rank is plain device memory (memkind 3)"
                self.have_p2p = self.p2p_why is None
            want = a.allreduce
            if want == "auto" and getattr(a, "auto_choice", None):      # main()'s rehearsal has timed the kinds
                want = a.auto_choice
            if want == "auto":
                want = "rccl" if self.have_rccl else ("p2p" if self.have_p2p else "host")
            if want == "rccl" and not self.have_rccl:
                self.fatal("RCCL requested but unavailable: %s" % self.rccl_why)
            if want == "p2p" and not self.have_p2p:
                self.fatal("one-hop exchange requested but unavailable: %s" % self.p2p_why)
            self.use(want)

    def fatal(self, msg):
        if self.rank == 0:
            sys.stderr.write("bench.py: %s\n" % msg)
        sys.exit(4)

    def use(self, kind):
        """every rank switches at the same point of the run"""
        if kind == "rccl":
            self.eng.set_allreduce(1)
            self.kind = "rccl"
        elif kind == "p2p":
            self.eng.set_allreduce(2)
            self.kind = "one-hop exchange (IPC-mapped slots, rank-order sum inside the solve's launch)"
        else:
            self.kind = "host-staged gloo"
            if self.rccl_why or self.p2p_why:
                self.kind += " (RCCL: %s; exchange: %s)" % (self.rccl_why, self.p2p_why)

    @property
    def host_staged(self):
        return self.kind.startswith("host-staged")

    def collect_charge(self):
        if not self.host_staged:
            self.eng.interaction_collect_charge()
            return
        import torch
        t = torch.from_numpy(self.eng.charge_local())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.eng.charge_reduced(t.numpy())

    def init_field(self):
        t0 = time.perf_counter()
        self.eng.particle_load()
        self.load_s = time.perf_counter() - t0
        self.collect_charge()
        self.eng.field_solve_electric()
        self.eng.sync()

    def call_sites(self, nsteps):
        """the reference's own sequence, src/pic1dp.F90:79-93"""
        for _ in range(nsteps):
            for irk in (1, 2):
                self.eng.interaction_push_particle(irk)
                self.eng.particle_optimize(irk)
                self.collect_charge()
                self.eng.field_solve_electric()

    def run(self, nsteps):
        if not self.a.unfused and not self.host_staged:
            self.eng.step(nsteps)
        else:
            self.call_sites(nsteps)


def measure_traffic(a, key):
    """HBM bytes per launch of the dominant kernel, measured for THIS command: two child processes run it again
    for five steps under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, --kernel-trace
    only) while this process idles; bytes = 2 * FETCH_SIZE KiB (gfx950 wide-read correction) + WRITE_SIZE KiB,
    mean over the kernel's launches (profiles/summarize_pmc.py).  (None, why) when that is not possible."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        import summarize_pmc
    except ImportError as e:
        return None, "profiles/summarize_pmc.py: %s" % e
    # the program after `--` must be the interpreter binary itself: a shim or a script would be an exec hop
    # under the profiler's GPU-initialising preload
    python = os.path.realpath(sys.executable)
    try:
        with open(python, "rb") as f:
            if f.read(4) != b"\x7fELF":
                return None, "%s is not an ELF binary (an exec hop under the profiler is not allowed)" % python
    except OSError as e:
        return None, "cannot read %s: %s" % (python, e)
    base = tempfile.mkdtemp(prefix="pic1dp_pmc_", dir="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(base, counter.lower())
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   python, os.path.join(ROOT, "bench.py"), "--config", a.config, "--steps", "5", "--warmup", "1",
                   "--no-cpu-baseline", "--no-strong", "--no-traffic-pass"]
            # everything that shapes the launch (grid size -> flush atomics and staging bytes) goes to the child
            for flag, val in (("--particles", a.particles), ("--nx", a.nx), ("--threads", a.threads),
                              ("--blocks-per-cu", a.blocks_per_cu)):
                if val:
                    cmd += [flag, str(val)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                                   timeout=300)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, "rocprofv3 --pmc %s pass: %s" % (counter, e)
            if r.returncode != 0:
                tail = " | ".join((r.stderr or "").strip().splitlines()[-4:])
                return None, "rocprofv3 --pmc %s pass failed (exit %d): %s" % (counter, r.returncode, tail[-600:])
            csvs = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not csvs:
                return None, "rocprofv3 --pmc %s pass wrote no counter file" % counter
            mean, _ = summarize_pmc.mean_by_kernel(",".join(csvs), counter)
            if key not in mean:
                return None, "%s is not in the %s pass" % (key, counter)
            vals[counter] = mean[key]
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return vals["FETCH_SIZE"] * 1024 * 2.0 + vals["WRITE_SIZE"] * 1024, None


def launch_ranks(n, cmd=None):
    """`python bench.py --gpus N` started as ONE process (the form the driver uses for N = 1; the reference's one
    launch line is `mpiexec -n $(NPE_RUN) ./pic1dp`, run/Makefile:41): start N fresh rank processes of this very
    command -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run
    would hand them -- wait, pass their output through (rank 0 prints the one JSON line) and return the worst exit
    code.  This parent never loads the HIP library, never initialises a GPU and never replaces itself with another
    program; a rank that fails is not started again, and when one fails the others get 20 s to follow before they
    are terminated (exact PIDs).  cmd: the rank command (tests hand in a stand-in; default: this very command)."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the one-hop exchange and RCCL need it
    env.setdefault("OMP_NUM_THREADS", "1")
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), PIC1DP_BENCH_LAUNCHER="self")
    cmd = cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0"),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is the job's stdout (the other ranks print nothing there).  It is passed through by a reader
    # thread: the loop below watches ALL ranks meanwhile, so a rank that dies while rank 0 waits for it in a
    # collective (RCCL, or gloo's 300 s timeout) ends the job after the 20 s grace instead of hanging with it
    import threading
    lines = [0]

    def pass_through():
        for ln in procs[0].stdout:
            out = sys.stdout if ln.startswith("{") else sys.stderr     # anything a library printed there: not the line
            out.write(ln)
            out.flush()
            lines[0] += ln.startswith("{")

    reader = threading.Thread(target=pass_through, daemon=True)
    reader.start()
    worst, deadline = 0, None
    while any(p.poll() is None for p in procs):
        codes = [p.poll() for p in procs]
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.monotonic() + float(os.environ.get("PIC1DP_BENCH_GRACE_S", "20"))
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    lines = lines[0]
    for r, p in enumerate(procs):
        c = p.wait()
        if c:
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, c))
            worst = max(worst, abs(c) if c > 0 else 128 - c)
    if not worst and lines != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON lines instead of one\n" % lines)
        worst = 5
    return min(worst, 255)


def rehearse_charge_sums(a, phys, rank, world, device, dist, shared, steps=60):
    """time a short run of the strong-scaling share with every charge sum that comes up; every rank returns the same
    dict: ms per step per kind (max over ranks), what failed, and the kind chosen (None: leave it to availability)"""
    import torch
    from pic1dp_amd import parallel
    out = {"markers_total": int(a.strong_total), "steps": steps, "ms_per_step": {}, "failed": {}, "chosen": None}
    n_tot = max(int(a.strong_total), 2 * world)
    rj = Job(a, "rehearsal", pic1dp_amd.make_input(nparticle_max=n_tot, **phys), rank, world, device, dist, shared)
    kinds = [k for k, ok in (("rccl", rj.have_rccl), ("p2p", rj.have_p2p)) if ok]
    if a.rehearse_with_host:
        kinds.append("host")
    out["unavailable"] = {"rccl": rj.rccl_why, "one-hop exchange": rj.p2p_why}
    if len(kinds) < 2:
        rj.eng.close()
        out["why"] = "fewer than two kinds of charge sum came up: nothing to choose between"
        return out
    rj.init_field()
    for kind in kinds:
        rj.use(kind)
        dist.barrier()
        err = None
        try:
            rj.run(15)
            rj.eng.sync()
            dist.barrier()
            t0 = time.perf_counter()
            rj.run(steps)
            rj.eng.sync()
            el = time.perf_counter() - t0
        except pic1dp_amd.Pic1dpError as e:
            err, el = str(e), 0.0
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if not parallel.agree(dist, err is None):
            out["failed"][kind] = err or "failed on another rank"
            break                      # (an engine whose exchange failed cannot go on: the kinds not yet timed stay untimed)
        out["ms_per_step"][kind] = float(t.item()) / steps * 1e3
    try:
        rj.eng.close()
    except pic1dp_amd.Pic1dpError:
        pass
    timed_kinds = {k: v for k, v in out["ms_per_step"].items() if k != "host" or a.rehearse_with_host}
    if timed_kinds:
        out["chosen"] = min(timed_kinds, key=timed_kinds.get)
        out["why"] = ("the shortest step of the rehearsal (%s)"
                      % ", ".join("%s %.4f ms" % (k, v) for k, v in sorted(timed_kinds.items(), key=lambda kv: kv[1])))
    else:
        out["why"] = "no kind completed the rehearsal"
    return out


def main():
    if os.environ.get("PIC1DP_BENCH_TRACE"):     # debugging aid: dump all stacks after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PIC1DP_BENCH_TRACE"]), exit=True)
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))            # before anything of this process has touched the GPU
    global pic1dp_amd
    # (tests) PIC1DP_BENCH_ENGINE=<dir>: a stand-in for the pic1dp_amd package (tests/fake_engine) -- the control plane of an
    # N-rank run rehearsed on CPUs at rank counts the GPU box cannot host; the line then says so in "data"
    fake_engine = os.environ.get("PIC1DP_BENCH_ENGINE")
    if fake_engine:
        sys.path.insert(0, fake_engine)
    import pic1dp_amd       # loads libpic1dp_hip.so first: one HIP runtime per process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size holds\n" % (a.gpus, world))
        a.gpus = world

    cfg = CONFIGS[a.config]
    phys = dict(cfg["inp"])
    if a.nx:
        phys["nx"] = a.nx
    strong_cfg = "total" in cfg
    # N > 1 on the metric's own configuration: `value` is the strong-scaling figure (10^8 markers in total), the weak one
    # (10^8 per GPU) is the object measured beside it; --headline weak swaps them.  N = 1: the two coincide.
    headline_strong = False
    if strong_cfg:
        total = a.particles or cfg["total"]
        other_total, other_key = None, None
    else:
        weak_total = (a.particles or cfg["per_gpu"]) * world
        headline_strong = world > 1 and a.config == "c3" and not a.no_strong and a.headline != "weak"
        total = a.strong_total if headline_strong else weak_total
        other_total = weak_total if headline_strong else a.strong_total
        other_key = "weak_per_gpu" if headline_strong else "strong_1e8_total"
    per_gpu = total // world

    dist = None
    if world > 1 or a.allreduce == "host":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # control plane only (barrier, max over ranks, id / handle exchange); the data path is
        # inside libpic1dp_hip.so, on the engine's stream
        # gloo's C++ side prints its "connected to N peer ranks" note on stdout: stdout is for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=300))
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # one GPU per rank; more ranks than visible GPUs (a rehearsal of the multi-rank
    # control flow on a one-GPU box: exchange or host-staged sum only) share the devices
    ndev = pic1dp_amd.device_count()
    device = local_rank % max(ndev, 1)
    shared = world > ndev
    if shared and a.allreduce == "rccl":
        sys.exit("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank; --allreduce auto, p2p or "
                 "host rehearses the control flow on fewer)" % (world, ndev))

    # --allreduce auto on several ranks: which charge sum?  Not by preference but by REHEARSAL: a
    # scratch job of the strong-scaling share's size (10^8 markers in total over the ranks: where a step is short enough
    # for the sum to matter) is stepped with every kind that came up -- RCCL's all-reduce behind the marker launch's
    # packing tail, and the one-hop exchange posted from that tail --, max over ranks, and the shorter step wins.  On a
    # scratch job because an exchange that fails (a peer that never delivers: bounded wait, PIC1DP_ERR_COMM) leaves
    # an engine that cannot go on; the headline job is created afterwards with the kind chosen.
    a.auto_choice = None
    rehearsal = None
    if world > 1 and a.allreduce == "auto":
        rehearsal = rehearse_charge_sums(a, phys, rank, world, device, dist, shared)
        a.auto_choice = rehearsal.get("chosen")

    # every rank owns one reference block of the global array (PETSC_DECIDE split)
    job = Job(a, "headline", pic1dp_amd.make_input(nparticle_max=total, **phys), rank, world, device, dist, shared)
    eng = job.eng
    job.init_field()

    def device_sync(e=eng):
        e.sync()
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

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(j, nsteps, fn=None):
        """EXACTLY nsteps steps between barrier + device sync on both sides; max over ranks"""
        fn = fn or j.run
        barrier()
        device_sync(j.eng)
        t0 = time.perf_counter()
        fn(nsteps)
        device_sync(j.eng)
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    def kernel_table(e):
        names = ("k_push_fused", "k_push", "k_deposit", "k_step_half", "k_step_full")
        tab = {nm: e.kernel_stats(k) for k, nm in enumerate(names)}
        tab["k_step_one"] = e.kernel_stats(6)
        return tab

    def attribution(j, nsteps=10):
        """device time per step under the reference's timer ids (HIP events on the stream; the one-hop exchange, which
        runs inside the field solve's launch, from a wall clock read in the kernel), in a pass of its own so that the
        timed region carries no extra events.  Per charge-sum kind: with RCCL the step is marker kernel | pack |
        all-reduce | paired solve; with the exchange marker kernel | [exchange + paired solve] in one launch -- the line
        shows directly whether the communication latency or the marker kernel limits a strong-scaled step."""
        e = j.eng
        e.timers_enable(True)
        # a few steps first, not counted: the step after a change of path (the call sites before, the timers themselves)
        # restarts the prediction with a first-sub-step pass of its own, which is not what a step of the run costs
        j.run(3)
        device_sync(e)
        e.timers_reset()
        if world > 1:
            e.xchg_time(reset=True)
        barrier()
        j.run(nsteps)
        device_sync(e)
        out = {"particle_kernels_ms_per_step": e.timer_ms(IWT_PUSH) / nsteps,
               "charge_pack_ms_per_step": e.timer_ms(IWT_COLLECT) / nsteps,
               "allreduce_ms_per_step": e.timer_ms(IWT_ALLREDUCE) / nsteps,
               "field_solve_ms_per_step": e.timer_ms(IWT_FIELD) / nsteps,
               "exchange_inside_field_launch_ms_per_step": None,
               "steps": nsteps, "charge_sum": j.kind}
        if world > 1:
            x_ms, x_n = e.xchg_time(reset=True)
            out["exchange_inside_field_launch_ms_per_step"] = x_ms / nsteps
            out["exchanges_per_step"] = x_n / nsteps
        e.timers_enable(False)
        for k in ("particle_kernels_ms_per_step", "charge_pack_ms_per_step", "allreduce_ms_per_step",
                  "field_solve_ms_per_step", "exchange_inside_field_launch_ms_per_step"):
            if out[k] is not None:
                out[k] = max_over_ranks(out[k])
        out["note"] = ("max over ranks, device time.  rccl: allreduce_ms_per_step is the ncclAllReduce of the packed vector "
                       "(nx + 8 doubles), charge_pack the kernel in front of it, field_solve the paired solve behind it.  "
                       "one-hop exchange: the sum runs inside the field solve's launch -- field_solve_ms_per_step contains "
                       "exchange_inside_field_launch_ms_per_step (stores into the peers' slots, the wait for their flags, "
                       "the rank-order sum; in-kernel 100 MHz clock), allreduce_ms_per_step is then 0")
        barrier()
        return out

    # the first ~60 ms of stepping after an idle period of the GPU (the loader runs on the host) run 1-10 % slower
    # than the steady state (tools/ramp_probe.py, 10-step batches at 10^8 markers: 1.06 0.99 0.98 0.97 0.96 0.96 ...
    # ms per step): about 0.1 s of steps (the same count on every rank: from the marker count, not from a clock) are
    # added in front of the caller's W and reported as warmup_effective
    settle = max(0, min(2000, max(30, int(0.1 / (per_gpu * 2.0 / 2.0e11)))) - a.warmup)
    device_sync()      # the first call initialises torch's device context (seconds): not between warm-up and timing
    job.run(settle)
    job.run(a.warmup)
    sync_kind = device_sync()
    eng.kernel_stats_enable(True)
    eng.timers_reset()
    # R blocks of EXACTLY a.steps steps each, back to back, nothing else inside a timed region: boxes of the pool differ by
    # +-1.5 % and a 20 ms region is one draw -- the line carries the median block and the spread
    repeats = max(1, a.repeats)
    blocks_s = [timed(job, a.steps) for _ in range(repeats)]
    elapsed = sorted(blocks_s)[(repeats - 1) // 2]         # the median block (the lower middle one for an even count)
    ktab = kernel_table(eng)
    eng.kernel_stats_enable(False)
    energy = eng.field_energy()
    _, np_local = eng.local_sizes()
    headline_kind = job.kind

    # the same work through the reference's own three call sites per sub-step (push,
    # collect_charge, solve_field -- src/pic1dp.F90:80-89), which the library serves lazily
    # with the same whole-step kernels; reported beside `value`, never instead of it
    calls_elapsed = None
    if not a.unfused and not job.host_staged:
        calls_elapsed = timed(job, a.steps, job.call_sites)
    attr = attribution(job)

    # ---- the other reading of "10^8 markers, N GPUs" beside the headline: weak (10^8 per GPU) beside the strong headline
    # of N > 1, or strong (10^8 in total) beside a weak one (N = 1: same run; --headline weak; c2 / c5) ------
    strong = None
    sjob = None
    if not a.no_strong and not strong_cfg:
        if world == 1 and a.strong_total == total:
            strong = {"value": total * 2.0 * a.steps / elapsed, "ms_per_step": elapsed / a.steps * 1e3,
                      "same_run_as_headline": True}
        else:
            sjob = Job(a, other_key, pic1dp_amd.make_input(nparticle_max=other_total, **phys), rank, world, device,
                       dist, shared)
            if job.kind != sjob.kind:        # the two objects are measured with the same charge sum
                sjob.use("rccl" if job.kind == "rccl" else ("p2p" if job.kind.startswith("one-hop") else "host"))
            sjob.init_field()
            sjob.run(settle + a.warmup)
            sjob.eng.kernel_stats_enable(True)
            sjob.eng.timers_reset()
            s_blocks = [timed(sjob, a.steps) for _ in range(repeats)]
            s_el = sorted(s_blocks)[(repeats - 1) // 2]
            stab = kernel_table(sjob.eng)
            sjob.eng.kernel_stats_enable(False)
            s_energy = sjob.eng.field_energy()
            strong = {"value": other_total * 2.0 * a.steps / s_el, "ms_per_step": s_el / a.steps * 1e3,
                      "same_run_as_headline": False, "repeats": repeats,
                      "ms_per_step_min": min(s_blocks) / a.steps * 1e3, "ms_per_step_max": max(s_blocks) / a.steps * 1e3,
                      "particle_kernel_avg_ms": {k: v[0] / v[1] for k, v in stab.items() if v[1]},
                      "field_energy_end": s_energy, "attribution": attribution(sjob)}
        s_tot = other_total if sjob is not None else total
        strong.update({"unit": "updates/s", "scaling": "weak" if headline_strong else "strong", "particles_total": s_tot,
                       "particles_per_gpu": s_tot // world, "allreduce": job.kind,
                       "what": ("the headline physics with %g markers PER GPU on the %d GPUs (weak scaling)" % (s_tot // world, world))
                               if headline_strong else
                               ("the headline physics with %g markers IN TOTAL split over the %d GPU(s)" % (s_tot, world))})

    # ---- the one-hop exchange beside RCCL (measured LAST: a failed exchange leaves a run
    # that cannot go on, and everything else is already measured) -----------------------
    exchange = None
    primary = "rccl" if job.kind == "rccl" else "host"
    if world > 1 and not job.kind.startswith("one-hop"):
        if job.have_p2p and (sjob is None or sjob.have_p2p):
            exchange = {}
            from pic1dp_amd import parallel
            head_label = "strong_1e8_total" if headline_strong else ("strong_total" if strong_cfg else "weak")
            for label, j, tot in ((head_label, job, total), ("weak" if headline_strong else "strong_1e8_total", sjob, other_total)):
                if j is None:
                    continue
                # a rank whose exchange fails (a peer that never delivers: the kernels give up
                # after PIC1DP_XCHG_TIMEOUT_MS, never hang) still meets the others at every
                # collective below, and all ranks drop the measurement together
                err, x_el, x_attr = None, 0.0, None
                j.use("p2p")
                barrier()
                try:
                    j.run(5)
                    j.eng.sync()
                except pic1dp_amd.Pic1dpError as e:
                    err = str(e)
                barrier()
                t0 = time.perf_counter()
                try:
                    if err is None:
                        j.run(a.steps)
                        j.eng.sync()
                except pic1dp_amd.Pic1dpError as e:
                    err = str(e)
                barrier()
                x_el = max_over_ranks(time.perf_counter() - t0)
                if parallel.agree(dist, err is None):
                    x_attr = attribution(j)
                    exchange[label] = {"value": tot * 2.0 * a.steps / x_el, "unit": "updates/s",
                                       "ms_per_step": x_el / a.steps * 1e3, "attribution": x_attr,
                                       "field_energy_end": j.eng.field_energy()}
                    j.use(primary)
                else:
                    exchange[label] = {"error": err or "failed on another rank"}
                    break
            exchange["what"] = ("the same workloads with the charge summed by the library's one-hop exchange "
                                "(pic1dp_hip_xchg_*: every GPU stores its charge into its slot on every peer, "
                                "every GPU adds the slots in rank order inside the field solve's launch) instead "
                                "of the headline's sum (%s)" % headline_kind)
        else:
            exchange = {"unavailable": job.p2p_why or (sjob.p2p_why if sjob else None)}

    # ---- BASELINE.json's own multi-GPU configurations, measured by the driver's plain command: configs[3] (two-stream,
    # 10^8 markers in total, nx 512) when N = 4, configs[4] (Landau, 10^8 per GPU, nx 4096: "the scaling-curve run") when
    # N = 8 -- a short run each (three blocks of --extra-steps steps) with the headline's charge sum and the same attribution
    def measure_extra(name):
        cx = CONFIGS[name]
        px = dict(cx["inp"])
        strong_x = "total" in cx
        if a.extra_particles:
            tot = a.extra_particles * (1 if strong_x else world)
        else:
            tot = cx["total"] if strong_x else cx["per_gpu"] * world
        xj = Job(a, name, pic1dp_amd.make_input(nparticle_max=tot, **px), rank, world, device, dist, shared)
        want_kind = "rccl" if headline_kind == "rccl" else ("p2p" if headline_kind.startswith("one-hop") else "host")
        if world > 1 and xj.kind != headline_kind:
            xj.use(want_kind)
        xj.init_field()
        x_settle = max(30, min(2000, int(0.1 / (max(tot // world, 1) * 2.0 / 2.0e11))))
        xj.run(x_settle)
        xb = [timed(xj, a.extra_steps) for _ in range(3)]
        x_el = sorted(xb)[1]
        x_attr = attribution(xj)
        x_energy = xj.eng.field_energy()
        x_kernel = xj.eng.kernel_bytes(6)["name"]
        xj.eng.close()
        return {"value": tot * 2.0 * a.extra_steps / x_el, "unit": "updates/s", "ms_per_step": x_el / a.extra_steps * 1e3,
                "ms_per_step_blocks": [b / a.extra_steps * 1e3 for b in xb], "steps": a.extra_steps, "warmup": x_settle,
                "scaling": "strong" if strong_x else "weak", "particles_total": tot, "particles_per_gpu": tot // world,
                "nx": px["nx"], "allreduce": xj.kind, "marker_kernel": x_kernel, "attribution": x_attr,
                "field_energy_end": x_energy, "steps_before_field_energy_end": x_settle + 3 * a.extra_steps + 3 + x_attr["steps"],
                "what": "BASELINE configs[%d]: %s, %g markers %s over %d GPU(s), nx=%d"
                        % (cx["index"], cx["what"], tot if strong_x else tot // world, "in total" if strong_x else "per GPU",
                           world, px["nx"])}

    extras = {}
    if a.extra_configs == "auto":
        extra_names = ["c4"] if world == 4 else (["c5"] if world == 8 else [])
    else:
        extra_names = [n for n in a.extra_configs.split(",") if n in CONFIGS]
    for name in extra_names:
        if name != a.config and not (exchange and any("error" in v for v in exchange.values() if isinstance(v, dict))):
            extras["configs%d_%s" % (CONFIGS[name]["index"], name)] = measure_extra(name)

    if rank == 0:
        value = total * 2.0 * a.steps / elapsed
        # what each marker kernel moves per marker and launch comes from the LIBRARY (pic1dp_hip_kernel_bytes: the
        # instantiation it launched -- mode, carry on or off), not from constants here
        names = {0: "k_push_fused", 1: "k_push", 2: "k_deposit", 3: "k_step_half", 4: "k_step_full", 6: "k_step_one"}
        kb = {nm: eng.kernel_bytes(k) for k, nm in names.items()}
        half_ms, half_n = ktab["k_step_half"]
        full_ms, full_n = ktab["k_step_full"]
        one_ms, one_n = ktab["k_step_one"]
        one_name = kb["k_step_one"]["name"]
        sums = bool(one_n) and one_name.startswith("k_step_sums")        # the large-grid kernel
        priv = bool(one_n) and one_name.startswith("k_step_one<sums>")   # six sums in thread-private LDS slots
        lazy = ", reached through the three reference call sites per sub-step (lazy call sites)" if a.unfused else ""
        if sums:
            dom = "k_step_one"
            kname = ("k_step_sums (one pass per step on a grid whose prediction tiles outgrow the LDS: recompute half-step "
                     "state, push+gather, wrap, deposit, store in place, and the six sums that predict the next step's "
                     "half-step field)")
            path = ("one pass over the markers per step (k_step_sums; the next half-step field follows from six sums "
                    "over the markers taken by the previous step's kernel)" + lazy)
        elif priv:
            dom = "k_step_one"
            kname = ("k_step_one<sums> (one pass per step: recompute half-step state, push+gather, wrap, deposit, store in "
                     "place, and six sums over the markers -- in thread-private LDS slots -- that predict the next step's "
                     "half-step field)")
            path = ("one pass over the markers per step (k_step_one; the next half-step field follows from six sums over "
                    "the markers taken by the previous step's kernel)" + lazy)
        elif one_n:
            dom = "k_step_one"
            kname = ("k_step_one (one pass per step: recompute half-step state, push+gather, wrap, deposit, store in "
                     "place, and the deposits that predict the next step's first-sub-step charge)")
            path = ("one pass over the markers per step (k_step_one; the first sub-step's charge is predicted by the "
                    "previous step's kernel as coefficients of the kept field modes)" + lazy)
        elif full_n:
            dom = "k_step_full"
            kname = "k_step_full (2nd sub-step: recompute half-step state, push+gather, wrap, deposit, store in place)"
            path = "whole-step kernels k_step_half + k_step_full (half-step state recomputed, not stored)" + lazy
        elif ktab["k_push"][1]:
            dom = "k_push"
            kname = "k_push (separate gather+push; the launch of the second sub-step is priced)"
            path = "separate push / deposit kernels (RK ping-pong sets)"
        else:
            dom = "k_push_fused"
            kname = "k_push<fused push+gather+deposit> (the launch of the second sub-step is priced)"
            path = "two fused push+gather+deposit sub-steps (RK ping-pong sets: 56 / 80 B per marker)"
        kms, kn = ktab[dom]
        rd, wr, carry_b = kb[dom]["read"], kb[dom]["written"], kb[dom]["carry"]
        kbytes, kcomp = rd + wr + carry_b, rd + wr
        one_launch_is_a_step = dom == "k_step_one"
        avg_ms = kms / max(kn, 1)
        achieved = kbytes * np_local / (avg_ms * 1e-3) / 1e9 if kn else 0.0
        achieved_compulsory = kcomp * np_local / (avg_ms * 1e-3) / 1e9 if kn else 0.0
        # HBM bytes per launch of that kernel: rocprofv3 --pmc passes of this command, committed under
        # profiles/ (a profile constant of the same workload, NOT measured in this run)
        traffic, traffic_src, traffic_why = None, None, None
        tkey = "k_step_one" if one_n else ("k_step_full" if full_n else "k_push")
        if sums:
            tkey = "k_step_sums"
        if world == 1 and not a.no_traffic_pass and not a.unfused and a.step_mode == 0:
            traffic, traffic_why = measure_traffic(a, tkey)
            if traffic is not None:
                traffic_src = ("measured for this command in this run: two child processes, rocprofv3 --pmc FETCH_SIZE "
                               "and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over 5 steps; "
                               "2 x FETCH_SIZE KiB (gfx950 wide-read correction) + WRITE_SIZE KiB, mean over the "
                               "launches of %s" % tkey)
        for tname in ("traffic_%s.json" % a.config, "traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is not None or not os.path.exists(tpath):
                continue
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("particles_per_gpu") == per_gpu and tj.get("nx") == phys["nx"]:
                    key = tkey
                    traffic = tj.get("hbm_bytes_per_launch_by_kernel", {}).get(key)
                    traffic_src = "profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
                                  "this command (committed profile of the same workload, not measured in this run%s)" \
                                  % (tname, ": " + traffic_why if traffic_why else "")
            except (OSError, ValueError):
                pass
        # the box's own streaming rates (second denominator): plain copy and the
        # dominant kernel's traffic shape (4 arrays read, 3 written)
        from pic1dp_amd import probe       # libpic1dp_probe.so: measurement code, not part of the product library
        probe_n = int(min(np_local, 10**8))
        copy_gbs = max(probe.stream(1, 1, probe_n, 10, device=device) for _ in range(3))
        shape_gbs = max(probe.stream(4, 3, probe_n, 10, device=device) for _ in range(3))
        # the same traffic in the layout the markers are stored in (x | v | w | p tiles, three written back in
        # place): what a kernel that did nothing but stream them would reach -- the denominator that does not
        # depend on where the allocator puts seven separate arrays
        tiled_ms = min(probe.layout(probe_n, 12, 10, device=device)[1] for _ in range(2))
        tiled_gbs = 56.0 * (probe_n // 4096 * 4096) / (tiled_ms * 1e-3) / 1e9
        # bytes the timed steps had to move: every launch of the three whole-step kernels at its own price
        step_bytes = None
        if full_n or one_n:
            step_bytes = sum(ktab[nm][1] * (kb[nm]["read"] + kb[nm]["written"] + kb[nm]["carry"])
                             for nm in ("k_step_half", "k_step_full", "k_step_one")) * np_local / (a.steps * repeats)
        out = {
            "metric": "particle-updates/sec", "value": value, "unit": "updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "warmup_effective": a.warmup + settle,
            "settle_steps_before_warmup": settle,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "repeats": repeats, "ms_per_step_min": min(blocks_s) / a.steps * 1e3,
            "ms_per_step_max": max(blocks_s) / a.steps * 1e3,
            "ms_per_step_blocks": [b / a.steps * 1e3 for b in blocks_s],
            "value_note": "median of %d back-to-back timed blocks of %d steps each (every block bracketed by barrier + "
                          "device sync, max over ranks); kernel averages are over all blocks" % (repeats, a.steps),
            "scaling": "strong" if (strong_cfg or headline_strong) else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not fake_engine else "NOT A MEASUREMENT: stand-in engine (%s), control-plane rehearsal" % fake_engine,
            "speedup_basis": (("value(N) / value(1) is the STRONG-scaling speed-up: the same %g markers in total on 1, 2, 4, 8 "
                               "GPUs -- the curve north_star's '>= 6x from 1 -> 8 GPUs' is judged on; the weak reading (%g "
                               "markers per GPU) is weak_per_gpu, measured in this run with the same charge sum")
                              % (total, (other_total or 0) // world)) if headline_strong else
                             ("one GPU: the strong and the weak reading of '10^8 markers on N GPUs' coincide" if world == 1 else
                              "value is %s-scaled (--headline / --config)" % ("strong" if strong_cfg else "weak")),
            "config": {
                "workload": "%s delta-f, %g markers %s, nx=%d (BASELINE configs[%d])%s; multirand constant seeds "
                            "(al_int=3, seed_type=1), one reference rank block per GPU"
                            % (cfg["what"], total if (strong_cfg or headline_strong) else per_gpu,
                               ("IN TOTAL over the %d GPUs" % world) if (strong_cfg or headline_strong) else "per GPU",
                               phys["nx"], cfg["index"],
                               ": strong scaling of the metric's own configuration, the figure the >= 6x target of 1 -> 8 "
                               "GPUs is judged on" if headline_strong else ""),
                "particles_total": total, "particles_per_gpu": per_gpu, "nx": phys["nx"],
                "nmode": 1, "dt": 0.05,
                "parallelism": "particle shard x%d, replicated grid, charge vector summed over GPUs once per "
                               "sub-step" % world,
                "path": path, "allreduce": headline_kind, "rccl_ranks": world if headline_kind == "rccl" else 0,
                "charge_sum_not_used_because": {"rccl": job.rccl_why, "one-hop exchange": job.p2p_why} if world > 1 else None,
                "charge_sum_chosen_by": rehearsal,
                "exchange_memkind": {0: None, 1: "fine-grained", 2: "uncached", 3: "plain"}[job.xchg_memkind],
                "marker_layout": "x, v, w, p interleaved in 32 KiB tiles in one slab per species",
                "timed_blocks": repeats,
                "kernel_launches_in_timed_steps": {("k_step_sums" if k == "k_step_one" and sums else k): v[1]
                                                   for k, v in ktab.items() if v[1]},
                "sync": sync_kind, "load_seconds": job.load_s,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname, "avg_launch_ms": avg_ms, "launches": kn,
                "bytes_per_marker": kbytes, "bytes_per_marker_source":
                    "pic1dp_hip_kernel_bytes: the library reports what the instantiation it launched moves (%s)"
                    % kb[dom]["name"],
                "bytes_per_marker_note":
                    "traffic THIS kernel moves: %g B read (x, v, p%s) + %g B written + %g B carry of -f0'/f0 (traffic the "
                    "kernel chooses to spend instead of evaluating it again)%s"
                    % (rd, ", w" if rd > 24.0 else "", wr, carry_b,
                       "; one launch = one whole time step (two particle-updates) per marker" if one_launch_is_a_step
                       else "; one launch = one particle-update per marker"),
                # the same launch priced on the strictly compulsory bytes only (no carry): 0.73 on 72 B flatters a kernel
                # that spends 16 of them by choice
                "bytes_compulsory": kcomp, "achieved_compulsory": achieved_compulsory,
                "frac_compulsory": achieved_compulsory / HBM_PEAK_GBS,
                "updates_per_launch": np_local * (2 if one_launch_is_a_step else 1),
                "step_half_kernel_avg_ms": half_ms / half_n if half_n else None,
                "step_half_kernel_GBs": ((kb["k_step_half"]["read"] + kb["k_step_half"]["carry"]) * np_local
                                         / (half_ms / half_n * 1e-3) / 1e9) if half_n else None,
                "whole_step_bytes": step_bytes,
                "whole_step_GBs": (step_bytes / (elapsed / a.steps) / 1e9) if step_bytes else None,
                "whole_step_frac": (step_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS) if step_bytes else None,
                "reference_priced_GBs": PRICED_BYTES_PER_UPDATE * np_local * (2 if one_launch_is_a_step else 1) / (avg_ms * 1e-3) / 1e9
                                        if kn else None,
                "reference_priced_note": "SURVEY 8(d) prices a push+gather sub-step that stores and reloads the RK "
                                         "state at 80 B per update (160 B per marker and step); this design moves %g B per "
                                         "marker and step -- how far the restructuring beats the priced data flow, "
                                         "not an HBM fraction" % (step_bytes / np_local if step_bytes else rd + wr),
                "measured_copy_GBs": copy_gbs, "measured_4read_3write_GBs": shape_gbs,
                "frac_of_measured_4read_3write": achieved / shape_gbs if shape_gbs else None,
                "measured_tiled_4read_3write_GBs": tiled_gbs,
                "frac_of_measured_tiled_stream": achieved / tiled_gbs if tiled_gbs else None,
                "measured_tiled_note": "a pure stream of 56 B per marker (4 tiles read, 3 written back in place)",
                "traffic_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and kn) else None,
            },
            "attribution": attr,
            "field_energy_end": energy, "steps_before_field_energy_end": settle + a.warmup + repeats * a.steps,
        }
        # the strong figure is the one measured with the headline's charge sum; the one-hop exchange measured beside
        # it is listed, never promoted
        xs = (exchange or {}).get("strong_1e8_total")
        if headline_strong:
            sv = {"value": value, "ms_per_step": elapsed / a.steps * 1e3}
            out["strong_1e8_total"] = dict(sv, unit="updates/s", scaling="strong", particles_total=total,
                                           particles_per_gpu=per_gpu, allreduce=headline_kind, same_run_as_headline=True,
                                           field_energy_end=energy,
                                           what="`value` itself: %g markers IN TOTAL split over the %d GPUs" % (total, world))
            if xs and "value" in xs:
                out["strong_1e8_total"]["by_charge_sum"] = {
                    headline_kind: sv, "one-hop exchange": {"value": xs["value"], "ms_per_step": xs["ms_per_step"]}}
            if strong is not None:
                out["weak_per_gpu"] = strong
        elif strong is not None:
            if xs and "value" in xs and not strong.get("same_run_as_headline"):
                strong["by_charge_sum"] = {
                    headline_kind: {"value": strong["value"], "ms_per_step": strong["ms_per_step"]},
                    "one-hop exchange": {"value": xs["value"], "ms_per_step": xs["ms_per_step"]}}
            out["strong_1e8_total"] = strong
        out.update(extras)
        if exchange is not None:
            out["exchange"] = exchange
        if calls_elapsed:
            out["drop_in_call_sites"] = {
                "value": total * 2.0 * a.steps / calls_elapsed, "unit": "updates/s",
                "ms_per_step": calls_elapsed / a.steps * 1e3,
                "what": "the same %d steps through pic1dp_hip_push / particle_optimize / collect_charge / "
                        "solve_field, the three call sites of src/pic1dp.F90:80-89 (served lazily by the "
                        "whole-step kernels)" % a.steps,
            }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_particles, 256, a.cpu_seconds)
        print(json.dumps(out), flush=True)

    if sjob is not None:
        sjob.eng.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
