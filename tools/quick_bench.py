#!/usr/bin/env python3
"""quick_bench.py -- kernel-level timing without torch: one engine, the
streaming probes, then the whole-step and two-sub-step paths.
    python tools/quick_bench.py [particles] [nx] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5 * 10**7
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
import json  # noqa: E402
extra = json.loads(os.environ.get("PIC1DP_INPUT", "{}"))     # e.g. '{"iptcldist": 2, "species_v0": [3.0]}'
# PIC1DP_NPE: reference ranks reproduced as virtual ranks of this one process (the field solve then sums in the npe-rank order)
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx, **extra), npe=int(os.environ.get("PIC1DP_NPE", "1")))
if os.environ.get("PIC1DP_THREADS") or os.environ.get("PIC1DP_BPC"):
    eng.set_launch(int(os.environ.get("PIC1DP_THREADS", "0")), int(os.environ.get("PIC1DP_BPC", "0")))
if "--probe" in sys.argv:
    for nr, nw in ((1, 1), (4, 0), (4, 3), (7, 3), (1, 0), (7, 0)):
        from pic1dp_amd import probe
        print("probe read %d write %d: %.0f GB/s" % (nr, nw, probe.stream(nr, nw, n, 10)), flush=True)
eng.particle_load()
eng.interaction_collect_charge()
eng.field_solve_electric()
only0 = os.environ.get("PIC1DP_QB_ONLY_STEP") == "1"     # A/B runs: the whole-step path alone
for mode in ((0,) if only0 else (0, 1)):
    eng.set_step_mode(mode)
    eng.step(int(os.environ.get("PIC1DP_QB_WARMUP", "3")))
    eng.sync()
    # the step time WITHOUT the per-kernel HIP events (two hipEventRecord per marker launch cost ~4 us each on the
    # stream: 8 us of a 140 us step, profiles/r04/experiments/event_overhead.log); then the same steps with them
    t0 = time.perf_counter()
    eng.step(steps)
    t_enq = time.perf_counter() - t0          # the host's share: enqueueing (the GPU runs behind it)
    eng.sync()
    dt = time.perf_counter() - t0
    eng.kernel_stats_enable(True)
    eng.timers_reset()
    t0 = time.perf_counter()
    eng.step(steps)
    eng.sync()
    dt_ev = time.perf_counter() - t0
    ks = [eng.kernel_stats(k) for k in (0, 1, 2, 3, 4, 6)]
    names = ["fused", "push", "deposit", "step_half", "step_full", "step_one"]
    parts = ["%s %.4f ms" % (nm, ms / cnt) for nm, (ms, cnt) in zip(names, ks) if cnt]
    print("mode %d: %.4e updates/s  %.4f ms/step  | %s | with the events in the stream %.4f ms/step | host enqueue %.4f ms/step"
          % (mode, n * 2 * steps / dt, dt / steps * 1e3, ", ".join(parts), dt_ev / steps * 1e3, t_enq / steps * 1e3), flush=True)
    eng.kernel_stats_enable(False)
    if eng.predict_kind() == 1:   # the tiles: their bound on |w| and the terms that went past the fixed-point sums so far
        print("        tiles: bound on |q| %.3e, %d terms in doubles so far" % eng.kernel_stats(13), flush=True)

if only0:
    eng.close()
    sys.exit(0)

# the reference's own call sequence (src/pic1dp.F90:79-93): three call sites per
# sub-step, served lazily by the whole-step kernels unless PIC1DP_LAZY_CALLS=0
eng.set_step_mode(0)
eng.sync()
eng.kernel_stats_enable(True)
eng.timers_reset()
t0 = time.perf_counter()
for _ in range(steps):
    for irk in (1, 2):
        eng.interaction_push_particle(irk)
        eng.particle_optimize(irk)
        eng.interaction_collect_charge()
        eng.field_solve_electric()
eng.sync()
dt = time.perf_counter() - t0
ks = [eng.kernel_stats(k) for k in (0, 1, 2, 3, 4, 6)]
parts = ["%s %.4f ms" % (nm, ms / cnt) for nm, (ms, cnt) in zip(names, ks) if cnt]
print("calls : %.4e updates/s  %.4f ms/step  | %s" % (n * 2 * steps / dt, dt / steps * 1e3, ", ".join(parts)), flush=True)
