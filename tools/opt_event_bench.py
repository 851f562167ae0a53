#!/usr/bin/env python3
"""opt_event_bench.py -- one marker optimisation event (src/pic1dp_particle.F90:356-813) at 1e7 markers: the markers
staying on the device (keys to the host, decisions back: kernels_opt.hip) against the pass on host copies
(PIC1DP_OPT_HOST=1); wall time of the step that carries the event and bytes between host and device.  With [blocks]
reference rank blocks (virtual ranks) the walks of the blocks run side by side on host threads (round 5): one thread
(PIC1DP_OPT_THREADS=1) against the default.
    python tools/opt_event_bench.py [markers] [blocks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd as amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
npe = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# two events of a kind, one step apart: the first one of a run also sets up the workers' streams and staging
EVENTS = [("merge", dict(nmerge=2, tmerge=[0.1, 0.15], thshmerge=[0.5, 0.6])),
          ("remove typeremove 2", dict(nremove=2, tremove=[0.1, 0.15], typeremove=2)),
          ("remove typeremove 1", dict(nremove=2, tremove=[0.1, 0.15], typeremove=1, thshremove=[0.4, 0.5], remove_frac=0.3)),
          ("split", dict(nsplit=2, tsplit=[0.1, 0.15], thshsplit=[0.3, 0.6], split_ngroup=2))]
os.environ["PIC1DP_OPT_TIMING"] = "1"
# (the first event of a process pays the one-off costs -- code objects, the first allocations -- for the others)
w = amd.Pic1dp(amd.make_input(nparticle_max=300000, species_nparticle_init=[200000], nx=256, nv=128, **EVENTS[0][1]), npe=npe)
w.particle_load()
w.interaction_collect_charge()
w.field_solve_electric()
w.step(2)
w.close()
for name, kw in EVENTS:
    for host, threads in ((0, None), (0, 1), (1, None)) if npe > 1 else ((0, None), (1, None)):
        os.environ["PIC1DP_OPT_HOST"] = str(host)
        if threads:
            os.environ["PIC1DP_OPT_THREADS"] = str(threads)
        else:
            os.environ.pop("PIC1DP_OPT_THREADS", None)
        e = amd.Pic1dp(amd.make_input(nparticle_max=n + n // 2, species_nparticle_init=[n], nx=256, nv=128, **kw), npe=npe)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.step(1)
        e.sync()
        b0 = e.kernel_stats(8)[1]
        n0 = e.local_sizes()[1]
        t0 = time.perf_counter()
        e.step(1)                     # 0.05 + dt >= 0.1: the event fires in this step
        e.sync()
        dt = time.perf_counter() - t0
        n1 = e.local_sizes()[1]
        b1 = e.kernel_stats(8)[1]
        t0 = time.perf_counter()
        e.step(1)                     # and the second one in the next
        e.sync()
        dt2 = time.perf_counter() - t0
        print("%-20s %-32s step with the event %8.1f ms   %6.2f B per marker over PCIe   markers %d -> %d"
              % (name, "host copies" if host else ("markers on the device" + (", %d thread" % threads if threads else "")),
                 dt * 1e3, (b1 - b0) / n, n0, n1), flush=True)
        print("%-20s %-32s second event of the run %8.1f ms   markers %d -> %d"
              % (name, "host copies" if host else ("markers on the device" + (", %d thread" % threads if threads else "")),
                 dt2 * 1e3, n1, e.local_sizes()[1]), flush=True)
        e.close()
