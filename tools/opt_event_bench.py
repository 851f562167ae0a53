#!/usr/bin/env python3
"""opt_event_bench.py -- one marker optimisation event (src/pic1dp_particle.F90:356-813) at 1e7 markers: the markers
staying on the device (keys to the host, decisions back: kernels_opt.hip) against the pass on host copies
(PIC1DP_OPT_HOST=1); wall time of the step that carries the event and bytes between host and device.
    python tools/opt_event_bench.py [markers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pic1dp_amd as amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
EVENTS = [("merge", dict(nmerge=1, tmerge=[0.1], thshmerge=[0.5])),
          ("remove typeremove 2", dict(nremove=1, tremove=[0.1], typeremove=2)),
          ("remove typeremove 1", dict(nremove=1, tremove=[0.1], typeremove=1, thshremove=[0.4], remove_frac=0.7)),
          ("split", dict(nsplit=1, tsplit=[0.1], thshsplit=[0.3], split_ngroup=3))]
for name, kw in EVENTS:
    for host in (0, 1):
        os.environ["PIC1DP_OPT_HOST"] = str(host)
        e = amd.Pic1dp(amd.make_input(nparticle_max=n + n // 2, species_nparticle_init=[n], nx=256, nv=128, **kw))
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.step(1)
        e.sync()
        b0 = e.kernel_stats(8)[1]
        t0 = time.perf_counter()
        e.step(1)                     # 0.05 + dt >= 0.1: the event fires in this step
        e.sync()
        dt = time.perf_counter() - t0
        print("%-20s %-22s step with the event %8.1f ms   %6.2f B per marker over PCIe   markers %d -> %d"
              % (name, "host copies" if host else "markers on the device", dt * 1e3, (e.kernel_stats(8)[1] - b0) / n, n,
                 e.local_sizes()[1]), flush=True)
        e.close()
