import os, sys
sys.path.insert(0, ".")
import pic1dp_amd
n = int(float(sys.argv[1])); nx = int(sys.argv[2]); steps = int(sys.argv[3])
eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
eng.particle_load(); eng.interaction_collect_charge(); eng.field_solve_electric(); eng.step(40)
names = {0: "plain/plain", 1: "nt/nt", 2: "half nt, full plain", 3: "half plain, full nt"}
for rnd in range(2):
    for f in (0, 1, 2, 3):
        os.environ["PIC1DP_NT_FORCE"] = str(f)
        eng.step(3); eng.sync(); eng.kernel_stats_enable(True); eng.timers_reset()
        eng.step(steps); eng.sync()
        (hm, hn), (fm, fn) = eng.kernel_stats(3), eng.kernel_stats(4)
        print("n=%g round %d %-20s: step_half %.4f ms  step_full %.4f ms" % (n, rnd, names[f], hm / hn, fm / fn), flush=True)
