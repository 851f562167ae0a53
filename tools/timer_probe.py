import sys, time
sys.path.insert(0, "/root/repo")
import pic1dp_amd
for n, nx in ((6400000, 192),):
    for every in (1, 2, 3, 5, 17, 64, 0):
        eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(nparticle_max=n, nx=nx))
        eng.particle_load(); eng.interaction_collect_charge(); eng.field_solve_electric()
        eng.step(50); eng.sync()
        eng.timers_enable(every)
        t0 = time.perf_counter()
        for _ in range(100):
            eng.step(10)
        eng.sync()
        wall = (time.perf_counter() - t0) / 1000 * 1e6
        print("step(10) x100  every %2d: wall %.1f us/step  push %.1f  field %.1f us/step" % (every, wall, eng.timer_ms(4), eng.timer_ms(7)), flush=True)
        eng.timers_reset(); eng.sync()
        t0 = time.perf_counter()
        for _ in range(1000):
            for irk in (1, 2):
                eng.interaction_push_particle(irk); eng.interaction_collect_charge(); eng.field_solve_electric()
        eng.sync()
        wall = (time.perf_counter() - t0) / 1000 * 1e6
        print("call sites     every %2d: wall %.1f us/step  push %.1f  collect %.1f  field %.1f us/step" % (every, wall, eng.timer_ms(4), eng.timer_ms(6), eng.timer_ms(7)), flush=True)
        eng.close()
