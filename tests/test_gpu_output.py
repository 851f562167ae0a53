"""GPU tests of the rows SURVEY 8(f) marks next: N1 on-GPU diagnostics of
output_all (energy sums, (x,v) / v distributions) against the oracle, N2 the
pic1dp.out writer on a real run, and the Fortran host (flang, ISO_C_BINDING)
driving the same library."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from util import both_inputs, relerr

pytestmark = pytest.mark.gpu

DIST_KEYS = ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v")


def accumulated_times(dt, steps):
    """global_time is accumulated, time = time + dt (src/pic1dp.F90:93)"""
    t, out = 0.0, []
    for i in range(max(steps) + 1):
        if i in steps:
            out.append(t)
        t = t + dt
    return out


def started(oracle, amd, npe=1, steps=3, **kw):
    kw.setdefault("nparticle_max", 150001)
    kw.setdefault("nx", 64)
    o, g = both_inputs(oracle, amd, **kw)
    sim = oracle.Sim(o, npe=npe)
    assert sim.load() == 0
    eng = amd.Pic1dp(g, npe=npe)
    eng.particle_load()
    sim.collect_charge()
    sim.solve_field()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    for _ in range(steps):
        sim.step(1)
        eng.step(1)
    return sim, eng


@pytest.mark.parametrize("kw", [
    dict(),
    dict(linear=1),
    dict(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0]),
    dict(deltaf=0, iptcldist=3),
    dict(deltaf=0, iptcldist=2, species_density=[1.0], species_v0=[3.0]),
    dict(deltaf=0, iptcldist=1, species_density=[1.0]),
    dict(nx_opd=16, nv_opd=200),              # too large for LDS: global-atomic path
    dict(nx_opd=3, nv_opd=2),
], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()) or "default")
def test_ptcldist_and_output_scalars(oracle_mod, amd, kw):
    sim, eng = started(oracle_mod, amd, **kw)
    # put the oracle on exactly the engine's particles (they agree to ~1e-13 after
    # a few steps; the histograms are compared at matching inputs)
    g = eng.particles_download()
    for k in "xvpw":
        sim.array(0, 0, k)[:] = g[k]
    for finish in (False, True):
        d_g = eng.ptcldist(0, finish=finish)
        d_o = sim.ptcldist(0, finish=finish)
        for k in DIST_KEYS:
            scale = np.max(np.abs(d_o[k])) or 1.0
            assert np.max(np.abs(d_g[k] - d_o[k])) < 1e-11 * scale, (k, finish)
    sim.set_field(eng.get_field()["electric"])
    s_g, s_o = eng.output_scalars(), sim.output_scalars()
    assert s_g[0] == s_o[0]
    assert np.max(np.abs(s_g[1:] - s_o[1:]) / (np.abs(s_o[1:]) + 1e-300)) < 1e-10


def test_diagnostics_follow_the_markers(oracle_mod, amd):
    """histograms and kinetic sums come from one fused pass whose results are kept
    until the markers change: every way of changing them must be noticed"""
    sim, eng = started(oracle_mod, amd)

    def check(tag):
        g = eng.particles_download()
        for k in "xvpw":
            sim.array(0, 0, k)[:] = g[k]
        d_g, d_o = eng.ptcldist(0, finish=True), sim.ptcldist(0, finish=True)
        for k in DIST_KEYS:
            scale = np.max(np.abs(d_o[k])) or 1.0
            assert np.max(np.abs(d_g[k] - d_o[k])) < 1e-11 * scale, (tag, k)
        s_g, s_o = np.array(eng.energy_sums(0)), np.array(sim.energy_sums(0))
        assert np.max(np.abs(s_g - s_o) / np.abs(s_o)) < 1e-10, tag
        return d_g["pertb_xv"].copy()

    first = check("start")
    assert np.array_equal(check("cached"), first)
    eng.step(1)
    assert not np.array_equal(check("step"), first)
    eng.substep(1)
    check("substep 1")
    eng.substep(2)
    check("substep 2")
    eng.interaction_push_particle(1)
    check("push")
    g = eng.particles_download()
    g["w"] = g["w"] * 0.5
    eng.particles_upload(g["x"], g["v"], g["p"], g["w"])
    second = check("upload")
    assert not np.array_equal(second, first)


def test_ptcldist_conservation(amd):
    """bilinear weights sum to one: the marker histogram integrates to the
    number of markers with |v| < v_max, in (x,v) and in v"""
    eng = amd.Pic1dp(amd.make_input(nparticle_max=10**6, nx=128))
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.step(2)
    d = eng.ptcldist(0, finish=False)
    v = eng.particles_download()["v"]
    inside = np.count_nonzero(np.abs(v) < 8.0)
    assert abs(d["markr_xv"].sum() - inside) < 1e-6
    assert abs(d["markr_v"].sum() - inside) < 1e-6
    assert abs(d["total_xv"].sum() - d["total_v"].sum()) < 1e-9 * abs(d["total_v"].sum())


def test_writer_on_real_run_and_virtual_ranks(oracle_mod, amd, tmp_path):
    from pic1dp_amd import output
    o, g = both_inputs(oracle_mod, amd, nparticle_max=60000, nx=48, time_max=2.0)
    eng = amd.Pic1dp(g, npe=2)
    eng.particle_load()
    path = str(tmp_path / "pic1dp.out")
    lines = []
    with output.OutputWriter(path, g) as w:
        eng.run(on_output=lambda e: lines.append(output.progress_line(g, e.itime, e.time, w.write_record(e))))
    d = output.OutputData(path)
    assert d.ntime == 5 and list(d.scalars[:, 0]) == accumulated_times(0.05, [0, 10, 20, 30, 40])
    assert os.path.getsize(path) == output.header_bytes(g) + 5 * output.record_bytes(g)
    assert lines[0].startswith("i  0.0%      0    0.000")
    sim = oracle_mod.Sim(o, npe=2)
    sim.load()
    sim.collect_charge()
    sim.solve_field()
    want = [sim.output_scalars()]
    for _ in range(4):
        sim.step(10)
        want.append(sim.output_scalars())
    want = np.array(want)
    assert np.max(np.abs(d.scalars[:, 1] / want[:, 1] - 1.0)) < 1e-10       # int E^2 dx
    assert np.max(np.abs(d.scalars[:, 2:] / want[:, 2:] - 1.0)) < 1e-9      # kinetic sums
    assert relerr(d.electric[-1], sim.get_field()[0]) < 1e-10
    assert relerr(d.ptcldist[-1][0]["total_xv"].ravel(), sim.ptcldist()["total_xv"]) < 1e-9


def test_fortran_host_drop_in(oracle_mod, amd, tmp_path):
    """the Fortran driver (flang, ISO_C_BINDING) with the three reference call
    sites replaced by the C ABI: its pic1dp.out equals the oracle's run"""
    exe = os.path.join(ROOT, "pic1dp_amd", "fortran", "pic1dp_host")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.dirname(exe)], capture_output=True, text=True)
        if not os.path.exists(exe):
            flang = shutil.which("flang") or (os.path.exists("/opt/rocm/lib/llvm/bin/flang") and "/opt/rocm/lib/llvm/bin/flang")
            # a box with the Fortran compiler must build the host: a broken build is a failure,
            # not a skip (this is the only Fortran-through-the-ABI test)
            assert not flang, "Fortran host does not build although flang is present:\n" + r.stdout[-1500:] + r.stderr[-1500:]
            pytest.skip("no Fortran compiler on this box")
    from pic1dp_amd import output
    env = dict(os.environ, PIC1DP_NPARTICLE="80000", PIC1DP_NX="64", PIC1DP_TIME_MAX="1.0")
    outs = {}
    for fused in ("0", "1", "2"):        # three call sites / fused sub-step / whole-step kernels
        wd = tmp_path / ("fused" + fused)
        wd.mkdir()
        r = subprocess.run([exe], cwd=str(wd), env=dict(env, PIC1DP_FUSED=fused), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "progrss  itime     time  int E^2 dx" in r.stdout
        assert r.stdout.count("%") == 3                        # step 0, 10, 20
        outs[fused] = output.OutputData(str(wd / "pic1dp.out"))
    d = outs["0"]
    assert d.ntime == 3 and d.nx == 64 and list(d.scalars[:, 0]) == accumulated_times(0.05, [0, 10, 20])
    sim = oracle_mod.Sim(oracle_mod.make_input(nparticle_max=80000, nx=64, time_max=1.0))
    sim.load()
    sim.collect_charge()
    sim.solve_field()
    want = [sim.output_scalars()]
    for _ in range(2):
        sim.step(10)
        want.append(sim.output_scalars())
    want = np.array(want)
    for dd in outs.values():
        assert np.max(np.abs(dd.scalars[:, 1] / want[:, 1] - 1.0)) < 1e-10
        assert np.max(np.abs(dd.scalars[:, 2:] / want[:, 2:] - 1.0)) < 1e-9
        assert relerr(dd.electric[-1], sim.get_field()[0]) < 1e-10


def test_split_phase_diagnostics_for_a_host_that_owns_the_reduction(amd):
    """a host with its own MPI_Reduce (src/pic1dp_output.F90:126-151,333-356) takes the local sums, reduces them
    and hands them back: on one rank the two halves must compose to what the library writes itself -- delta-f,
    linear (total += pertb) and full-f (pertb = total - f0)"""
    for kw in (dict(), dict(linear=1), dict(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0])):
        eng = amd.Pic1dp(amd.make_input(nparticle_max=100_001, nx=64, **kw))
        eng.particle_load()
        eng.interaction_collect_charge()
        eng.field_solve_electric()
        eng.step(3)
        whole = eng.output_scalars()
        halves = eng.output_scalars_from(eng.energy_sums())
        assert np.array_equal(whole, halves), kw
        fin = eng.ptcldist(0, finish=True)
        raw = eng.ptcldist(0, finish=False)
        again = eng.ptcldist_finish(raw)
        for k in fin:
            assert np.array_equal(fin[k].ravel(), again[k]), (kw, k)
        # and twice the markers (two identical ranks) is twice the sums before the finishing, not after
        two = eng.ptcldist_finish({k: 2.0 * v for k, v in raw.items()})
        assert np.allclose(two["markr_xv"], 2.0 * fin["markr_xv"].ravel(), rtol=1e-14)


def fortran_host_exe():
    exe = os.path.join(ROOT, "pic1dp_amd", "fortran", "pic1dp_host")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.dirname(exe)], capture_output=True, text=True)
        if not os.path.exists(exe):
            flang = shutil.which("flang") or (os.path.exists("/opt/rocm/lib/llvm/bin/flang") and "/opt/rocm/lib/llvm/bin/flang")
            assert not flang, "Fortran host does not build although flang is present:\n" + r.stdout[-1500:] + r.stderr[-1500:]
            pytest.skip("no Fortran compiler on this box")
    return exe


@pytest.mark.parametrize("nranks,fused", [(2, "0"), (3, "2"), (4, "0"), (4, "3")],
                         ids=["two_ranks_call_sites", "three_ranks_whole_step", "four_ranks_call_sites", "four_ranks_steps_to_output"])
def test_fortran_host_as_one_of_n_ranks(oracle_mod, amd, tmp_path, nranks, fused):
    """the reference is an MPI program (src/pic1dp.F90:43-52; `mpiexec -n 4`, run/Makefile:41): N processes of the
    shipped Fortran host -- rank / size from the environment, each owning its PETSC_DECIDE block, the charge
    summed by the library's one-hop exchange (here between processes sharing the box's one GPU), the exchange
    handles all-gathered and the diagnostics reduced to rank 0 through host_ranks.F90 (files standing in for the
    MPI the image lacks) -- write on rank 0 the pic1dp.out of the N-rank oracle run.  Four ranks: the reference's own launch
    line (`NPE_RUN := 4`, Makefile:39) and the most this box admits with one process per rank (six processes on the card at
    once, this test's own included; the eight-rank exchange runs as four processes of two ranks, test_gpu_exchange.py)"""
    exe = fortran_host_exe()
    from pic1dp_amd import output
    rdv = tmp_path / "rendezvous"
    rdv.mkdir()
    env = dict(os.environ, PIC1DP_NPARTICLE="90001", PIC1DP_NX="64", PIC1DP_TIME_MAX="1.0", PIC1DP_FUSED=fused,
               PIC1DP_NRANKS=str(nranks), PIC1DP_RENDEZVOUS=str(rdv), HSA_ENABLE_IPC_MODE_LEGACY="0",
               PIC1DP_XCHG_TIMEOUT_MS="60000")
    procs = []
    for r in range(nranks):
        wd = tmp_path / ("rank%d" % r)
        wd.mkdir()
        procs.append(subprocess.Popen([exe], cwd=str(wd), env=dict(env, PIC1DP_RANK=str(r)), stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r])
    assert "progrss  itime     time  int E^2 dx" in outs[0] and outs[0].count("%") == 3
    for r in range(1, nranks):                                  # PetscPrintf: rank 0 only
        assert "%" not in outs[r] and not os.path.exists(str(tmp_path / ("rank%d" % r) / "pic1dp.out"))
    d = output.OutputData(str(tmp_path / "rank0" / "pic1dp.out"))
    assert d.ntime == 3 and d.nx == 64 and list(d.scalars[:, 0]) == accumulated_times(0.05, [0, 10, 20])
    sim = oracle_mod.Sim(oracle_mod.make_input(nparticle_max=90001, nx=64, time_max=1.0), npe=nranks)
    sim.load()
    sim.collect_charge()
    sim.solve_field()
    want = [sim.output_scalars()]
    for _ in range(2):
        sim.step(10)
        want.append(sim.output_scalars())
    want = np.array(want)
    assert np.max(np.abs(d.scalars[:, 1] / want[:, 1] - 1.0)) < 1e-10       # int E^2 dx
    assert np.max(np.abs(d.scalars[:, 2:] / want[:, 2:] - 1.0)) < 1e-9      # kinetic sums over all ranks
    assert relerr(d.electric[-1], sim.get_field()[0]) < 1e-10
    assert relerr(d.ptcldist[-1][0]["total_xv"].ravel(), sim.ptcldist()["total_xv"]) < 1e-9
    assert relerr(d.ptcldist[-1][0]["markr_v"].ravel(), sim.ptcldist()["markr_v"]) < 1e-9


@pytest.mark.parametrize("kw,mode", [
    (dict(), "step"), (dict(), "calls"), (dict(linear=1), "step"),
    (dict(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0]), "step"),
    (dict(iptcldist=3, species_temperature=[1.3], species_temperature2=[0.7], species_mass=[1.1]), "step"),
    (dict(nparticle_max=150_001, species_nparticle_init=[140_000]), "step")],
    ids=["default_step", "default_call_sites", "linear", "full_f", "general_constants_carry", "odd_with_tail_slots"])
def test_output_diagnostics_inside_the_step(amd, kw, mode):
    """with output fusion insisted on (2), the step that precedes output_all takes the histograms of
    output_ptcldist and the kinetic sums of output_field inside k_step_full (DIAG variant):
    no separate pass over the markers, same numbers as the separate pass
    (src/pic1dp_output.F90:126-151, :239-315) up to the order of the atomics"""
    base = dict(nparticle_max=300_000, nx=128, output_interval=0.5)
    base.update(kw)
    engs = []
    for fuse in (True, False):
        e = amd.Pic1dp(amd.make_input(**base))
        e.set_output_fusion(2 if fuse else 0)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        engs.append(e)
    fused, plain = engs
    nout = 0
    for chunk in (10, 10, 7, 3):              # outputs fall due after steps 10, 20, 30 (dt 0.05, interval 0.5)
        for e in engs:
            if mode == "step":
                e.step(chunk)
            else:
                for _ in range(chunk):
                    for irk in (1, 2):
                        e.interaction_push_particle(irk)
                        e.interaction_collect_charge()
                        e.field_solve_electric()
                    e.set_time(e.itime + 1, e.time + e.inp.dt)
        if not fused.output_due():
            continue
        nout += 1
        before = fused.kernel_stats(5)[1]
        a, b = fused.output_scalars(), plain.output_scalars()
        # two engines differ in the order of their charge atomics (trajectories at ~1e-15), and the
        # perturbed kinetic sum cancels heavily: compare against the scale of its terms, not its value
        tol = 1e-10 * np.abs(b)
        ns = (b.size - 2) // 3
        for isp in range(ns):                      # [sum v^2, total KE sum, perturbed KE sum] per species
            tol[4 + 3 * isp] = max(tol[4 + 3 * isp], 1e-12 * abs(b[3 + 3 * isp]))
        assert np.all(np.abs(a - b) <= tol), (a, b)
        pa, pb = fused.ptcldist(), plain.ptcldist()
        for k in pa:
            assert relerr(pa[k], pb[k]) < 1e-11, k
        assert fused.kernel_stats(5)[1] == before          # no k_ptcldist pass was needed
    assert nout == 3
    assert plain.kernel_stats(5)[1] == 3 and fused.kernel_stats(5)[1] == 0
    # and the markers are what they are without the fusion (two runs differ in the order of their
    # charge atomics, so not bit for bit)
    ga, gb = fused.particles_download(), plain.particles_download()
    for k in "xvw":
        assert np.max(np.abs(ga[k] - gb[k])) < 1e-10 * max(1.0, np.max(np.abs(gb[k]))), k
    fused.kernel_stats_enable(True)        # (from here on: which kernels does a further output step run?)
    fused.set_time(0, 0.0)
    if mode == "step":
        fused.step(10)
        # the output step is k_step_full<DIAG>; having lost the prediction to the previous one, the first step is two passes
        assert fused.kernel_stats(4)[1] == 1 and fused.kernel_stats(6)[1] == 9 and fused.kernel_stats(3)[1] == 1


@pytest.mark.parametrize("mode", ["step", "calls"])
def test_output_steps_keep_the_prediction(oracle_mod, amd, mode):
    """with output fusion 'where it pays' (1: what the Fortran host and Pic1dp.run ask for) a
    predicted one-pass step that output_all follows stays k_step_one -- the diagnostics take a pass of their own, which
    changes no marker, so the prediction survives and the step after the output is one pass again: ONE first-sub-step
    pass in the whole run (the first step's), one diagnostics pass per record, and the records against the oracle
    (src/pic1dp_output.F90:126-151, 239-315)"""
    kw = dict(nparticle_max=300_000, nx=128, output_interval=0.5)
    eng = amd.Pic1dp(amd.make_input(**kw))
    eng.set_output_fusion(1)
    eng.kernel_stats_enable(True)
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    nout = 0
    for chunk in (10, 10, 10, 10):
        if mode == "step":
            eng.step(chunk)
        else:
            for _ in range(chunk):
                for irk in (1, 2):
                    eng.interaction_push_particle(irk)
                    eng.interaction_collect_charge()
                    eng.field_solve_electric()
                eng.set_time(eng.itime + 1, eng.time + eng.inp.dt)
        sim.step(chunk)
        assert eng.output_due()
        nout += 1
        a, b = eng.output_scalars(), np.array(sim.output_scalars())
        assert abs(a[1] / b[1] - 1.0) < 1e-10                       # int E^2 dx
        assert np.max(np.abs(a[2:4] / b[2:4] - 1.0)) < 1e-9         # sum v^2, total kinetic energy
        pa, pb = eng.ptcldist(), sim.ptcldist()
        for k in ("markr_xv", "total_xv", "markr_v", "total_v"):
            assert relerr(np.asarray(pa[k]).ravel(), np.asarray(pb[k]).ravel()) < 1e-9, k
    assert eng.kernel_stats(5)[1] == nout            # one diagnostics pass per record
    assert eng.kernel_stats(3)[1] == 1               # k_step_half: the first step of the run only
    assert eng.kernel_stats(6)[1] == 40 and eng.kernel_stats(4)[1] == 0    # every step k_step_one, no k_step_full


@pytest.mark.parametrize("fused", ["0", "3"], ids=["call_sites", "whole_steps"])
def test_fortran_host_with_rccl(amd, tmp_path, fused):
    """the charge sum the reference makes with MPI_Allreduce (src/pic1dp_interaction.F90:130-135;
    communicator set-up src/pic1dp.F90:43-52) as an RCCL all-reduce reached FROM THE FORTRAN HOST: PIC1DP_ALLREDUCE=rccl
    -- rank 0 draws the unique id, host_ranks.F90 broadcasts it, every rank calls pic1dp_hip_comm_init.  One GPU can run
    it with the one-rank communicator (every launch of the N-rank step: marker kernel with the packing in its tail,
    ncclAllReduce, paired solve): against the plain run of the same host, markers bit for bit where the charge sums
    have one order (one wave of markers), the records to 1e-10 at a realistic count."""
    exe = fortran_host_exe()
    from pic1dp_amd import output
    for n, exact in (("96", True), ("80000", False)):
        res = {}
        for how in ("plain", "rccl"):
            wd = tmp_path / ("%s_%s_%s" % (how, n, fused))
            wd.mkdir()
            env = dict(os.environ, PIC1DP_NPARTICLE=n, PIC1DP_NX="64", PIC1DP_TIME_MAX="1.0", PIC1DP_FUSED=fused,
                       PIC1DP_DUMP_MARKERS=str(wd / "markers.bin"))
            if how == "rccl":
                env["PIC1DP_ALLREDUCE"] = "rccl"
            r = subprocess.run([exe], cwd=str(wd), env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            m = np.fromfile(str(wd / "markers.bin"))
            assert int(m[0]) == int(n) and m.size == 1 + 4 * int(n)
            res[how] = (output.OutputData(str(wd / "pic1dp.out")), m, open(str(wd / "pic1dp.out"), "rb").read())
        (da, ma, ra), (db, mb, rb) = res["plain"], res["rccl"]
        assert da.ntime == 3 and db.ntime == 3
        if exact and fused == "3":
            assert np.array_equal(ma, mb)                       # x, v, p, w of every marker
            assert ra == rb                                     # and pic1dp.out byte for byte
        elif exact:
            # (through the call sites the plain one-rank run takes its half-step field from the pair solve, the RCCL run
            # from the kept mode's content of the predicted charge density solved again: the same field to rounding)
            assert np.max(np.abs(ma - mb)) < 1e-12 * np.max(np.abs(ma))
            assert np.max(np.abs(da.scalars[:, 1] / db.scalars[:, 1] - 1.0)) < 1e-12
        else:
            assert np.max(np.abs(da.scalars[:, 1] / db.scalars[:, 1] - 1.0)) < 1e-10
            assert np.max(np.abs(da.scalars[:, 2:] / db.scalars[:, 2:] - 1.0)) < 1e-9
            assert relerr(da.electric[-1], db.electric[-1]) < 1e-10
            assert np.max(np.abs(ma - mb)) < 1e-9


def test_fortran_host_default_workload_slice(oracle_mod, amd, tmp_path):
    """BASELINE configs[0]'s workload -- the reference's default input, src/pic1dp_input.F90:35,109,113,
    128,250: 6.4e6 markers, nx 192, dt 0.05, output_all every 0.5 -- through the shipped Fortran host as the reference
    driver runs it (src/pic1dp.F90:78-109; pic1dp.out layout src/pic1dp_output.F90:74-92,173-186), its first 1 000 steps
    (the whole 10 000 with 1 001 records: tools/default_run.py, profiles/r05/default_run_10000_steps.log): 101 records, 44 +
    101 x 103 000 bytes, through the three call sites and as whole steps batched up to each output; the two against each
    other over the whole slice, the first 50 steps against the oracle, and the growth of the linear phase under way."""
    exe = fortran_host_exe()
    from pic1dp_amd import output
    inp = amd.make_input(time_max=50.0)
    outs = {}
    for fused in ("0", "3"):
        wd = tmp_path / ("fused" + fused)
        wd.mkdir()
        env = dict(os.environ, PIC1DP_TIME_MAX="50.0", PIC1DP_FUSED=fused, PIC1DP_HOST_PROFILE="1")
        for k in ("PIC1DP_NPARTICLE", "PIC1DP_NX"):
            env.pop(k, None)
        r = subprocess.run([exe], cwd=str(wd), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert r.stdout.count("%") == 101 and "101 records)" in r.stdout
        path = str(wd / "pic1dp.out")
        assert os.path.getsize(path) == output.header_bytes(inp) + 101 * output.record_bytes(inp) == 44 + 101 * 103000
        outs[fused] = output.OutputData(path)
        os.remove(path)
    a, b = outs["0"], outs["3"]
    assert a.ntime == b.ntime == 101 and a.nx == 192
    assert list(a.scalars[:, 0]) == accumulated_times(0.05, list(range(0, 1001, 10)))
    assert np.max(np.abs(a.scalars[:, 1] / b.scalars[:, 1] - 1.0)) < 1e-10          # int E^2 dx, t <= 50: the linear phase
    assert np.max(np.abs(a.scalars[:, 2:4] / b.scalars[:, 2:4] - 1.0)) < 1e-10
    sim = oracle_mod.Sim(oracle_mod.make_input(time_max=50.0))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    want = [sim.output_scalars()]
    for _ in range(5):
        sim.step(10)
        want.append(sim.output_scalars())
    want = np.array(want)
    for d in (a, b):
        assert np.max(np.abs(d.scalars[:6, 1] / want[:, 1] - 1.0)) < 1e-10
        assert np.max(np.abs(d.scalars[:6, 2:4] / want[:, 2:4] - 1.0)) < 1e-9
    assert relerr(a.electric[5], sim.get_field()[0]) < 1e-10
    g2 = a.growthrate_energy_fit(15.0, 45.0)
    assert abs(g2 / 0.16766 - 1.0) < 0.03, g2


@pytest.mark.parametrize("kw", [dict(), dict(linear=1), dict(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0]),
                                dict(nparticle_max=150_001, species_nparticle_init=[140_000])],
                         ids=["default", "linear", "full_f", "odd_with_tail_slots"])
def test_fixed_point_histograms_equal_the_double_sums(oracle_mod, amd, monkeypatch, kw):
    """round 5: from the second record on k_ptcldist sums the (x, v) histograms as 64-bit fixed-point numbers in the LDS
    (ds_add_u64 runs 1.9x the rate of ds_add_f64 at random bins; scale per plane from the species' max |p|, max |w| of the
    pass before) -- a term rounded once to 2^-44 of its plane's bound, the sums exact and independent of the atomics'
    order: against the double sums (PIC1DP_DIAG_FX=0) 1e-12 of the largest bin, against the oracle as before; markers
    beyond the bounds (the weights scaled up behind the library's back) make the pass repeat in doubles"""
    base = dict(nparticle_max=300_000, nx=128, output_interval=0.5)
    base.update(kw)
    engs = []
    for fx in ("1", "0"):
        monkeypatch.setenv("PIC1DP_DIAG_FX", fx)
        e = amd.Pic1dp(amd.make_input(**base))
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        engs.append(e)
    a, b = engs
    sim = oracle_mod.Sim(oracle_mod.make_input(**base))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    for rec in range(4):
        pa, pb, po = a.ptcldist(), b.ptcldist(), sim.ptcldist()
        for k in pa:
            scale = max(np.max(np.abs(pb[k])), 1e-300)
            assert np.max(np.abs(pa[k] - pb[k])) <= 1e-12 * scale, (rec, k)
            if rec > 0 or True:
                assert relerr(np.asarray(pa[k]).ravel(), np.asarray(po[k]).ravel()) < 1e-9, (rec, k)
        assert np.allclose(a.output_scalars(), b.output_scalars(), rtol=1e-12, atol=0.0)
        for e in (a, b):
            e.step(10)
        sim.step(10)
    fx_passes, repeats = a.kernel_stats(12)[1], a.kernel_stats(12)[0]
    assert fx_passes == 3 and repeats == 0 and b.kernel_stats(12)[1] == 0        # the first record: bounds unknown, doubles
    # an upload voids the bounds: the next record is summed in doubles, the one after in fixed point again
    g = a.particles_download()
    npv = a.local_sizes()[1]
    a.particles_upload(g["x"], g["v"], g["p"], g["w"], np_valid=npv)
    b.particles_upload(g["x"], g["v"], g["p"], g["w"], np_valid=npv)
    pa, pb = a.ptcldist(), b.ptcldist()
    assert a.kernel_stats(12)[1] == 3
    for k in pa:
        assert np.max(np.abs(pa[k] - pb[k])) <= 1e-12 * max(np.max(np.abs(pb[k])), 1e-300), k


def test_fixed_point_histograms_repeat_in_doubles_on_overflow(amd, monkeypatch):
    """a marker beyond the bounds a fixed-point pass was scaled for (here: the margin on max |w| set below one, so that the
    largest weights of the very next record exceed it) makes the pass report it and the library repeat the pass with
    double sums: the record is right either way"""
    monkeypatch.setenv("PIC1DP_DIAG_FX_MARGIN", "0.5")
    a = amd.Pic1dp(amd.make_input(nparticle_max=300_000, nx=128))
    monkeypatch.setenv("PIC1DP_DIAG_FX", "0")
    b = amd.Pic1dp(amd.make_input(nparticle_max=300_000, nx=128))
    for e in (a, b):
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
    for rec in range(3):
        pa, pb = a.ptcldist(), b.ptcldist()
        for k in pa:
            assert np.max(np.abs(pa[k] - pb[k])) <= 1e-12 * max(np.max(np.abs(pb[k])), 1e-300), (rec, k)
        for e in (a, b):
            e.step(10)
    assert a.kernel_stats(12)[1] == 2 and a.kernel_stats(12)[0] == 2.0     # both fixed-point passes overflowed and were repeated


@pytest.mark.parametrize("margin", [None, "0.5"], ids=["in_bounds", "overflow_repeats"])
@pytest.mark.parametrize("predict", ["1", "0"], ids=["one_pass", "two_passes"])
def test_fixed_point_histograms_inside_the_step_kernel(oracle_mod, amd, monkeypatch, predict, margin):
    """the diagnostics taken INSIDE the step output_all follows (k_step_full<DIAG>; set_output_fusion 2 always, 1 where a
    step is two passes anyway) sum their histograms as 64-bit fixed-point numbers as well, scaled from the record before:
    against the same engine with double sums 1e-12 of the largest bin and against the oracle; with the margin on max |w|
    below one the next record's largest weights are beyond the bounds: the collector notices and repeats in doubles"""
    kw = dict(nparticle_max=300_000, nx=128, output_interval=0.5)
    monkeypatch.setenv("PIC1DP_PREDICT", predict)
    if margin:
        monkeypatch.setenv("PIC1DP_DIAG_FX_MARGIN", margin)
    engs = []
    for fx in ("1", "0"):
        monkeypatch.setenv("PIC1DP_DIAG_FX", fx)
        e = amd.Pic1dp(amd.make_input(**kw))
        e.set_output_fusion(2 if predict == "1" else 1)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
        e.kernel_stats_enable(True)
        engs.append(e)
    a, b = engs
    sim = oracle_mod.Sim(oracle_mod.make_input(**kw))
    assert sim.load() == 0
    sim.collect_charge()
    sim.solve_field()
    for rec in range(4):
        pa, pb, po = a.ptcldist(), b.ptcldist(), sim.ptcldist()
        for k in pa:
            scale = max(np.max(np.abs(pb[k])), 1e-300)
            assert np.max(np.abs(pa[k] - pb[k])) <= 1e-12 * scale, (rec, k)
            assert relerr(np.asarray(pa[k]).ravel(), np.asarray(po[k]).ravel()) < 1e-9, (rec, k)
        assert np.allclose(a.output_scalars(), b.output_scalars(), rtol=1e-12, atol=0.0)
        for e in (a, b):
            e.step(10)                      # the tenth step is followed by output_all: it takes the diagnostics along
        sim.step(10)
    fx_passes, repeats = a.kernel_stats(12)[1], a.kernel_stats(12)[0]
    # record 0: the diagnostics' own pass (no step before it), bounds unknown: doubles.  Records 1 .. 3 and the step after
    # the last look: inside the step kernel, in fixed point from the moment the bounds are known
    assert fx_passes >= 3 and b.kernel_stats(12)[1] == 0, (fx_passes, repeats)
    assert (repeats >= 2.0) if margin else (repeats == 0.0), repeats
    assert a.kernel_stats(5)[1] == 1 + int(repeats)   # k_ptcldist: record 0, and the repeats' double passes


@pytest.mark.parametrize("kw", [dict(), dict(linear=1), dict(deltaf=0, iptcldist=0, species_density=[1.0], species_v0=[0.0]),
                                dict(nparticle_max=150_001, species_nparticle_init=[140_000]),
                                dict(nspecies=2, species_charge=[-1.0, 1.0], species_mass=[1.0, 4.0], species_temperature=[1.0, 1.0],
                                     species_temperature2=[1.0, 1.0], species_density=[0.9, 0.9], species_v0=[5.0, 5.0],
                                     species_nparticle_init=[120_000, 90_000])],
                         ids=["default", "linear", "full_f", "odd_with_tail_slots", "two_species"])
def test_output_all_in_one_call_equals_the_separate_calls(amd, kw):
    """pic1dp_hip_output_all -- the record of output_all (src/pic1dp_output.F90:100-189, 196-477) in one call and one wait
    -- against output_scalars + get_field + ptcldist of every species on a twin engine: the same numbers (the passes'
    atomics in another order: 1e-12), over several records and with steps in between"""
    base = dict(nparticle_max=200_000, nx=96, output_interval=0.5)
    base.update(kw)
    a, b = amd.Pic1dp(amd.make_input(**base)), amd.Pic1dp(amd.make_input(**base))
    for e in (a, b):
        e.set_output_fusion(1)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
    ns = base.get("nspecies", 1)
    for rec in range(3):
        scal, fld, dists = a.output_all()
        want_scal, want_fld = b.output_scalars(), b.get_field()
        tol = 1e-11 * np.abs(want_scal)
        for isp in range(ns):                     # the perturbed kinetic sum cancels heavily: against the scale of its terms
            tol[4 + 3 * isp] = max(tol[4 + 3 * isp], 1e-12 * abs(want_scal[3 + 3 * isp]))
        assert np.all(np.abs(scal - want_scal) <= tol), (rec, scal, want_scal)
        for k in ("electric", "chargeden", "mode_re", "mode_im"):
            assert relerr(fld[k], want_fld[k]) < 1e-11, (rec, k)
        for isp in range(ns):
            want = b.ptcldist(isp)
            for k in want:
                assert np.max(np.abs(dists[isp][k] - want[k])) <= 1e-11 * max(np.max(np.abs(want[k])), 1e-300), (rec, isp, k)
        # asked for again without a step in between: served from the cached pass
        passes = a.kernel_stats(5)[1]
        scal2, _, _ = a.output_all()
        assert a.kernel_stats(5)[1] == passes and np.array_equal(scal2, scal)
        for e in (a, b):
            e.step(10)


@pytest.mark.parametrize("two_species", [False, True], ids=["one_species", "two_species"])
def test_output_all_record_survives_a_fixed_point_repeat(amd, monkeypatch, two_species):
    """when a fixed-point diagnostics pass meets a marker beyond its bounds, pic1dp_hip_output_all repeats
    that species' pass in doubles -- and the repeat must not stage through the part of the pinned record that still holds
    E, chargeden, the modes, int E^2 dx and the other species' sums.  The margin on max |w| below one makes every record
    after the first overflow; fields, scalars and histograms against a twin engine with double sums and the separate calls"""
    base = dict(nparticle_max=200_000, nx=96, output_interval=0.5)
    if two_species:
        base.update(nspecies=2, species_charge=[-1.0, 1.0], species_mass=[1.0, 4.0], species_temperature=[1.0, 1.0],
                    species_temperature2=[1.0, 1.0], species_density=[0.9, 0.9], species_v0=[5.0, 5.0],
                    species_nparticle_init=[120_000, 90_000])
    monkeypatch.setenv("PIC1DP_DIAG_FX_MARGIN", "0.5")
    a = amd.Pic1dp(amd.make_input(**base))
    monkeypatch.setenv("PIC1DP_DIAG_FX", "0")
    b = amd.Pic1dp(amd.make_input(**base))
    for e in (a, b):
        e.set_output_fusion(1)
        e.particle_load()
        e.interaction_collect_charge()
        e.field_solve_electric()
    ns = base.get("nspecies", 1)
    for rec in range(3):
        scal, fld, dists = a.output_all()
        want_scal, want_fld = b.output_scalars(), b.get_field()
        tol = 1e-11 * np.abs(want_scal)
        for isp in range(ns):
            tol[4 + 3 * isp] = max(tol[4 + 3 * isp], 1e-12 * abs(want_scal[3 + 3 * isp]))
        assert np.all(np.abs(scal - want_scal) <= tol), (rec, scal, want_scal)
        for k in ("electric", "chargeden", "mode_re", "mode_im"):
            assert relerr(fld[k], want_fld[k]) < 1e-11, (rec, k)
        for isp in range(ns):
            want = b.ptcldist(isp)
            for k in want:
                assert np.max(np.abs(dists[isp][k] - want[k])) <= 1e-11 * max(np.max(np.abs(want[k])), 1e-300), (rec, isp, k)
        for e in (a, b):
            e.step(10)
    assert a.kernel_stats(12)[0] >= 2.0          # the records after the first one were repeated in doubles


def test_fortran_host_writes_its_buffered_record_before_a_stop(amd, tmp_path):
    """the host assembles a record in memory and writes it while the next steps run on the GPU; a library error in
    between (here forced: a call the library refuses, at the step right after an output) must not lose the record the
    reference would have written by then -- pic1dp_hip_check runs the host's abort hook before it stops"""
    exe = fortran_host_exe()
    from pic1dp_amd import output
    env = dict(os.environ, PIC1DP_NPARTICLE="50001", PIC1DP_NX="64", PIC1DP_TIME_MAX="2.0", PIC1DP_FUSED="0",
               PIC1DP_HOST_FAIL_AT="10")                # output_all at steps 0, 10, 20, ...: the failure comes right behind record 1
    r = subprocess.run([exe], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout[-1500:] + r.stderr[-1500:]
    assert "push (forced failure)" in r.stdout and "irk must be 1 or 2" in r.stdout
    d = output.OutputData(str(tmp_path / "pic1dp.out"))
    assert d.ntime == 2 and list(d.scalars[:, 0]) == accumulated_times(0.05, [0, 10])       # both records whole
    inp = amd.make_input(nparticle_max=50001, nx=64)
    assert os.path.getsize(str(tmp_path / "pic1dp.out")) == output.header_bytes(inp) + 2 * output.record_bytes(inp)
