"""One-hop charge exchange (pic1dp_hip_xchg_*, kernels_field.hip exchange_charge): N
processes share the box's one GPU and map each other's exchange areas through IPC
handles -- the same code path as N GPUs over xGMI, minus the link.  Checks: the
exchange runs, the summed charge / field are BIT-identical on all ranks (fixed
rank-order sum), and the physics equals a one-process run holding the same reference
rank blocks as virtual ranks (replaces MPI_Allreduce, src/pic1dp_interaction.F90:130-135)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_ranks(tmp_path, nproc, kw, steps, mode, port, timeout_ms="60000", per=1):
    """nproc processes of `per` ranks each (one context and one host thread per rank)"""
    out = str(tmp_path / ("xchg_%s_%d" % (mode, nproc)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PIC1DP_XCHG_TIMEOUT_MS=timeout_ms,
               PIC1DP_RANKS_PER_PROC=str(per))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "xchg_worker.py"),
           out, json.dumps(kw), str(steps), mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [np.load(out + ".rank%d.npz" % k) for k in range(nproc * per)]


@pytest.mark.parametrize("nproc,mode,kind,nx", [(2, "step", 1, 128), (3, "step", 1, 128), (2, "calls", 1, 128),
                                                (3, "step", 2, 128), (2, "calls", 2, 128),
                                                (2, "step", 1, 2048), (2, "step", 2, 4096)],
                         ids=["2-step", "3-step", "2-calls", "3-step-sums", "2-calls-sums", "2-step-nx2048", "2-step-sums-nx4096"])
def test_exchange_ranks_share_the_gpu(amd, tmp_path, monkeypatch, nproc, mode, kind, nx):
    """kind 2: the one-pass prediction travels as six sums behind charge2 (k_step_sums, the large-grid kernel,
    insisted on at the small grid); the large grids: the paired solve's exchange vector beyond 64 KiB of LDS"""
    kw = dict(nparticle_max=600_000, nx=nx)
    steps = 12
    monkeypatch.setenv("PIC1DP_PRED_KIND", str(kind))
    ranks = run_ranks(tmp_path, nproc, kw, steps, mode, 29541 + nproc + 10 * kind + (nx > 128) * 20)
    # every rank holds the same field, bit for bit
    for r in ranks[1:]:
        assert np.array_equal(r["E"], ranks[0]["E"])
        assert np.array_equal(r["cd"], ranks[0]["cd"])
        assert np.array_equal(r["hist"], ranks[0]["hist"])
        if mode == "calls":
            assert np.array_equal(r["fields"], ranks[0]["fields"])
    # step(): the initial deposit, the first step's own first sub-step, then ONE exchange per step (the
    # new state's charge and the prediction of the next first sub-step's travel together); through the
    # call sites: the initial deposit and one exchange per collect_charge
    expect = 2 + steps if mode == "step" else 1 + 2 * steps
    assert all(int(r["exchanges"]) == expect for r in ranks)
    # six sums, step(): this rank's half of the exchange -- charge2 and the sums formed, stored into every rank's slots,
    # flagged -- rides in the tail of the marker launch (kernels.hpp StepTail); the field launch only waits and adds
    assert all(int(r["tails"]) == (steps if (mode == "step" and kind == 2) else 0) for r in ranks)
    # the exchange areas are fine-grained device memory (coherent across agents inside a kernel),
    # not one of the fall-backs
    assert all(int(r["memkind"]) == 1 for r in ranks)
    # one process, the same reference rank blocks as virtual ranks
    eng = amd.Pic1dp(amd.make_input(**kw), npe=nproc)
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    e0 = eng.field_energy()
    eng.step(steps)
    assert abs(float(ranks[0]["e0"]) / e0 - 1.0) < 1e-12
    if mode == "step":
        assert np.max(np.abs(ranks[0]["hist"] / eng.energy_history() - 1.0)) < 1e-10
    assert abs(float(ranks[0]["energy"]) / eng.field_energy() - 1.0) < 1e-10
    x = eng.particles_download()["x"]
    got = np.concatenate([r["x"] for r in ranks])
    assert np.max(np.abs(got - x)) < 1e-9


def test_exchange_gives_up_on_a_missing_rank(amd, tmp_path):
    """a peer that stops delivering: the waiting rank's kernels time out (PIC1DP_XCHG_TIMEOUT_MS), every
    launch still finishes, and the next synchronising call returns PIC1DP_ERR_COMM naming the rank --
    never a hang (the spin is bounded by a wall clock inside the kernel)"""
    import time
    t0 = time.time()
    ranks = run_ranks(tmp_path, 2, dict(nparticle_max=400_000, nx=64), 3, "timeout", 29549, timeout_ms="4000")
    assert time.time() - t0 < 120
    e0, e1 = str(ranks[0]["err"]), str(ranks[1]["err"])
    assert e0.startswith("5|") and "rank 1" in e0, e0        # PIC1DP_ERR_COMM on the rank that waited
    assert e1 == ""                                          # the rank that stopped early saw nothing wrong


def test_exchange_posted_from_the_marker_launch_equals_the_field_launchs_own(amd, tmp_path, monkeypatch):
    """with PIC1DP_TAIL=0 the field launch forms, stores and flags this rank's vector itself (as
    before round 5); with the tail the last workgroup of the marker launch has done so.  Same accumulators, same
    additions in the same order: with one wave of markers per rank the two runs agree bit for bit; at a marker count
    whose charge atomics arrive in varying order, to 1e-11.  Three ranks share the box's GPU."""
    monkeypatch.setenv("PIC1DP_PRED_KIND", "2")
    steps = 10
    for n, tol in ((3 * 96, 0.0), (450_000, 1e-11)):
        kw = dict(nparticle_max=n, nx=64)
        da, db = tmp_path / ("tail_%d" % n), tmp_path / ("plain_%d" % n)
        da.mkdir()
        db.mkdir()
        monkeypatch.setenv("PIC1DP_TAIL", "1")
        a = run_ranks(da, 3, kw, steps, "step", 29601 + (n > 1000))
        monkeypatch.setenv("PIC1DP_TAIL", "0")
        b = run_ranks(db, 3, kw, steps, "step", 29611 + (n > 1000))
        assert all(int(r["tails"]) == steps for r in a) and all(int(r["tails"]) == 0 for r in b)
        for ra, rb in zip(a, b):
            assert np.array_equal(ra["E"], a[0]["E"]) and np.array_equal(rb["E"], b[0]["E"])
            if tol == 0.0:
                for k in ("E", "cd", "hist", "x"):
                    assert np.array_equal(ra[k], rb[k]), k
            else:
                assert np.max(np.abs(ra["hist"] / rb["hist"] - 1.0)) < tol
                assert np.max(np.abs(ra["x"] - rb["x"])) < 1e-10


@pytest.mark.parametrize("nproc,per,mode,nx", [(4, 2, "step", 64), (4, 2, "calls", 64), (4, 2, "step", 1024), (4, 2, "calls", 1024),
                                               (4, 1, "step", 64), (2, 3, "step", 96)],
                         ids=["8-ranks-step-nx64", "8-ranks-calls-nx64", "8-ranks-step-nx1024", "8-ranks-calls-nx1024",
                              "4-processes-step", "6-ranks-2x3-step"])
def test_exchange_at_the_targets_rank_count(amd, tmp_path, nproc, per, mode, nx):
    """the target machine is 8 GPUs and `make run` starts four ranks (Makefile:39) -- eight exchange
    slots and flags, the 8-lane poll of exchange_wait_sum, the 8-rank summation order of the solve, eight tails posting into
    eight areas, through step() and through the three call sites.  A GPU box admits SIX processes on its card at once (this
    test's own included), so eight separate processes cannot run here: the eight ranks are FOUR processes of TWO ranks, each
    rank a context with its own host thread -- every rank still reaches six of its seven peers through hipIpc, the seventh
    directly (include/pic1dp_hip.h: ranks of one process).  Also four processes of one rank (one short of the limit, on
    purpose) and two of three.  Bit-identical E / chargeden / energy history on all ranks; 1e-10 against one engine holding the same reference
    blocks as virtual ranks (replaces MPI_Allreduce, src/pic1dp_interaction.F90:130-135)."""
    world = nproc * per
    kw = dict(nparticle_max=10_000 * world + 3, nx=nx)        # (not a multiple of the rank count: PETSC_DECIDE blocks of two sizes)
    steps = 6
    ranks = run_ranks(tmp_path, nproc, kw, steps, mode, 29700 + world * 7 + (nx > 100) * 3 + (mode == "calls"), per=per)
    assert len(ranks) == world
    for r in ranks[1:]:
        for k in ("E", "cd", "hist") + (("fields",) if mode == "calls" else ()):
            assert np.array_equal(r[k], ranks[0][k]), k
    expect = 2 + steps if mode == "step" else 1 + 2 * steps
    assert all(int(r["exchanges"]) == expect for r in ranks)
    assert all(int(r["tails"]) == (steps if mode == "step" else 0) for r in ranks)   # the six sums' tail posts (one kept mode)
    assert all(int(r["memkind"]) == 1 for r in ranks)
    eng = amd.Pic1dp(amd.make_input(**kw), npe=world)
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    e0 = eng.field_energy()
    eng.step(steps)
    assert abs(float(ranks[0]["e0"]) / e0 - 1.0) < 1e-12
    if mode == "step":
        assert np.max(np.abs(ranks[0]["hist"] / eng.energy_history() - 1.0)) < 1e-10
    assert abs(float(ranks[0]["energy"]) / eng.field_energy() - 1.0) < 1e-10
    f = eng.get_field()
    assert np.max(np.abs(ranks[0]["E"] - f["electric"])) <= 1e-10 * np.max(np.abs(f["electric"]))
    assert np.max(np.abs(ranks[0]["cd"] - f["chargeden"])) <= 1e-10 * np.max(np.abs(f["chargeden"]))
    x = eng.particles_download()["x"]
    got = np.concatenate([r["x"] for r in ranks])
    assert got.shape == x.shape and np.max(np.abs(got - x)) < 1e-9
