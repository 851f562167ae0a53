"""bench.py end to end on the GPU box: the default contract line on one rank and a
rehearsal of the multi-rank control flow -- two ranks launched by
torch.distributed.run sharing the one GPU, charge summed through the host-staged
path (RCCL needs one GPU per rank) -- whose physics must equal a one-process
run holding the same two reference rank blocks as virtual ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(args, nproc=1, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_line_small():
    d = run_bench(["--particles", "2000000", "--nx", "256", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert abs(d["value"] - 2_000_000 * 2 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["drop_in_call_sites"]["value"] > 0


def test_bench_two_ranks_share_the_gpu(amd):
    n_per, nx, steps, warm = 1_500_000, 128, 6, 2
    d = run_bench(["--gpus", "2", "--particles", str(n_per), "--nx", str(nx), "--steps", str(steps),
                   "--warmup", str(warm), "--force-host-allreduce", "--no-cpu-baseline"], nproc=2)
    assert d["n_gpus"] == 2 and d["config"]["particles_total"] == 2 * n_per
    assert d["config"]["allreduce"].startswith("host-staged")
    # the same global problem in one process: two reference rank blocks as virtual ranks
    eng = amd.Pic1dp(amd.make_input(nparticle_max=2 * n_per, nx=nx), npe=2)
    eng.particle_load()
    eng.interaction_collect_charge()
    eng.field_solve_electric()
    eng.step(d["settle_steps_before_warmup"] + warm + steps)
    assert abs(d["field_energy_end"] / eng.field_energy() - 1.0) < 1e-10
