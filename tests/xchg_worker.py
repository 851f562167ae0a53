"""Ranks of a multi-rank run on ONE GPU (helper of test_gpu_exchange.py, not a
test): the charge of all ranks is summed by the one-hop exchange through
IPC-mapped memory.  Launched by torch.distributed.run; writes rank files.

PIC1DP_RANKS_PER_PROC=k (default 1): every process hosts k consecutive ranks, one
context and one host thread each -- ranks of one process reach each other's exchange
areas directly, ranks of other processes through hipIpc (include/pic1dp_hip.h).  A GPU
box admits six processes on its card at a time, the test's own included: the target's
eight ranks are four processes of two."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pic1dp_amd  # noqa: E402
from pic1dp_amd import parallel  # noqa: E402


def connect(engs, dist, world):
    """every rank's 64-byte handle to every rank, in rank order; k handles per process"""
    import torch
    if len(engs) == 1:
        why = parallel.bootstrap_exchange(engs[0], dist)
        assert why is None, why
        return
    mine = torch.tensor(list(b"".join(e.xchg_create() for e in engs)), dtype=torch.uint8)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    handles = b"".join(bytes(g.tolist()) for g in gathered)
    assert len(handles) == 64 * world
    for e in engs:
        e.xchg_connect(handles)


def rank_body(eng, rank, out, steps, mode, errors):
    try:
        eng.interaction_collect_charge()
        eng.field_solve_electric()
        e0 = eng.field_energy()
        fields = []
        if mode == "timeout":
            # rank 1 stops one step early: rank 0's kernels must give up waiting (bounded), finish, and
            # the next synchronising call must report the missing rank
            err = ""
            try:
                eng.step(steps if rank == 0 else steps - 1)
                eng.sync()
            except pic1dp_amd.Pic1dpError as e:
                err = "%d|%s" % (e.code, e)
            np.savez(out + ".rank%d.npz" % rank, err=err)
            return
        if mode == "step":
            eng.step(steps)
        else:  # the reference's call sites
            for _ in range(steps):
                for irk in (1, 2):
                    eng.interaction_push_particle(irk)
                    eng.interaction_collect_charge()
                    eng.field_solve_electric()
                    fields.append(eng.get_field()["electric"].copy())
        eng.sync()
        f = eng.get_field()
        memkind, nx = eng.xchg_info()
        np.savez(out + ".rank%d.npz" % rank, e0=e0, hist=eng.energy_history(), E=f["electric"], cd=f["chargeden"],
                 fields=np.array(fields), energy=eng.field_energy(), memkind=memkind, exchanges=nx,
                 x=eng.particles_download()["x"], tails=eng.kernel_stats(10)[1])
    except BaseException as e:      # noqa: BLE001  (reported by the main thread)
        errors.append("rank %d: %r" % (rank, e))


def main():
    import torch.distributed as dist
    out, kw, steps, mode = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    proc, nproc = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    per = int(os.environ.get("PIC1DP_RANKS_PER_PROC", "1"))
    world = nproc * per
    dist.init_process_group(backend="gloo", rank=proc, world_size=nproc)
    ranks = [proc * per + i for i in range(per)]
    engs = [pic1dp_amd.Pic1dp(pic1dp_amd.make_input(**kw), rank=r, nranks=world, device=0) for r in ranks]
    for e in engs:
        e.particle_load()
    connect(engs, dist, world)
    for e in engs:
        e.set_allreduce(2)
    dist.barrier()
    errors = []
    if per == 1:
        rank_body(engs[0], ranks[0], out, steps, mode, errors)
    else:  # one host thread per rank, as one MPI process each would be (the library's calls release the GIL: ctypes)
        threads = [threading.Thread(target=rank_body, args=(e, r, out, steps, mode, errors)) for e, r in zip(engs, ranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    dist.barrier()      # nobody frees an exchange area a peer's kernels may still store into
    for e in engs:
        e.close()
    dist.barrier()
    dist.destroy_process_group()
    if errors:
        sys.exit("\n".join(errors))


if __name__ == "__main__":
    main()
