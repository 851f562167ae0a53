"""One rank of a multi-process run on ONE GPU (helper of test_gpu_exchange.py, not a
test): the charge of all ranks is summed by the one-hop exchange through
IPC-mapped memory.  Launched by torch.distributed.run; writes rank files."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pic1dp_amd  # noqa: E402
from pic1dp_amd import parallel  # noqa: E402


def main():
    import torch.distributed as dist
    out, kw, steps, mode = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    eng = pic1dp_amd.Pic1dp(pic1dp_amd.make_input(**kw), rank=rank, nranks=world, device=0)
    eng.particle_load()
    why = parallel.bootstrap_exchange(eng, dist)
    assert why is None, why
    eng.set_allreduce(2)
    dist.barrier()
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
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
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
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
