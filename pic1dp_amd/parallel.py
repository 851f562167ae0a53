"""Particle decomposition and communicator bootstrap (host logic, no compute).

The reference has exactly one parallel strategy: contiguous PETSC_DECIDE blocks
of the particle arrays per MPI rank, a replicated grid, and a sum all-reduce of
the deposited charge (src/pic1dp_particle.F90:89-94,129;
src/pic1dp_interaction.F90:132).  Here one process drives one GPU; the
all-reduce is RCCL inside libpic1dp_hip.so, and this module only distributes
the RCCL unique id over whatever torch.distributed backend the host uses.
"""
import numpy as np


def local_size(nglobal, rank, size):
    """PETSC_DECIDE ownership: n/size + (rank < n%size)"""
    return nglobal // size + (1 if nglobal % size > rank else 0)


def block_np(nparticle_max, nparticle_init, mype, npe):
    """particle_np of reference rank mype (src/pic1dp_particle.F90:240-248)"""
    spare = nparticle_max - nparticle_init
    unload = spare // npe + (spare % npe if mype == 0 else 0)
    return local_size(nparticle_max, mype, npe) - unload


def owned_blocks(rank, nranks, npe=0):
    """reference ranks (blocks) reproduced by process `rank` of `nranks`"""
    npe = npe or nranks
    if npe % nranks:
        raise ValueError("npe must be a multiple of nranks")
    per = npe // nranks
    return list(range(rank * per, (rank + 1) * per))


def block_offsets(nglobal, npe):
    """global index of the first slot of every reference block (+ total)"""
    sizes = [local_size(nglobal, r, npe) for r in range(npe)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def bootstrap_comm(engine, dist=None):
    """Create the engine's RCCL communicator: rank 0 draws the unique id, every
    rank receives it through torch.distributed (any backend, gloo included) and
    calls comm_init.  No-op for a single process."""
    if engine.nranks == 1:
        return
    import torch
    if dist is None:
        import torch.distributed as dist
    if engine.rank == 0:
        uid = torch.tensor(list(engine.comm_unique_id()), dtype=torch.uint8)
    else:
        uid = torch.zeros(128, dtype=torch.uint8)
    backend = dist.get_backend()
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        uid = uid.to(dev)
        dist.broadcast(uid, src=0)
        uid = uid.cpu()
    else:
        dist.broadcast(uid, src=0)
    engine.comm_init(bytes(uid.tolist()))
