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


def _on_backend_device(t, dist):
    """tensors of a torch.distributed collective live where the backend wants them"""
    if dist.get_backend() == "nccl":
        import torch
        return t.to(torch.device("cuda", torch.cuda.current_device()))
    return t


def agree(dist, ok):
    """True when `ok` holds on EVERY rank (MIN all-reduce): the ranks of a job take
    the same branch, whatever failed wherever"""
    import torch
    flag = _on_backend_device(torch.tensor([1 if ok else 0], dtype=torch.int32), dist)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.cpu().item()) == 1


def bootstrap_comm(engine, dist=None):
    """Create the engine's RCCL communicator.  The bootstrap cannot leave a subset of
    the ranks inside a collective of the host's process group: rank 0 ALWAYS broadcasts
    a payload (ok flag + the 128-byte unique id, zeros when it could not draw one), every
    rank adds whether it can load librccl, and only when all agree does any rank enter
    ncclCommInitRank; afterwards the ranks agree once more on its outcome.  Returns None
    on success, or the reason (the same decision on every rank) why RCCL is not used --
    the caller then takes another charge sum on all ranks alike.  If ncclCommInitRank
    fails on SOME ranks only, the others are still inside it: the failed ranks wait in
    the agreement until the process group's timeout raises, and the launcher tears the
    job down (bounded, never a silent hang).  No-op (None) for a single process."""
    if engine.nranks == 1:
        return None
    import torch
    if dist is None:
        import torch.distributed as dist
    from ._lib import COMM_ID_BYTES
    payload = torch.zeros(1 + COMM_ID_BYTES, dtype=torch.uint8)
    reason = None
    if engine.rank == 0:
        try:
            payload[1:] = torch.tensor(list(engine.comm_unique_id()), dtype=torch.uint8)
            payload[0] = 1
        except Exception as e:      # noqa: BLE001  (reported, decided collectively below)
            reason = "rank 0 could not draw a unique id: %s" % e
    payload = _on_backend_device(payload, dist)
    dist.broadcast(payload, src=0)
    payload = payload.cpu()
    mine = int(payload[0]) == 1
    if mine:
        try:
            engine.comm_available()
        except Exception as e:      # noqa: BLE001
            mine, reason = False, "rank %d cannot load librccl: %s" % (engine.rank, e)
    if not agree(dist, mine):
        return reason or "RCCL unavailable on another rank"
    try:
        engine.comm_init(bytes(payload[1:].tolist()))
    except Exception as e:          # noqa: BLE001
        mine, reason = False, "rank %d: %s" % (engine.rank, e)
    if not agree(dist, mine):
        return reason or "ncclCommInitRank failed on another rank"
    return None


def bootstrap_exchange(engine, dist=None):
    """Connect the one-hop charge exchange (pic1dp_hip_xchg_*): every rank creates its
    exchange area, the 64-byte IPC handles are all-gathered, every rank maps its peers.
    Returns None on success or the agreed reason for not using it (same decision on
    every rank; nothing is left half-connected: the exchange is only switched on by
    set_allreduce(2) afterwards).  No-op (None) for a single process."""
    if engine.nranks == 1:
        return None
    import torch
    if dist is None:
        import torch.distributed as dist
    from ._lib import XCHG_HANDLE_BYTES
    reason = None
    mine = torch.zeros(1 + XCHG_HANDLE_BYTES, dtype=torch.uint8)
    try:
        mine[1:] = torch.tensor(list(engine.xchg_create()), dtype=torch.uint8)
        mine[0] = 1
    except Exception as e:          # noqa: BLE001
        reason = "rank %d: %s" % (engine.rank, e)
    mine = _on_backend_device(mine, dist)
    gathered = [torch.zeros_like(mine) for _ in range(engine.nranks)]
    dist.all_gather(gathered, mine)
    gathered = [g.cpu() for g in gathered]
    ok = all(int(g[0]) == 1 for g in gathered)
    if ok:
        try:
            engine.xchg_connect(b"".join(bytes(g[1:].tolist()) for g in gathered))
        except Exception as e:      # noqa: BLE001
            ok, reason = False, "rank %d: %s" % (engine.rank, e)
    if not agree(dist, ok):
        return reason or "exchange set-up failed on another rank"
    return None
