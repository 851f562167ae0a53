"""The N > 1 path on CPU with torch.distributed (gloo, world_size 2): block
ownership, per-rank RNG streams, the charge all-reduce and the communicator
bootstrap.  The GPU kernels cannot run here, so per-rank deposits are done by
the oracle (the checker) -- what is under test is the host logic that decides
who owns what and how the partial charges are combined."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import torch
    import torch.distributed as dist

    import oracle
    import pic1dp_amd
    from pic1dp_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, nx, npe = 40003, 32, 4            # 2 processes x 2 reference blocks each
        kw = dict(nparticle_max=n, nx=nx)
        ginp = pic1dp_amd.make_input(**kw)
        oinp = oracle.make_input(**kw)
        L = pic1dp_amd._lib.load()
        charge2 = np.zeros(nx)
        np_owned = 0
        for blk in parallel.owned_blocks(rank, world, npe):
            nalloc = parallel.local_size(n, blk, npe)
            arrs = [np.empty(nalloc) for _ in range(4)]
            pic1dp_amd._lib.check(L.pic1dp_hip_host_particle_load(
                C.byref(ginp), blk, npe, *[a.ctypes.data_as(C.c_void_p) for a in arrs], nalloc))
            x, v, p, w = arrs
            npv = parallel.block_np(n, n, blk, npe)
            c1 = np.zeros(nx)
            oracle.lib().orc_deposit_species(C.byref(oinp), npv, x, w, c1)
            charge2 += c1 * ginp.species_charge[0]
            np_owned += npv
        t = torch.from_numpy(charge2.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)           # MPI_Allreduce, src/pic1dp_interaction.F90:132
        cnt = torch.tensor([np_owned])
        dist.all_reduce(cnt)
        chargeden = np.empty(nx)
        oracle.lib().orc_chargeden_from_charge(C.byref(oinp), t.numpy(), chargeden)

        # communicator bootstrap: rank 0 draws the id, everyone receives the same bytes
        class FakeEngine:
            def __init__(self):
                self.rank, self.nranks, self.got = rank, world, None

            def comm_unique_id(self):
                return bytes((i * 7 + 3) % 256 for i in range(128))

            def comm_available(self):
                pass

            def comm_init(self, uid):
                self.got = uid

            # one-hop exchange: every rank's handle reaches every rank, in rank order
            def xchg_create(self):
                return bytes((rank * 64 + i) % 256 for i in range(64))

            def xchg_connect(self, handles):
                self.handles = handles

        fe = FakeEngine()
        assert parallel.bootstrap_comm(fe, dist) is None
        assert parallel.bootstrap_exchange(fe, dist) is None
        assert fe.handles == b"".join(bytes((r * 64 + i) % 256 for i in range(64)) for r in range(world))

        # failures are decided collectively: no rank enters comm_init / connect, none hangs
        class NoId(FakeEngine):
            def comm_unique_id(self):
                raise RuntimeError("no librccl here")

        class OneRankBlind(FakeEngine):
            def comm_available(self):
                if rank == 1:
                    raise RuntimeError("cannot load")

            def xchg_create(self):
                if rank == 1:
                    raise RuntimeError("no ipc")
                return FakeEngine.xchg_create(self)

        for cls in (NoId, OneRankBlind):
            e = cls()
            assert parallel.bootstrap_comm(e, dist) is not None and e.got is None

        class InitFailsEverywhere(FakeEngine):      # e.g. two ranks on one GPU: RCCL says "invalid usage"
            def comm_init(self, uid):
                raise RuntimeError("ncclCommInitRank: invalid usage")

        why = parallel.bootstrap_comm(InitFailsEverywhere(), dist)
        assert why is not None and "invalid usage" in why
        e = OneRankBlind()
        assert parallel.bootstrap_exchange(e, dist) is not None and not hasattr(e, "handles")
        q.put((rank, chargeden, int(cnt.item()), fe.got))
    finally:
        dist.destroy_process_group()


def test_two_process_charge_allreduce_matches_four_rank_oracle(oracle_mod):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, nx = 40003, 32
    sim = oracle_mod.Sim(oracle_mod.make_input(nparticle_max=n, nx=nx), npe=4)
    sim.load()
    sim.collect_charge()
    want = sim.get_field()[1]
    for rank, chargeden, cnt, uid in res:
        assert cnt == n
        assert np.max(np.abs(chargeden - want)) < 1e-13 * np.max(np.abs(want))
        assert uid == bytes((i * 7 + 3) % 256 for i in range(128))
    assert np.array_equal(res[0][1], res[1][1])            # identical on every rank


def test_bootstrap_comm_single_rank_is_noop(amd):
    class E:
        rank, nranks = 0, 1

        def comm_unique_id(self):
            raise AssertionError("must not be called")

    amd.parallel.bootstrap_comm(E())
