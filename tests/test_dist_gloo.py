"""world_size-2 `gloo` run of the multi-GPU MSM path on CPU: point sharding, the all-gather of
144-byte partial points and the group-law fold (product host code).  The per-rank partial MSM
is the GPU kernel in production; here the oracle stands in for it so that the collective and
the fold are exercised without a device."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import plonk_prototype_amd  # noqa: F401
    from plonk_prototype_amd.dist import allgather_fold, shard_range
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = CpuOracle()
        k0 = ints_to_limbs([99], 4)[0]
        d = ints_to_limbs([0x1000000000001], 4)[0]
        pts = o.g1_bases_arith(k0, d, n, 1)
        sc = o.fr_sample(31337, n)
        lo, hi = shard_range(n, rank, world)
        part_aff = o.g1_msm(pts[lo:hi], sc[lo:hi])
        part = np.zeros(18, np.uint64)
        one = o.fp_to_mont(ints_to_limbs([1], 6))[0]
        if part_aff.any():
            part[:12] = part_aff
            part[12:] = one
        else:
            part[6:12] = one
        total = allgather_fold(part)
        # the batched form the sharded prover uses: k partial points per rank in one all_gather
        from plonk_prototype_amd.dist import allgather_fold_many
        ident = np.zeros(18, np.uint64)
        ident[6:12] = one
        many = allgather_fold_many(np.stack([part, ident, part]))
        assert many.shape == (3, 18) and np.array_equal(many[0], total) and np.array_equal(many[2], total)
        assert not many[1][12:].any()                                  # identity + identity = identity (Z = 0)
        # a rank whose local work failed still enters the collective (abort marker): every rank learns of it
        assert allgather_fold_many(None if rank == 1 else np.stack([part])) is None
        q.put((rank, total.tolist(), o.g1_msm(pts, sc).tolist(), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 2])
def test_sharded_msm_fold_world2(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = sorted(r[3] for r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n
    for rank, total, full, _ in res:
        assert total[:12] == full and total[12:] != [0] * 6, rank   # same point on every rank, Z = 1


def test_shard_range_partitions():
    from plonk_prototype_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 5):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_local_group_collectives_on_the_cpu():
    """dist.LocalGroup (ranks = threads of one process), the transport of the world-8 rehearsals and of the distributed
    prover's callbacks, by itself on the CPU: the fixed-size message all-gather, the host-staged all-to-all and the
    gather + fold with its abort marker at world 8 -- and the DistGroup all-gather callback driven through ctypes as
    the library drives it."""
    import ctypes as C
    import threading
    import plonk_prototype_amd as pa
    from plonk_prototype_amd import _lib
    from plonk_prototype_amd.dist import DistGroup, LocalGroup
    from oracle.cpu_oracle import CpuOracle, ints_to_limbs
    world = 8
    g = LocalGroup(world, timeout=60)
    o = CpuOracle()
    G = o.g1_generator()
    one = o.fp_to_mont(ints_to_limbs([1], 6))[0]
    res = [None] * world

    def body(r):
        words = np.arange(_lib.COMM_MSG_WORDS, dtype=np.uint64) + 1000 * r
        gathered = g.allgather_words(r, words)
        send = (np.arange(world * 5, dtype=np.uint64) + 100 * r)           # block p goes to rank p
        recv = g.alltoall_host(r, send)
        part = np.zeros((2, 18), np.uint64)
        part[0, :12], part[0, 12:] = o.g1_mul(G, ints_to_limbs([r + 1], 4)[0]), one
        part[1, 6:12] = one                                                    # identity
        folded = g.allgather_fold_many(r, part)
        gave_up = g.allgather_fold_many(r, None if r == 3 else part)
        # the callback the library calls (pm_allgather_fn): msg in, world x msg out
        grp = DistGroup(rank=r, local=g)
        msg = (C.c_uint64 * _lib.COMM_MSG_WORDS)(*([7 + r] * _lib.COMM_MSG_WORDS))
        out = (C.c_uint64 * (_lib.COMM_MSG_WORDS * world))()
        rc = grp._ag(None, msg, out)
        res[r] = (gathered, recv, folded, gave_up, rc, [out[p * _lib.COMM_MSG_WORDS] for p in range(world)])
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
        assert not t.is_alive()
    for r, (gathered, recv, folded, gave_up, rc, firsts) in enumerate(res):
        assert gathered.shape == (world, _lib.COMM_MSG_WORDS) and all(gathered[p, 5] == 5 + 1000 * p for p in range(world))
        assert recv.tolist() == [100 * p + 5 * r + i for p in range(world) for i in range(5)]
        assert np.array_equal(pa.g1_to_affine(folded[0])[0], o.g1_mul(G, ints_to_limbs([36], 4)[0])) and pa.g1_to_affine(folded[1])[1]
        assert gave_up is None and rc == 0 and firsts == [7 + p for p in range(world)]
