"""The four-step NTT split over ranks (dist.FourStepNTT): two ranks on
the test box's one GPU (gloo carries the all-to-all), each holding half of the vector; every rank's
block of the result must equal the same block of the single-GPU transform -- all four flag
combinations.  Also the world-size-1 path in-process."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, log_ns, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import FourStepNTT
    from oracle.cpu_oracle import CpuOracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = CpuOracle()
        ctx = pa.Context(0)
        ok = True
        for log_n in log_ns:
            n = 1 << log_n
            full = o.fr_sample(4242 + log_n, n)
            blk = n // world
            mine = torch.from_numpy(full[rank * blk:(rank + 1) * blk].view(np.int64).copy()).cuda()
            plan = FourStepNTT(ctx, log_n)
            for flags in (0, 1, 2, 3):
                got = plan(mine, flags).cpu().numpy().view(np.uint64)
                exp = ctx.fr_ntt(full, log_n, flags)[rank * blk:(rank + 1) * blk]
                ok = ok and bool(np.array_equal(got, exp))
            # PM_NTT_TRANSPOSED: the forward result stays in the [N1 / W][N2] matrix order of the last sub-transform
            # (one all-to-all less): X[k2 N1 + k1] at k1 N2 + k2; the inverse takes that order and returns natural order
            l1 = log_n // 2
            n1, n2 = 1 << l1, 1 << (log_n - l1)
            for cos in (0, 2):
                tr = plan(mine, cos | 4)
                nat = ctx.fr_ntt(full, log_n, cos).reshape(n2, n1, 4)            # [k2][k1]
                exp_t = np.ascontiguousarray(nat.transpose(1, 0, 2)).reshape(n, 4)[rank * blk:(rank + 1) * blk]
                ok = ok and bool(np.array_equal(tr.cpu().numpy().view(np.uint64), exp_t))
                back = plan(tr, cos | 1 | 4)
                ok = ok and bool(torch.equal(back, mine))
            assert torch.equal(mine.cpu(), torch.from_numpy(full[rank * blk:(rank + 1) * blk].view(np.int64)))   # input untouched
        q.put((rank, ok))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_ns", [(2, (4, 9, 14)), (4, (4, 7, 11, 13))])
def test_four_step_ntt_over_ranks(world, log_ns):
    """`world` ranks on the one GPU of the test box, the all-to-all through the callback (gloo); four ranks
    exercise the per-peer block indexing of the pack / unpack kernels that two ranks cannot tell apart."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, log_ns, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("log_n", [2, 3, 5, 7, 12, 16])
def test_four_step_ntt_single_rank(ctx, oracle, log_n):
    import torch
    from plonk_prototype_amd.dist import FourStepNTT
    n = 1 << log_n
    a = oracle.fr_sample(99 + log_n, n)
    x = torch.from_numpy(a.view(np.int64).copy()).cuda()
    plan = FourStepNTT(ctx, log_n)
    for flags in (0, 1, 2, 3):
        got = plan(x, flags).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, oracle.fr_ntt(a, log_n, flags, 4)), flags
    # forward then inverse is the identity
    back = plan(plan(x, 2), 3)
    assert torch.equal(back, x)
    # ... also through the block-transposed order (two transposes per transform instead of three)
    l1 = log_n // 2
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    for cos in (0, 2):
        tr = plan(x, cos | 4)
        nat = oracle.fr_ntt(a, log_n, cos, 4).reshape(n2, n1, 4)
        assert np.array_equal(tr.cpu().numpy().view(np.uint64), np.ascontiguousarray(nat.transpose(1, 0, 2)).reshape(n, 4)), cos
        assert torch.equal(plan(tr, cos | 1 | 4), x), cos


def test_four_step_ntt_2_20_equals_the_library_plan(ctx, oracle):
    """Full size on one rank against pm_fr_ntt_dev (the oracle would take too long here): bit for bit, all flags."""
    import torch
    from plonk_prototype_amd.dist import FourStepNTT
    log_n = 20
    n = 1 << log_n
    x = torch.from_numpy(oracle.fr_sample(5, n).view(np.int64).copy()).cuda()
    ref = torch.empty_like(x)
    plan = FourStepNTT(ctx, log_n)
    for flags in (0, 1, 2, 3):
        ctx.fr_ntt_dev(x.data_ptr(), n, ref.data_ptr(), log_n, flags)
        ctx.sync()
        assert torch.equal(plan(x, flags), ref), flags


def test_four_step_native_entry_point_checks_its_arguments(ctx):
    import torch
    import plonk_prototype_amd as pa
    x = torch.zeros((256, 4), dtype=torch.int64, device="cuda")
    st = torch.zeros((512, 4), dtype=torch.int64, device="cuda")
    for world, rank, log_n, flags in ((3, 0, 8, 0), (2, 2, 8, 0), (32, 0, 8, 0), (1, 0, 1, 0), (1, 0, 27, 0), (1, 0, 8, 8)):
        with pytest.raises(pa.Error):
            ctx.fr_ntt_fourstep_dev(x.data_ptr(), st.data_ptr(), log_n, world, rank, flags)
    with pytest.raises(pa.Error) as e:                      # two ranks, no callback, no communicator
        ctx.fr_ntt_fourstep_dev(x.data_ptr(), st.data_ptr(), 9, 2, 0, 0)
    assert e.value.code == -7


def test_four_step_rejects_bad_shapes(ctx):
    import torch
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import FourStepNTT
    plan = FourStepNTT(ctx, 8)
    with pytest.raises(pa.Error):
        plan(torch.zeros((100, 4), dtype=torch.int64, device="cuda"), 0)
