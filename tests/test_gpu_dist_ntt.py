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
            assert torch.equal(mine.cpu(), torch.from_numpy(full[rank * blk:(rank + 1) * blk].view(np.int64)))   # input untouched
        q.put((rank, ok))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_four_step_ntt_world2():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, (4, 9, 14), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("log_n", [2, 7, 12, 16])
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


def test_four_step_rejects_bad_shapes(ctx):
    import torch
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import FourStepNTT
    plan = FourStepNTT(ctx, 8)
    with pytest.raises(pa.Error):
        plan(torch.zeros((100, 4), dtype=torch.int64, device="cuda"), 0)
