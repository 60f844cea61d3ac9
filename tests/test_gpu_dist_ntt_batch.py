"""pm_fr_ntt_fourstep_batch_dev (r05): several vectors through ONE sequence of all-to-alls, and the halo row of the
block-transposed forward result -- what the distributed prover's rounds are built on.  Ranks are threads of
this process (dist.LocalGroup); every rank's blocks must equal the same blocks of the single-GPU transforms, for every
flag combination, and the exchange counter must show the batch going out in one call per transpose step."""
import numpy as np
import pytest

from test_gpu_world8 import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,log_n,batch", [(1, 6, 3), (1, 11, 2), (2, 4, 2), (2, 9, 5), (4, 7, 3), (4, 12, 20), (8, 6, 2), (8, 11, 4)])
def test_batched_four_step(ctx, oracle, world, log_n, batch):
    import torch
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import DistGroup
    n = 1 << log_n
    blk = n // world
    l1 = log_n // 2
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    full = [oracle.fr_sample(900 + 17 * v + log_n, n) for v in range(batch)]
    ref = {f: [ctx.fr_ntt(x, log_n, f) for x in full] for f in (0, 1, 2, 3)}

    def body(r, g):
        c = pa.Context(0)
        try:
            grp = DistGroup(rank=r, local=g)
            cb = grp._aa if world > 1 else None
            mine = np.concatenate([x[r * blk:(r + 1) * blk] for x in full])
            stage = torch.empty((2 * batch * (blk + n2), 4), dtype=torch.int64, device="cuda")
            halo = torch.zeros((batch * n2, 4), dtype=torch.int64, device="cuda")
            out = {}
            for f in (0, 1, 2, 3):
                x = torch.from_numpy(mine.view(np.int64).copy()).cuda()
                c.comm_stats(reset=True)
                c.fr_ntt_fourstep_batch_dev(x.data_ptr(), batch, stage.data_ptr(), log_n, world, r, f, cb)
                c.sync()
                st = c.comm_stats()
                assert st["alltoall_calls"] == (3 if world > 1 else 0), st
                assert st["alltoall_bytes"] == (3 * batch * blk * 32 * (world - 1) // world if world > 1 else 0), st
                out[f] = x.cpu().numpy().view(np.uint64)
            for cos in (0, 2):
                x = torch.from_numpy(mine.view(np.int64).copy()).cuda()
                c.comm_stats(reset=True)
                c.fr_ntt_fourstep_batch_dev(x.data_ptr(), batch, stage.data_ptr(), log_n, world, r, cos | 4, cb, d_halo=halo.data_ptr())
                c.sync()
                assert c.comm_stats()["alltoall_calls"] == (2 if world > 1 else 0)
                out[cos | 4] = x.cpu().numpy().view(np.uint64).copy()
                out[("halo", cos)] = halo.cpu().numpy().view(np.uint64).copy()
                c.fr_ntt_fourstep_batch_dev(x.data_ptr(), batch, stage.data_ptr(), log_n, world, r, cos | 1 | 4, cb)
                c.sync()
                out[("back", cos)] = x.cpu().numpy().view(np.uint64)
            return mine, out
        finally:
            c.close()
    for r, (mine, out) in enumerate(run_ranks(world, body)):
        for f in (0, 1, 2, 3):
            exp = np.concatenate([y[r * blk:(r + 1) * blk] for y in ref[f]])
            assert np.array_equal(out[f], exp), (r, f)
        for cos in (0, 2):
            # block-transposed order: X[k2 N1 + k1] at k1 N2 + k2; rank r holds rows k1 in [r N1 / W, (r + 1) N1 / W)
            tr = [np.ascontiguousarray(y.reshape(n2, n1, 4).transpose(1, 0, 2)).reshape(n, 4) for y in ref[cos]]
            assert np.array_equal(out[cos | 4], np.concatenate([t[r * blk:(r + 1) * blk] for t in tr])), (r, cos)
            nxt = ((r + 1) % world) * blk        # the next rank's first row (rank 0's for the last rank), unshifted
            assert np.array_equal(out[("halo", cos)], np.concatenate([t[nxt:nxt + n2] for t in tr])), (r, cos, "halo")
            assert np.array_equal(out[("back", cos)], mine), (r, cos, "inverse of the transposed order")


def test_batched_four_step_with_more_sub_transforms_than_a_grid_dimension(ctx, oracle):
    """1100 vectors of 2^12 points on one rank: 70400 sub-transforms of 64 points per step -- more than the 65535 a launch takes
    on its batch dimension (a 2^24-gate distributed proof on ONE rank has 20 x 4096 of them): the library splits the step."""
    import torch
    log_n, batch = 12, 1100
    n = 1 << log_n
    x = oracle.fr_sample(77, n * batch)
    d = torch.from_numpy(x.view(np.int64).copy()).cuda()
    stage = torch.empty((batch * n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_ntt_fourstep_batch_dev(d.data_ptr(), batch, stage.data_ptr(), log_n, 1, 0, 2, None)
    ctx.sync()
    got = d.cpu().numpy().view(np.uint64)
    for v in (0, 1, 511, 512, 1023, 1099):
        assert np.array_equal(got[v * n:(v + 1) * n], ctx.fr_ntt(x[v * n:(v + 1) * n], log_n, 2)), v


def test_batched_four_step_rejects_bad_arguments(ctx):
    import torch
    import plonk_prototype_amd as pa
    x = torch.zeros((4 * 64, 4), dtype=torch.int64, device="cuda")
    stage = torch.zeros((2 * 4 * 80, 4), dtype=torch.int64, device="cuda")
    halo = torch.zeros((4 * 8, 4), dtype=torch.int64, device="cuda")
    for flags in (0, 1, 1 | 4):                         # a halo row exists only for a forward transposed transform
        with pytest.raises(pa.Error):
            ctx.fr_ntt_fourstep_batch_dev(x.data_ptr(), 4, stage.data_ptr(), 6, 1, 0, flags, None, d_halo=halo.data_ptr())
    with pytest.raises(pa.Error):
        ctx.fr_ntt_fourstep_batch_dev(x.data_ptr(), 0, stage.data_ptr(), 6, 1, 0, 0, None)
