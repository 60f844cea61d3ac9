"""pm_trim: a context's workspaces and twiddle caches only ever grow with the largest call it has seen; a long-lived
service gives them back with pm_trim and everything is rebuilt on demand, with the same results."""
import numpy as np
import pytest

from oracle.cpu_oracle import ints_to_limbs

pytestmark = pytest.mark.gpu

INVERSE, COSET = 1, 2


def test_trim_gives_memory_back_and_everything_rebuilds(oracle):
    import plonk_prototype_amd as pa
    import torch
    from test_gpu_prover import _srs
    ctx = pa.Context(0)
    try:
        k = 18
        n = 1 << k
        a = oracle.fr_sample(31, n)
        pts = oracle.g1_bases_arith(ints_to_limbs([77], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], 5000, 4)
        sc = oracle.fr_sample(32, 5000)
        bases = pa.host.Bases(ctx, pts).precompute(13)
        pt = oracle.fr_sample(33, 1)[0]
        d_a = pa.DeviceVector.from_host(ctx, a)
        circuit, wit, pi = pa.synthetic.mixed_circuit(256, 6)
        ck = pa.CommitKey(_srs(oracle, 256), ctx, precompute=True)
        pk = pa.prover.preprocess(circuit, ctx, ck)

        def everything():
            return (ctx.fr_ntt(a, k, 0), ctx.fr_ntt(a, k, INVERSE | COSET), bases.msm(sc), ctx.fr_evaluate(d_a.ptr, n, pt),
                    pa.prover.prove(pk, ck, wit, pi).to_bytes())

        first = everything()
        free0 = torch.cuda.mem_get_info()[0]
        freed = ctx.trim()
        # two pass buffers of n x 36 B, the host-call staging, the MSM workspace, the twiddle tables of the domain ...
        assert freed >= 2 * n * 36
        assert torch.cuda.mem_get_info()[0] >= free0 + freed // 2
        assert ctx.trim() == 0                                       # nothing left to give back
        again = everything()                                         # tables and workspaces come back on demand
        for x, y in zip(first, again):
            assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y
        # resident objects survive: the bases' table, the prover key
        ctx.trim()
        assert np.array_equal(bases.msm(sc[:100]), pa.host.Bases(ctx, pts[:100]).msm(sc[:100]))
        assert pa.prover.prove(pk, ck, wit, pi).to_bytes() == first[4]
    finally:
        ctx.close()
