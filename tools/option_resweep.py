"""GPU probe (not a test), r05: tunables whose defaults were measured in r01 - r03 against older kernels, re-measured with the r05 product
and tiles: ntt_xcd, ntt_direct_tw on the 2^20 / 2^24 transforms; msm_chunk, msm_lb on the 2^20 MSM with the table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
o = CpuOracle(); ctx = pa.Context(0)


def ntt_time(k, reps):
    n = 1 << k
    a = torch.from_numpy(o.fr_sample(k, n).view(np.int64)).cuda(); b = torch.empty_like(a)
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0); ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0); ctx.fr_ntt_dev(b.data_ptr(), n, b.data_ptr(), k, 1)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


for opt, vals in (("ntt_xcd", (1, 0, 1, 0)), ("ntt_direct_tw", (1, 0, 1, 0))):
    for v in vals:
        ctx.set_option(opt, v)
        print(f"{opt}={v}: 2^20 {ntt_time(20, 40):.1f} us   2^24 {ntt_time(24, 6):.1f} us per forward + inverse", flush=True)
    ctx.set_option(opt, 1)
k = 20; n = 1 << k
pts = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xabcdef123456789abcdef], 4)[0], n, 16)
sc = torch.from_numpy(o.fr_sample(0x5343414C, n).view(np.int64)).cuda()
bases = pa.host.Bases(ctx, pts).precompute()
ref = bases.msm_dev(sc.data_ptr(), n)
for opt, vals in (("msm_chunk", (0, 64, 80, 96, 104, 112, 128, 160, 208, 0)), ("msm_lb", (0, 8, 16, 32, 0))):
    for v in vals:
        ctx.set_option(opt, v)
        assert np.array_equal(bases.msm_dev(sc.data_ptr(), n), ref)
        best = 1e9
        for _ in range(3):
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(5): bases.msm_dev(sc.data_ptr(), n)
            best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
        print(f"{opt}={v}: 2^20 MSM {best:.3f} ms", flush=True)
    ctx.set_option(opt, 0)
