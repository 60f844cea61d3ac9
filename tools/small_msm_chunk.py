import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
o = CpuOracle()
ctx = pa.Context(0)
for k, kb in ((12, 4), (12, 1), (10, 4), (14, 4)):
    n = 1 << k
    pts = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xabcdef123456789abcdef], 4)[0], n, 16)
    scs = np.concatenate([o.fr_sample(100 + j, n) for j in range(kb)])
    d = torch.from_numpy(np.ascontiguousarray(scs).view(np.int64)).cuda()
    bases = pa.host.Bases(ctx, pts).precompute()
    ref = None
    for chunk in (0, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        ctx.set_option("msm_chunk", chunk)
        r = bases.msm_batch_dev(d.data_ptr(), n, kb)
        if ref is None: ref = r
        assert np.array_equal(r, ref)
        best = 1e9
        for rep in range(3):
            ctx.sync(); ctx.profile(True)
            t0 = time.perf_counter()
            for _ in range(5): bases.msm_batch_dev(d.data_ptr(), n, kb)
            dt = (time.perf_counter() - t0) / 5
            prof = ctx.profile_read(); ctx.profile(False)
            if dt < best:
                best, ks = dt, {s.replace("msm_", ""): round(v[1] / 5 * 1e3) for s, v in prof.items()}
        print(f"2^{k} batch {kb} chunk={chunk:2d} {best*1e3:7.3f} ms  {ks}", flush=True)
    ctx.set_option("msm_chunk", 0)
    bases.free()
