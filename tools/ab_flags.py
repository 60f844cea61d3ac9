"""A/B of library builds (compiler scheduling flags): run once per build with PM_LIB_PATH set.
Prints NTT 2^20 / 2^24 throughput and the 2^20 MSM (table) time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402

ctx = pa.Context(0)
tag = os.environ.get("PM_LIB_PATH", "default")
rng = np.random.default_rng(1)
for k, steps in ((20, 200), (24, 10)):
    n = 1 << k
    a = torch.from_numpy(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
    b, c = torch.empty_like(a), torch.empty_like(a)

    def step():
        ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0)
        ctx.fr_ntt_dev(b.data_ptr(), n, c.data_ptr(), k, pa.NTT_INVERSE)
    for _ in range(5):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    assert torch.equal(a, c)
    print(f"{tag}: NTT 2^{k} fwd+inv {dt * 1e6:9.1f} us  {n * k / dt:.3e} butterflies/s", flush=True)
    del a, b, c
# MSM 2^20 over a GPU-generated SRS
n = 1 << 20
ck = pa.CommitKey.setup(n - 1, pa.field.fr_to_limbs(0x123456789ABCDEF), ctx, precompute=True)
sc = torch.from_numpy(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
r0 = ck._bases.msm_dev(sc.data_ptr(), n)
ctx.profile(True)
t0 = time.perf_counter()
for _ in range(5):
    r = ck._bases.msm_dev(sc.data_ptr(), n)
dt = (time.perf_counter() - t0) / 5
prof = ctx.profile_read()
ctx.profile(False)
print(f"{tag}: MSM 2^20 {dt * 1e3:.3f} ms  " + " ".join(f"{k.replace('msm_', '')}={v[1] / v[0] * 1e3:.0f}us" for k, v in prof.items()),
      "checksum", hex(int(r[0])), flush=True)
