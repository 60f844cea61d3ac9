"""GPU timing (not a test): small NTTs, kernel times per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plonk_prototype_amd as pa
ctx = pa.Context(0); st = torch.cuda.current_stream().cuda_stream
for k in (8, 10, 11, 12, 13, 14):
    n = 1 << k
    a = torch.randint(0, 2**31, (n * 4,), dtype=torch.int64, device="cuda") & ((1 << 60) - 1); b = torch.empty_like(a)
    for _ in range(3): ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, stream=st)
    ctx.sync(); torch.cuda.synchronize(); ctx.profile(True)
    for _ in range(20): ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, 0, stream=st)
    ctx.sync(); pr = ctx.profile_read(); ctx.profile(False)
    print(k, pa.ntt_plan(k), {s: round(v[1] / v[0] * 1e3, 1) for s, v in pr.items()}, flush=True)
