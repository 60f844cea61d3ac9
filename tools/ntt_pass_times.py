"""GPU probe (not a test): per-pass kernel times of one forward / coset transform at the given sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plonk_prototype_amd as pa
ctx = pa.Context(0)
for k in [int(x) for x in sys.argv[1:]] or [22, 24, 26, 28]:
    n = 1 << k
    a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    for flags, in_len, tag in ((0, n, "fwd"), (pa.NTT_COSET, n // 4, "coset n/4 -> n"), (pa.NTT_INVERSE | pa.NTT_COSET, n, "coset inv")):
        ctx.fr_ntt_dev(a.data_ptr(), in_len, b.data_ptr(), k, flags)
        ctx.sync(); ctx.profile(True)
        t0 = time.perf_counter()
        for _ in range(3): ctx.fr_ntt_dev(a.data_ptr(), in_len, b.data_ptr(), k, flags)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3
        prof = ctx.profile_read(); ctx.profile(False)
        print(f"2^{k} {tag:16s} {dt*1e3:8.3f} ms  " + "  ".join(f"{s}: {v[1]/v[0]:.3f} ms x{v[0]//3}" for s, v in prof.items()), flush=True)
    del a, b
    torch.cuda.empty_cache(); ctx.trim()
