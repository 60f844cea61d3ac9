"""GPU sweep (not a test): per-kernel NTT times vs size, batch and tile; MSM kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
ctx = pa.Context(0)
st = torch.cuda.current_stream().cuda_stream
def run(k, batch, flags, tile, reps=10, maxr=10):
    n = 1 << k
    ctx.set_option("ntt_tile_log", tile)
    ctx.set_option("ntt_max_radix", maxr)
    a = torch.randint(0, 2**31, (batch * n * 4,), dtype=torch.int64, device="cuda")  # not canonical; timing only
    a &= (1 << 60) - 1
    b = torch.empty_like(a)
    for _ in range(2):
        ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, batch=batch, in_stride=n, out_stride=n, stream=st)
    ctx.sync(); torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, batch=batch, in_stride=n, out_stride=n, stream=st)
    ctx.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = ctx.profile_read(); ctx.profile(False)
    ks = {s: round(v[1] / v[0] * 1e3, 1) for s, v in prof.items()}
    bf = batch * (n // 2) * k / dt
    print(f"k={k} batch={batch} flags={flags} tile={tile} maxr={maxr}: {dt*1e6:8.1f} us/call  {bf:.3e} butterflies/s  kernels_us={ks}", flush=True)
for k in (16, 18, 20, 21, 22, 24):
    for maxr in (10, 9, 8, 7, 6):
        run(k, 1, 0, 11, maxr=maxr)
run(20, 1, 0, 12, maxr=10); run(20, 4, 0, 11, maxr=7); run(20, 4, 0, 12, maxr=10)
