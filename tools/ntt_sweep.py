"""GPU sweep (not a test): per-kernel NTT times vs size, batch and tile; MSM kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
ctx = pa.Context(0)
st = torch.cuda.current_stream().cuda_stream
def run(k, batch, flags, tile, reps=10, maxr=10, radix=8):
    n = 1 << k
    ctx.set_option("ntt_tile_log", tile)
    ctx.set_option("ntt_max_radix", maxr)
    ctx.set_option("ntt_radix", radix)
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
    print(f"k={k} batch={batch} flags={flags} tile={tile} maxr={maxr} radix={radix}: {dt*1e6:8.1f} us/call  {bf:.3e} butterflies/s  kernels_us={ks}", flush=True)
import sys
from oracle.cpu_oracle import CpuOracle
o = CpuOracle()
# correctness of the radix-4 kernels first
for k in (3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 20):
    a = o.fr_sample(k, 1 << k)
    ctx.set_option("ntt_radix", 4)
    for maxr in (10, 7):
        ctx.set_option("ntt_max_radix", maxr)
        for flags in (0, 1, 2, 3):
            ok = np.array_equal(ctx.fr_ntt(a, k, flags), o.fr_ntt(a, k, flags, 8))
            if not ok: print("RADIX4 MISMATCH", k, maxr, flags, flush=True)
print("radix-4 check done", flush=True)
for k in (16, 20, 22, 24):
    for xcd in (0, 1, 0, 1):
        ctx.set_option("ntt_xcd", xcd)
        print("xcd", xcd, end=" ")
        run(k, 1, 0, 0, maxr=10, radix=4)
