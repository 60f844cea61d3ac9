"""GPU A/B (not a test): tile width of the 2^20 / 2^22 NTT passes -- 4 columns per workgroup (2^12-element tile, one
workgroup of 1024 threads per CU) against 2 and 1 columns (2^11 / 2^10 elements: 2 / 4 independent workgroups per CU
whose barrier phases can overlap).  Results compared bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle
ctx = pa.Context(0)
o = CpuOracle()
st = torch.cuda.current_stream().cuda_stream
for k in (20, 18, 19):
    n = 1 << k
    host = o.fr_sample(k, n)
    a = torch.from_numpy(host.view(np.int64)).cuda()
    b, ref = torch.empty_like(a), None
    for batch_note, reps in (("", 30),):
        for tile in (0, 12, 11, 10, 0, 12, 11, 10):
            try:
                ctx.set_option("ntt_tile_log", tile)
                for flags in (0,):
                    for _ in range(2):
                        ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, stream=st)
                    ctx.sync(); torch.cuda.synchronize()
                    if ref is None:
                        ref = b.clone()
                    assert torch.equal(b, ref), (k, tile)
                    ctx.profile(True)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, stream=st)
                    ctx.sync(); torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                    prof = ctx.profile_read(); ctx.profile(False)
                    ks = {s.replace("ntt_pass_", ""): round(v[1] / v[0] * 1e3, 1) for s, v in prof.items()}
                    print(f"2^{k} tile_log={tile:2d} plan={pa.ntt_plan(k)} {dt*1e6:8.1f} us  {(n // 2) * k / dt:.3e} butterflies/s  kernels_us={ks}", flush=True)
            except Exception as e:
                print(f"2^{k} tile_log={tile}: {e}", flush=True)
ctx.set_option("ntt_tile_log", 0)
