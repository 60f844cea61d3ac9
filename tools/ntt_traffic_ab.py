"""GPU A/B (not a test): inter-pass twiddles from the per-pass N x 36 B tables ("direct", the default)
against two-level tables (2 x 2^(k/2) entries, L2-resident) plus one extra Fr product per element.
VERDICT r01 item 8.  Prints per-kernel times; results are compared bit for bit first."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle
ctx = pa.Context(0)
o = CpuOracle()
st = torch.cuda.current_stream().cuda_stream
for k in (20, 22, 24, 26):
    n = 1 << k
    host = o.fr_sample(k, n)
    a = torch.from_numpy(host.view(np.int64)).cuda()
    b, ref = torch.empty_like(a), None
    for flags in ((0, pa.NTT_INVERSE) if k <= 24 else (0,)):
        for direct in (1, 0, 1, 0):
            ctx.set_option("ntt_direct_tw", direct)
            for _ in range(2):
                ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, stream=st)
            ctx.sync(); torch.cuda.synchronize()
            if direct == 1 and ref is None or flags and direct == 1:
                ref = b.clone()
            else:
                assert torch.equal(b, ref), (k, flags, direct)
            reps = 20 if k <= 22 else 6
            ctx.profile(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags, stream=st)
            ctx.sync(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            prof = ctx.profile_read(); ctx.profile(False)
            ks = {s.replace("ntt_pass_", ""): round(v[1] / v[0] * 1e3, 1) for s, v in prof.items()}
            print(f"2^{k} flags={flags} {'direct   ' if direct else 'two-level'} {dt*1e6:9.1f} us  "
                  f"{(n // 2) * k / dt:.3e} butterflies/s  kernels_us={ks}", flush=True)
    del a, b, ref
ctx.set_option("ntt_direct_tw", 1)
