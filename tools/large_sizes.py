"""One-off checks at sizes beyond the test suite: 2^26-point NTT (the 4n domain of a 2^24-gate circuit):
round trips, and spot values against Horner evaluation (an independent kernel)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from plonk_prototype_amd.field import GENERATOR, R_MOD, fr_from_limbs, fr_to_limbs  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << k
ctx = pa.Context(0)
rng = np.random.default_rng(26)
a = torch.from_numpy((rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)).view(np.int64)).cuda()
b, c = torch.empty_like(a), torch.empty_like(a)
omega = fr_from_limbs(pa.domain_info(k)[0])
for flags, name in ((0, "fft"), (pa.NTT_COSET, "coset_fft")):
    ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.fr_ntt_dev(a.data_ptr(), n, b.data_ptr(), k, flags)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.fr_ntt_dev(b.data_ptr(), n, c.data_ptr(), k, flags | pa.NTT_INVERSE)
    ctx.sync()
    ok_rt = torch.equal(a, c)
    ok_pts = True
    for j in (0, 1, 12345677, n - 1):
        x = pow(omega, j, R_MOD) * (GENERATOR if flags else 1) % R_MOD
        want = ctx.fr_evaluate(a.data_ptr(), n, fr_to_limbs(x))
        got = b[j].cpu().numpy().view(np.uint64)
        ok_pts = ok_pts and bool(np.array_equal(want, got))
    print(f"2^{k} {name}: {dt * 1e3:.2f} ms ({(n // 2) * k / dt:.3e} butterflies/s)  round trip {ok_rt}  "
          f"Horner spot checks {ok_pts}", flush=True)
    assert ok_rt and ok_pts
