"""pm_fr_poly_ruffini_dev by size: us per call and the fraction of HBM of the algorithmic 64 B per element; the result is
compared with the C oracle up to 2^16 and, above, through q(X) (X - z) + p(z) = p(X) at a random point.
(r06 A/B of the scaled-prefix-sum kernels against the r03 - r05 product scans, removed since: profiles/r06_ruffini_ab.txt.)
usage: python tools/ruffini_ab.py [LOG_N ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from oracle.bigint_oracle import R_MOD  # noqa: E402
from oracle.cpu_oracle import CpuOracle  # noqa: E402  (input synthesis + the check)

orc = CpuOracle()
ctx = pa.Context(0)
lib, h = ctx._lib, ctx._h
fi, fl = pa.field.fr_from_limbs, pa.field.fr_to_limbs
sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 18, 20, 21, 22, 24]
pt = orc.fr_sample(13, 1)[0]
x = orc.fr_sample(14, 1)[0]
pp = pt.ctypes.data_as(C.POINTER(C.c_uint64))
for k in sizes:
    for n in ((1 << k), (1 << k) - 37):
        host = orc.fr_sample(5 + k, n)
        va = pa.DeviceVector.from_host(ctx, host)
        vo = pa.DeviceVector(ctx, n)
        ctx._check(lib.pm_fr_poly_ruffini_dev(h, va._p, n, pp, vo._p, None))
        ctx.sync()
        if k <= 16:
            ok = np.array_equal(vo.to_host()[: n - 1], orc.fr_poly_ruffini(host, pt)[: n - 1])
        else:
            px, pz, qx = (fi(ctx.fr_evaluate(va.ptr, n, x)), fi(ctx.fr_evaluate(va.ptr, n, pt)), fi(ctx.fr_evaluate(vo.ptr, n - 1, x)))
            ok = (qx * (fi(x) - fi(pt)) + pz - px) % R_MOD == 0
        ctx.profile(True)
        for _ in range(5):
            lib.pm_fr_poly_ruffini_dev(h, va._p, n, pp, vo._p, None)
        ctx.sync()
        (name, (cnt, ms)), = ctx.profile_read().items()
        ctx.profile(False)
        us = ms / cnt * 1e3
        print(f"n=2^{k}{'' if n == 1 << k else '-37'}: {us:8.1f} us ({64 * n / us / 1e3 / 8000:.3f} of HBM)   ok={ok}", flush=True)
        va.free()
        vo.free()
