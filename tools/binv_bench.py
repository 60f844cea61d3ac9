"""pm_fr_batch_inverse_dev: time per call by size and by quads per thread (option binv_quads; 0 = the library's choice).
usage: python tools/binv_bench.py [LOG_N ...]      -> one row per (size, Q): us, GB/s of the algorithmic 64 B per element"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_prototype_amd as pa  # noqa: E402
from oracle.cpu_oracle import CpuOracle  # noqa: E402  (input synthesis + the check)

orc = CpuOracle()
ctx = pa.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [20, 22, 24]
for k in sizes:
    n = 1 << k
    host = orc.fr_sample(77 + k, n)
    host[5] = 0
    host[n - 3] = 0
    v = pa.DeviceVector.from_host(ctx, host)
    orig = pa.DeviceVector.from_host(ctx, host)
    prod = pa.DeviceVector(ctx, n)
    one = orc.fr_to_mont(np.array([[1, 0, 0, 0]], np.uint64))[0]
    exp = np.tile(one, (n, 1))
    exp[5] = 0
    exp[n - 3] = 0
    for q in (0, 1, 2, 4, 8, 16, 32, 64):
        if 4 * q > n:
            continue
        ctx.set_option("binv_quads", q)
        # check this configuration: a * a^-1 = 1 on the non-zero entries, zeros stay
        ctx._check(ctx._lib.pm_dev_upload(ctx._h, v._p, host.ctypes.data, n * 32))
        ctx.fr_batch_inverse(v.ptr, n)
        ctx.fr_vec_op(2, orig.ptr, v.ptr, n, prod.ptr, n)
        ok = np.array_equal(prod.to_host(), exp)
        ctx.sync()
        ctx.profile(True)
        for _ in range(5):
            ctx.fr_batch_inverse(v.ptr, n)
        ctx.sync()
        (name, (cnt, ms)), = ctx.profile_read().items()
        ctx.profile(False)
        us = ms / cnt * 1e3
        print(f"2^{k} Q={q:3d}  {us:9.1f} us  {64 * n / us / 1e3:8.1f} GB/s  {64 * n / us / 1e3 / 8000:.4f} of HBM   ok={ok}", flush=True)
    ctx.set_option("binv_quads", 0)
    v.free()
    orig.free()
    prod.free()
