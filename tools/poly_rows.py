"""GPU timing of the O(n) polynomial helpers at 2^k elements; run it under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: python tools/poly_rows.py [log_n] [reps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << k
o = CpuOracle()
ctx = pa.Context(0)
if os.environ.get("PM_POLY_LOOKBACK") is not None:            # A/B: 0 = the three-stage scan (totals / scan / replay)
    ctx.set_option("poly_lookback", int(os.environ["PM_POLY_LOOKBACK"]))
va = pa.DeviceVector.from_host(ctx, o.fr_sample(11, n))
vo = pa.DeviceVector(ctx, n)
pt = o.fr_sample(13, 1)[0]
pp = pt.ctypes.data_as(C.POINTER(C.c_uint64))
lib, h = ctx._lib, ctx._h
rows = {"poly_ruffini": lambda: lib.pm_fr_poly_ruffini_dev(h, va._p, n, pp, vo._p, None),
        "prefix_product": lambda: lib.pm_fr_prefix_product_dev(h, va._p, n, vo._p, None),
        "batch_inverse": lambda: lib.pm_fr_batch_inverse_dev(h, vo._p, n, None)}
for name, fn in rows.items():
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:16s} 2^{k}: {dt*1e6:8.1f} us  {64*n/dt/1e9:8.1f} GB/s  {64*n/dt/8e12:.3f} of HBM", flush=True)
