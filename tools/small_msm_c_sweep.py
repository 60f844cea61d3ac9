"""GPU probe (not a test): window width of the resident table for SMALL commit keys.
usage: python tools/small_msm_c_sweep.py [LOGS] [CS] [BATCHES]     e.g.  10,12,14,16,17  11,13,14,16  1,4
Every (n, batch, c): result compared with the first width's, best of 3 x 5 calls, per-kernel event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs
logs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10, 11, 12, 13, 14, 15, 16, 17]
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 11, 13, 14, 16]
batches = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4]
o = CpuOracle()
ctx = pa.Context(0)
for k in logs:
    n = 1 << k
    pts = o.g1_bases_arith(ints_to_limbs([0x1234567], 4)[0], ints_to_limbs([0xabcdef123456789abcdef], 4)[0], n, 16)
    for kb in batches:
        scs = np.concatenate([o.fr_sample(100 + j, n) for j in range(kb)])
        d = torch.from_numpy(np.ascontiguousarray(scs).view(np.int64)).cuda()
        ref = None
        for c in cs:                       # 0 = no table (msm_variable_base's path, the library's own width)
            bases = pa.host.Bases(ctx, pts)
            if c:
                bases.precompute(c)
            r = bases.msm_batch_dev(d.data_ptr(), n, kb)
            if ref is None: ref = r
            assert np.array_equal(pa.g1_to_affine(r[0])[0], pa.g1_to_affine(ref[0])[0])
            best = 1e9
            for rep in range(3):
                ctx.sync(); ctx.profile(True)
                t0 = time.perf_counter()
                for _ in range(5): bases.msm_batch_dev(d.data_ptr(), n, kb)
                dt = (time.perf_counter() - t0) / 5
                prof = ctx.profile_read(); ctx.profile(False)
                if dt < best:
                    best, ks = dt, {s.replace("msm_", ""): round(v[1] / 5 * 1e3) for s, v in prof.items()}
            # the timers cost ~5 us per kernel: a second, untimed figure
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(10): bases.msm_batch_dev(d.data_ptr(), n, kb)
            ctx.sync(); untimed = (time.perf_counter() - t0) / 10
            print(f"2^{k} batch {kb} c={c:2d} {best*1e3:7.3f} ms timed {untimed*1e3:7.3f} ms untimed  {ks}", flush=True)
            bases.free()
