"""One-off GPU probe (not a test): MSM results under the extreme values the tunables accept, against the default result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, ints_to_limbs

o = CpuOracle()
ctx = pa.Context(0)
bad = 0
for n in (1, 50, 5000, 70000):
    pts = o.g1_bases_arith(ints_to_limbs([0x1234567 + n], 4)[0], ints_to_limbs([0x9E3779B9], 4)[0], n, 8)
    sc = o.fr_sample(n, n)
    for table in (0, 13, 16):
        bases = pa.host.Bases(ctx, pts)
        if table:
            bases.precompute(table)
        ref = bases.msm(sc)
        assert np.array_equal(pa.g1_to_affine(ref)[0], o.g1_msm(pts, sc, 0, 8))
        for key, vals in (("msm_chunk", (1, 2, 7, 15, 200, 1000, 4096)), ("msm_lb", (1, 16, 64, 256, 1024)),
                          ("msm_window_bits", (4, 5, 11, 17, 20))):
            for v in vals:
                if key == "msm_window_bits" and table:
                    continue
                ctx.set_option(key, v)
                try:
                    got = bases.msm(sc)
                    ok = np.array_equal(pa.g1_to_affine(got)[0], pa.g1_to_affine(ref)[0])
                except pa.Error as e:
                    ok = None
                    print(f"n={n} table={table} {key}={v}: refused: {e}", flush=True)
                finally:
                    ctx.set_option(key, 0)
                if ok is False:
                    bad += 1
                    print(f"n={n} table={table} {key}={v}: WRONG", flush=True)
        bases.free()
    print(f"n={n} done", flush=True)
print("mismatches:", bad)
