"""Ad-hoc GPU sanity run (not a test): field ops + NTT vs the CPU oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_prototype_amd as pa
from oracle.cpu_oracle import CpuOracle, INVERSE, COSET
o = CpuOracle()
ctx = pa.Context(0)
a = o.fr_sample(1, 1000); b = o.fr_sample(2, 1000)
print("fr mul", np.array_equal(ctx.field_op(0, a, b), o.fr_mul(a, b)))
# add/sub via oracle: use numpy python ints
from oracle import bigint_oracle as B
from oracle.cpu_oracle import limbs_to_ints, ints_to_limbs
ai = limbs_to_ints(a); bi = limbs_to_ints(b)
print("fr add", limbs_to_ints(ctx.field_op(1, a, b)) == [(x+y)%B.R_MOD for x,y in zip(ai,bi)])
print("fr sub", limbs_to_ints(ctx.field_op(2, a, b)) == [(x-y)%B.R_MOD for x,y in zip(ai,bi)])
ks = [int(x) for x in sys.argv[1:]] or [0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,16,18,20]
for k in ks:
    n = 1 << k
    x = o.fr_sample(100 + k, n)
    for flags in (0, INVERSE, COSET, INVERSE | COSET):
        t = time.time(); exp = o.fr_ntt(x, k, flags, 8); tc = time.time() - t
        t = time.time(); got = ctx.fr_ntt(x, k, flags); tg = time.time() - t
        ok = np.array_equal(got, exp)
        print(f"k={k} flags={flags} ok={ok} cpu={tc:.3f}s gpu(e2e)={tg:.3f}s")
        if not ok:
            bad = np.nonzero((got != exp).any(axis=1))[0]
            print("  mismatches", len(bad), bad[:8])
    # short input, zero padded
    if k >= 2:
        got = ctx.fr_ntt(x[: n // 4 + 1], k, COSET); exp = o.fr_ntt(x[: n // 4 + 1], k, COSET, 8)
        print(f"k={k} padded ok={np.array_equal(got, exp)}")
