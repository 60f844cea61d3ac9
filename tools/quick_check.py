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

# ---------------------------------------------------------------- Fp + MSM
af = o.fp_to_mont(ints_to_limbs([v % B.P_MOD for v in B.sample_fr(5, 500)], 6))
bf = o.fp_to_mont(ints_to_limbs([(v * 0x1234567 + 99) % B.P_MOD for v in B.sample_fr(6, 500)], 6))
print("fp mul", np.array_equal(ctx.field_op(3, af, bf), o.fp_mul(af, bf)))
afi = limbs_to_ints(af); bfi = limbs_to_ints(bf)
print("fp add", limbs_to_ints(ctx.field_op(4, af, bf)) == [(x + y) % B.P_MOD for x, y in zip(afi, bfi)])
print("fp sub", limbs_to_ints(ctx.field_op(5, af, bf)) == [(x - y) % B.P_MOD for x, y in zip(afi, bfi)])
k0 = ints_to_limbs([0x1234567], 4)[0]; dd = ints_to_limbs([0xabcdef123456789abcdef], 4)[0]
for n in [0, 1, 2, 31, 32, 33, 100, 1000, 5000, 1 << 14, 1 << 16]:
    pts = o.g1_bases_arith(k0, dd, max(n, 1), 8)[:n]
    sc = o.fr_sample(900 + n, n)
    if n > 8:
        sc[0] = 0; sc[1] = o.fr_to_mont(ints_to_limbs([1], 4))[0]; sc[2] = o.fr_to_mont(ints_to_limbs([B.R_MOD - 1], 4))[0]
        pts[3] = pts[4]; sc[4] = sc[3]          # duplicate base, same scalar
        pts[5, 6:] = o.fp_to_mont(ints_to_limbs([(B.P_MOD - v) % B.P_MOD for v in limbs_to_ints(o.fp_from_mont(pts[6, 6:].reshape(1, 6)))], 6))[0]
        pts[5, :6] = pts[6, :6]; sc[5] = sc[6]  # P and -P with the same scalar
        pts[7] = 0                               # infinity among the bases
    t = time.time(); exp = o.g1_msm(pts, sc, 0, 8); tc = time.time() - t
    t = time.time(); got = pa.msm_variable_base(pts, sc, ctx); tg = time.time() - t
    gaff, ident = pa.g1_to_affine(got)
    print(f"msm n={n} ok={np.array_equal(gaff, exp)} ident={ident} cpu={tc:.3f}s gpu(e2e)={tg:.3f}s")
# skewed: all scalars equal, and 0/1-heavy
n = 1 << 14
pts = o.g1_bases_arith(k0, dd, n, 8)
sc = np.repeat(o.fr_sample(7, 1), n, axis=0)
print("msm all-equal ok=", np.array_equal(pa.g1_to_affine(pa.msm_variable_base(pts, sc, ctx))[0], o.g1_msm(pts, sc, 0, 8)))
sc = o.fr_sample(8, n); one = o.fr_to_mont(ints_to_limbs([1], 4))[0]
sc[::2] = one; sc[1::4] = 0
print("msm 0/1-heavy ok=", np.array_equal(pa.g1_to_affine(pa.msm_variable_base(pts, sc, ctx))[0], o.g1_msm(pts, sc, 0, 8)))
