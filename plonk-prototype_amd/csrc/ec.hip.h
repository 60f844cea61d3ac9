// BLS12-381 G1 group law on the device in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2), over the carry-free Fp of fields.hip.h (14 x 28-bit limbs, R' = 2^392).
//
// This is the arithmetic under dusk_bls12_381::multiscalar_mul::msm_variable_base
// (dusk-bls12_381 0.8, pinned at ref:Cargo.toml:20): bucket += point
// (mixed addition), bucket + bucket, doubling.  Formulas: EFD madd-2008-s, add-2008-s,
// dbl-2008-s-1 / mdbl-2008-s-1 for y^2 = x^3 + b (a = 0).
//
// Bounds, written (B, V) = limbs < B*2^28, value < V*p.  fe_mul needs Ba*Bb < 17 and
// Va*Vb < 2520 and returns (1, <2).  A point held in registers or memory obeys the class
//     X (1+, <10)   Y (1+, <5)   ZZ, ZZZ (1, <2)            ("1+" = limbs < 2^28 + 16)
// which every routine below both accepts and produces.  The identity is carried as a flag
// in registers and as ZZ == 0 (all limbs) in memory.
//
// Exceptional cases are detected after the fact: P = U2 - U1 = 0 mod p makes ZZ3 = ..*P^2
// congruent to zero, and a Montgomery product (normalised, < 2p) that is 0 mod p has exactly
// the limbs of 0 or of p.  The slow path (doubling, or the identity for P + (-P)) is then
// taken by the few lanes that need it.
#pragma once
#include "fields.hip.h"

namespace pm {

struct Xyzz {
  Fp x, y, zz, zzz;
  bool inf;
};

PM_DEV bool fp_is_zero_product(const Fp& a) {  // a = fe_mul output: is it 0 mod p?
  constexpr Limbs<14> M = Consts<FpP>::mod_limbs();
  u32 z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    z |= a.l[i];
    e |= a.l[i] ^ M.v[i];
  }
  return z == 0 || e == 0;
}
// exact test for a lazily reduced value (V < 2520): reduce through a product first
PM_DEV bool fp_is_zero_lazy(const Fp& a) { return fp_is_zero_product(fe_mul<FpP>(a, fe_one<FpP>())); }

PM_DEV Xyzz xyzz_identity() {
  Xyzz r;
  r.x = fe_zero<FpP>();
  r.y = fe_zero<FpP>();
  r.zz = fe_zero<FpP>();
  r.zzz = fe_zero<FpP>();
  r.inf = true;
  return r;
}

// 2 * (x, y) for an affine point (mdbl-2008-s-1); x, y in class
PM_DEV Xyzz xyzz_double_affine(const Fp& x, const Fp& y) {
  Xyzz r;
  Fp U = fe_add<FpP>(y, y);                                 // (2+, <10)
  Fp V = fe_sqr<FpP>(U);
  Fp W = fe_mul<FpP>(U, V);
  Fp S = fe_mul<FpP>(x, V);
  Fp XX = fe_sqr<FpP>(x);
  Fp M = fe_add<FpP>(fe_add<FpP>(XX, XX), XX);              // (3, <6)
  Fp MM = fe_sqr<FpP>(M);
  r.x = fe_norm<FpP>(fe_sub<FpP, 5, 1>(MM, fe_add<FpP>(S, S)));   // (1+, <7)
  Fp D = fe_sub<FpP, 8, 1>(S, r.x);                         // (4, <10)
  Fp YA = fe_mul<FpP>(M, D);
  Fp YB = fe_mul<FpP>(W, y);
  r.y = fe_norm<FpP>(fe_sub<FpP, 3, 1>(YA, YB));            // (1+, <5)
  r.zz = V;
  r.zzz = W;
  r.inf = false;
  return r;
}

// 2 * p (dbl-2008-s-1)
PM_DEV Xyzz xyzz_double(const Xyzz& p) {
  if (p.inf) return p;
  Xyzz r = xyzz_double_affine(p.x, p.y);
  r.zz = fe_mul<FpP>(r.zz, p.zz);
  r.zzz = fe_mul<FpP>(r.zzz, p.zzz);
  return r;
}

// acc + (x2, y2) with (x2, y2) affine, not the identity; x2 (<=1, <1), y2 (<=3, <3)
PM_DEV Xyzz xyzz_madd(const Xyzz& acc, const Fp& x2, const Fp& y2) {
  if (acc.inf) {
    Xyzz r;
    r.x = x2;
    r.y = fe_norm<FpP>(y2);
    r.zz = fe_one<FpP>();
    r.zzz = fe_one<FpP>();
    r.inf = false;
    return r;
  }
  Xyzz r;
  // five rounds of two independent products each (fe_mul2 / fe_sqr2: two dependent-mad chains per wave)
  Fp U2, S2;
  fe_mul2<FpP>(x2, acc.zz, y2, acc.zzz, U2, S2);
  Fp P = fe_norm<FpP>(fe_sub<FpP, 11, 1>(U2, acc.x));      // (1+, <13)
  Fp R = fe_norm<FpP>(fe_sub<FpP, 6, 1>(S2, acc.y));       // (1+, <8)
  Fp PP, RR;
  fe_sqr2<FpP>(P, R, PP, RR);
  // PPP = P PP, Q = X1 PP and ZZ3 = ZZ1 PP: three independent products, three chains
  Fp PPP, Q;
  fe_mul3<FpP>(P, PP, acc.x, PP, acc.zz, PP, PPP, Q, r.zz);
  Fp t = fe_sub<FpP, 3, 1>(RR, PPP);                       // (4, <5)
  r.x = fe_norm<FpP>(fe_sub<FpP, 5, 1>(t, fe_add<FpP>(Q, Q)));   // (1+, <10)
  Fp D = fe_sub<FpP, 11, 1>(Q, r.x);                       // (4, <13)
  // Y3 = R D - Y1 PPP as R D + (6 p - Y1) PPP with ONE reduction (fe_mma2: 1 x 4 + 3 x 1 = 7 < 17 in the column
  // bound, value < (8 x 13 + 6 x 2) p^2 / R' + p < 1.05 p), in lock step with ZZZ3 = ZZZ1 PPP
  const Fp nY1 = fe_sub<FpP, 6, 1>(fe_zero<FpP>(), acc.y);  // (3, <6)
  fe_mma2<FpP>(R, D, nY1, PPP, acc.zzz, PPP, r.y, r.zzz);  // Y3 (1, <2)
  r.inf = false;
  if (fp_is_zero_product(r.zz)) {  // same x: acc == +-(x2, y2)
    if (fp_is_zero_lazy(R))
      r = xyzz_double_affine(x2, fe_norm<FpP>(y2));
    else
      r = xyzz_identity();
  }
  return r;
}

// a + b, both in class (add-2008-s)
PM_DEV Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  Xyzz r;
  Fp U1, U2, S1, S2;
  fe_mul2<FpP>(a.x, b.zz, b.x, a.zz, U1, U2);
  fe_mul2<FpP>(a.y, b.zzz, b.y, a.zzz, S1, S2);
  Fp P = fe_norm<FpP>(fe_sub<FpP, 3, 1>(U2, U1));          // (1+, <5)
  Fp R = fe_norm<FpP>(fe_sub<FpP, 3, 1>(S2, S1));          // (1+, <5)
  Fp PP, RR;
  fe_sqr2<FpP>(P, R, PP, RR);
  Fp PPP, Q, z12, zzz12;
  fe_mul2<FpP>(P, PP, U1, PP, PPP, Q);
  fe_mul2<FpP>(a.zz, b.zz, a.zzz, b.zzz, z12, zzz12);
  Fp t = fe_sub<FpP, 3, 1>(RR, PPP);
  r.x = fe_norm<FpP>(fe_sub<FpP, 5, 1>(t, fe_add<FpP>(Q, Q)));   // (1+, <10)
  Fp D = fe_sub<FpP, 11, 1>(Q, r.x);
  // Y3 = R D - S1 PPP with one reduction (S1 is a product: (1, <2), so 3 p - S1 has limbs < 3 x 2^28), next to ZZ3
  const Fp nS1 = fe_sub<FpP, 3, 1>(fe_zero<FpP>(), S1);     // (3, <3)
  fe_mma2<FpP>(R, D, nS1, PPP, z12, PP, r.y, r.zz);         // Y3 (1, <2)
  r.zzz = fe_mul<FpP>(zzz12, PPP);
  r.inf = false;
  if (fp_is_zero_product(r.zz)) {
    if (fp_is_zero_lazy(R))
      r = xyzz_double(a);
    else
      r = xyzz_identity();
  }
  return r;
}

// k * p for a small non-negative integer k (left-to-right double-and-add)
PM_DEV Xyzz xyzz_mul_small(const Xyzz& p, u32 k) {
  Xyzz r = xyzz_identity();
  if (k == 0 || p.inf) return r;
  for (int bit = 31 - __clz(k); bit >= 0; --bit) {
    r = xyzz_double(r);
    if ((k >> bit) & 1) r = xyzz_add(r, p);
  }
  return r;
}

// ---- one point on TWO neighbouring lanes --------------------------------------------------------
// The reduction kernels after the big accumulate run few waves with long chains of dependent group
// operations, and a lone wave issues one VALU instruction per ~7 cycles.  There a point lives on a lane pair:
// the even lane ("A") holds c0 = X, c1 = ZZ, the odd lane ("B") holds c0 = Y, c1 = ZZZ.  Both lanes run the
// SAME instruction stream; an addition is 7 rounds of one field product per lane instead of 14 on one lane
// (doubling: 5 instead of 9), with four (two) exchanges of one field element between the lanes of the pair
// (DPP quad_perm, no LDS).  Same formulas, same bound classes and the same after-the-fact handling of
// P + P / P - P as xyzz_add / xyzz_double above; the `inf` flag is kept equal on both lanes.
struct Half {
  Fp c0, c1;
  u32 inf;   // 0 / 1.  A whole word: with `bool` the struct has three padding bytes, and hipcc copies them through SCRATCH
             // on every assignment (scratch_load / s_waitcnt / scratch_store: two global-memory round trips per addition,
             // exposed when one wave runs per SIMD -- r04, found in the ISA of msm_bucket_reduce_kernel)
};
PM_DEV Fp fp_pair_swap(const Fp& v) {   // lanes 2k <-> 2k + 1
  Fp r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.l[i], 0xB1, 0xF, 0xF, false);
  return r;
}
PM_DEV Fp fp_select(bool c, const Fp& a, const Fp& b) {   // c ? a : b
  Fp r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
PM_DEV Half half_identity() {
  Half r;
  r.c0 = fe_zero<FpP>();
  r.c1 = fe_zero<FpP>();
  r.inf = true;
  return r;
}
// 2 p (dbl-2008-s-1)
PM_DEV Half half_double(const Half& p, bool isB) {
  if (p.inf) return p;
  const Fp t = isB ? fe_add<FpP>(p.c0, p.c0) : p.c0;       // B: U = 2 y (2+, <10);  A: x
  const Fp s1 = fe_sqr<FpP>(t);                             // A: XX;  B: V
  const Fp s1o = fp_pair_swap(s1);                          // A: V;   B: XX
  const Fp s2 = fe_mul<FpP>(t, fp_select(isB, s1, s1o));    // A: S = x V;  B: W = U V
  const Fp xx = fp_select(isB, s1o, s1);
  const Fp M = fe_add<FpP>(fe_add<FpP>(xx, xx), xx);        // (3, <6) on both lanes
  Half r;
  Fp s3;                                                    // A: MM;  B: YB = W y   | two chains per wave:
  fe_mul2<FpP>(fp_select(isB, s2, M), fp_select(isB, p.c0, M), fp_select(isB, s2, s1o), p.c1, s3, r.c1);   // A: ZZ3 = V zz;  B: ZZZ3 = W zzz
  const Fp x3 = fe_norm<FpP>(fe_sub<FpP, 5, 1>(s3, fe_add<FpP>(s2, s2)));      // A: X3 (1+, <7); B: unused
  const Fp D = fe_sub<FpP, 8, 1>(s2, x3);                   // (4, <10)
  const Fp ya = fe_mul<FpP>(M, D);                          // A: YA
  const Fp yao = fp_pair_swap(ya);                          // B: YA
  r.c0 = fp_select(isB, fe_norm<FpP>(fe_sub<FpP, 3, 1>(yao, s3)), x3);
  r.inf = false;
  return r;
}
// a + b (add-2008-s)
PM_DEV Half half_add(const Half& a, const Half& b, bool isB) {
  if (a.inf) return b;
  if (b.inf) return a;
  // products in rounds of two independent ones where the formulas allow it (fe_mul2: two chains per wave)
  Fp m1, m2;                                                // A: U1 = X1 ZZ2, U2 = X2 ZZ1;   B: S1 = Y1 ZZZ2, S2 = Y2 ZZZ1
  fe_mul2<FpP>(a.c0, b.c1, b.c0, a.c1, m1, m2);
  const Fp d = fe_norm<FpP>(fe_sub<FpP, 3, 1>(m2, m1));     // A: P;  B: R   (1+, <5)
  Fp e, z12;                                                // A: PP, ZZ1 ZZ2;  B: RR, ZZZ1 ZZZ2
  fe_mul2<FpP>(d, d, a.c1, b.c1, e, z12);
  const Fp dO = fp_pair_swap(d);                            // A: R;  B: P
  const Fp eO = fp_pair_swap(e);                            // A: RR; B: PP
  const Fp f = fe_mul<FpP>(fp_select(isB, dO, m1), fp_select(isB, eO, e));   // A: Q = U1 PP;  B: PPP = P PP
  const Fp fO = fp_pair_swap(f);                            // A: PPP;  B: Q
  Half r;
  const Fp rr = fp_select(isB, e, eO), ppp = fp_select(isB, f, fO), q = fp_select(isB, fO, f);
  const Fp t = fe_sub<FpP, 3, 1>(rr, ppp);                  // (4, <5)
  const Fp x3 = fe_norm<FpP>(fe_sub<FpP, 5, 1>(t, fe_add<FpP>(q, q)));   // (1+, <10), both lanes
  const Fp D = fe_sub<FpP, 11, 1>(q, x3);                   // (4, <13)
  Fp h;                                                     // A: ZZ3 = ZZ1 ZZ2 PP, YA = R D;  B: ZZZ3 = ZZZ1 ZZZ2 PPP, YB = S1 PPP
  fe_mul2<FpP>(z12, fp_select(isB, f, e), fp_select(isB, m1, dO), fp_select(isB, f, D), r.c1, h);
  const Fp hO = fp_pair_swap(h);                            // B: YA
  r.c0 = fp_select(isB, fe_norm<FpP>(fe_sub<FpP, 3, 1>(hO, h)), x3);
  r.inf = false;
  if (fp_is_zero_product(r.c1)) {     // P = 0 mod p makes ZZ3 and ZZZ3 vanish together: both lanes agree
    if (fp_is_zero_lazy(fp_select(isB, d, dO)))   // R
      r = half_double(a, isB);
    else
      r = half_identity();
  }
  return r;
}
PM_DEV Half half_shfl_down(const Half& v, int pairs) {
  Half r;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    r.c0.l[i] = __shfl_down(v.c0.l[i], 2 * pairs);
    r.c1.l[i] = __shfl_down(v.c1.l[i], 2 * pairs);
  }
  r.inf = __shfl_down((int)v.inf, 2 * pairs) != 0;
  return r;
}

// ---- memory format: 4 coordinates x 16 words (14 limbs + 2 pad) = 256 bytes ---------------
PM_DEV Fp ld_fp_limbs(const u32x4* p) {
  u32x4 a = p[0], b = p[1], c = p[2], d = p[3];
  Fp r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
  r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
PM_DEV void st_fp_limbs(u32x4* p, const Fp& v) {
  p[0] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
  p[1] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
  p[2] = u32x4{v.l[8], v.l[9], v.l[10], v.l[11]};
  p[3] = u32x4{v.l[12], v.l[13], 0u, 0u};
}
PM_DEV Xyzz ld_xyzz(const u32x4* base, size_t idx) {
  const u32x4* p = base + 16 * idx;
  Xyzz r;
  r.zz = ld_fp_limbs(p + 8);
  u32 z = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) z |= r.zz.l[i];
  r.inf = (z == 0);
  r.x = ld_fp_limbs(p);
  r.y = ld_fp_limbs(p + 4);
  r.zzz = ld_fp_limbs(p + 12);
  return r;
}
PM_DEV void st_xyzz(u32x4* base, size_t idx, const Xyzz& v) {
  u32x4* p = base + 16 * idx;
  if (v.inf) {
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = z;
    return;
  }
  st_fp_limbs(p, v.x);
  st_fp_limbs(p + 4, v.y);
  st_fp_limbs(p + 8, v.zz);
  st_fp_limbs(p + 12, v.zzz);
}

// one lane's half of a point in the 256-byte record: A reads / writes X and ZZ, B reads / writes Y and ZZZ
PM_DEV Half ld_half(const u32x4* base, size_t idx, bool isB) {
  const u32x4* p = base + 16 * idx;
  const Fp zz = ld_fp_limbs(p + 8);
  u32 z = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) z |= zz.l[i];
  Half r;
  r.inf = (z == 0);
  r.c0 = ld_fp_limbs(p + (isB ? 4 : 0));
  r.c1 = isB ? ld_fp_limbs(p + 12) : zz;
  return r;
}
PM_DEV void st_half(u32x4* base, size_t idx, const Half& v, bool isB) {
  u32x4* p = base + 16 * idx;
  if (v.inf) {
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[(isB ? 4 : 0) + i] = z;
      p[(isB ? 12 : 8) + i] = z;
    }
    return;
  }
  st_fp_limbs(p + (isB ? 4 : 0), v.c0);
  st_fp_limbs(p + (isB ? 12 : 8), v.c1);
}

}  // namespace pm
