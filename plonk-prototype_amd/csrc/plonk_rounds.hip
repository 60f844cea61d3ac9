// The pointwise work of the PLONK prover rounds between the NTT and MSM calls, one fused kernel per
// round (BASELINE.json configs[3] "Full PLONK prove"):
//   pm_fr_powers_dev         domain.elements(), coset points, powers of a challenge
//   pm_fr_lincomb_dev        linearisation polynomial / aggregated opening polynomial
//   pm_plonk_perm_terms_dev  numerator and denominator of the permutation grand product
//   pm_plonk_quotient_dev    quotient numerator / Z_H on the 4n coset
// Restates dusk_plonk::proof_system::{permutation, quotient_poly, linearisation_poly} of
// dusk-plonk 0.8.2 (ref:Cargo.toml:19; not in the reference tree): the arithmetic identity (times
// q_arith), the permutation identities, and the four widgets behind the gates the reference's gadgets
// emit -- range (ref:src/zk/gadgets.rs:88-91), logic / boolean (:211), fixed-base scalar
// multiplication (:34,37; ref:src/zk/circuits.rs:64) and variable-base curve addition (:40).  The
// formulas are restated from the published dusk-plonk 0.8 design (parity unpinned).
//
// Scaling bookkeeping (poly_common.hip.h): memory holds ABI form (x 2^256); a product of forms 2^a and
// 2^b is of form 2^(a+b-261).  Wires are moved to device form (2^261) once per point, challenges are
// prepared by the host in the form that makes every sum homogeneous, the result leaves in ABI form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "context.h"
#include "host_field.h"
#include "ntt_kernels.hip.h"
#include "poly_common.hip.h"

namespace pm {

// ABI (canonical) -> device form, (1, <1.01)
PM_DEV Fr to_dev(const Fr& x) { return fe_reduce_weak<FrP>(fr_shl5(x)); }

struct RoundConsts {
  u32 beta_k[4][9];  // beta k_j 2^266  (k_0 = 1):  x ABI  -> device form
  u32 beta[9];       // beta 2^266:                 sigma ABI -> device form
  u32 gamma[9];      // gamma 2^261
  u32 alpha[9];      // alpha 2^261
  u32 alpha2[9];     // alpha^2 2^266:              (ABI x ABI = 2^251) -> ABI
  u32 one_abi[9];    // 2^256
  u32 zh_inv[4][9];  // 1 / Z_H(x_i) by i mod 4, 2^261
};

// w + beta k x + gamma, all in device form: value < 1.01 r + 2 r + r, limbs < 3 * 2^29
PM_DEV Fr perm_factor(const Fr& w_dev, const Fr& x_abi, const u32* bk, const Fr& gamma) {
  return fe_add<FrP>(fe_add<FrP>(w_dev, fe_mul<FrP>(x_abi, fr_limbs(bk))), gamma);
}
// product of four factors (value < 4.1 r, limbs < 3 * 2^29 each) -> device form, (1, <1.1)
PM_DEV Fr prod4(const Fr& f0, const Fr& f1, const Fr& f2, const Fr& f3) {
  Fr p = fe_mul<FrP>(f0, fe_norm<FrP>(f1));   // 4.1^2 / 70 + 1 < 1.3
  p = fe_mul<FrP>(f2, p);
  return fe_mul<FrP>(f3, p);
}

// ------------------------------------------------------------------ powers
__global__ void __launch_bounds__(256) powers_kernel(u32x4* out, size_t n, const NttConsts c /* w8[0] = base,
                                                     w8[1] = base^T, scale, one = 2^256 */) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fr cur = fr_pow(fr_limbs(c.w8[0]), t, fr_limbs(c.scale));
  const Fr step = fr_limbs(c.w8[1]), one_abi = fr_limbs(c.one);
  for (size_t i = t; i < n; i += T) {
    st_canon(out, i, fe_mul<FrP>(cur, one_abi));
    cur = fe_mul<FrP>(cur, step);
  }
}

// ------------------------------------------------------------------ linear combination
struct LincombArgs {
  const u32x4* v[PM_LINCOMB_MAX];
  u32 c[PM_LINCOMB_MAX][9];   // device form
  u32 k;
};
__global__ void __launch_bounds__(256) lincomb_kernel(const LincombArgs a, u32x4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    Fr acc = fe_mul<FrP>(ld_canon(a.v[0], i), fr_limbs(a.c[0]));
    for (u32 j = 1; j < a.k; ++j)
      acc = fe_reduce_weak<FrP>(fe_add<FrP>(acc, fe_mul<FrP>(ld_canon(a.v[j], i), fr_limbs(a.c[j]))));
    st_canon(out, i, acc);
  }
}

// ------------------------------------------------------------------ permutation terms
struct PermPtrs {
  const u32x4* w[4];
  const u32x4* s[4];
  const u32x4* roots;
  u32x4* num;
  u32x4* den;
};
__global__ void __launch_bounds__(256) perm_terms_kernel(const PermPtrs p, const RoundConsts kc, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const Fr gamma = fr_limbs(kc.gamma), one_abi = fr_limbs(kc.one_abi);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const Fr x = ld_canon(p.roots, i);
    Fr w[4], f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = to_dev(ld_canon(p.w[j], i));
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = perm_factor(w[j], x, kc.beta_k[j], gamma);
    st_canon(p.num, i, fe_mul<FrP>(prod4(f[0], f[1], f[2], f[3]), one_abi));
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = perm_factor(w[j], ld_canon(p.s[j], i), kc.beta, gamma);
    st_canon(p.den, i, fe_mul<FrP>(prod4(f[0], f[1], f[2], f[3]), one_abi));
  }
}

// ------------------------------------------------------------------ quotient
struct QuotPtrs {
  const u32x4* w[4];
  const u32x4* z;
  const u32x4 *q_m, *q_l, *q_r, *q_o, *q_4, *q_c, *pi;
  const u32x4* q_arith;                      // nullptr = the constant 1
  const u32x4 *q_range, *q_logic, *q_fixed, *q_var;   // nullptr = identically zero: the widget is skipped
  const u32x4* s[4];
  const u32x4* l1;
  const u32x4* x;
  u32x4* out;
};
// Everything the widgets multiply by, in device form (x 2^261)
struct WidgetConsts {
  u32 c1[9], c2[9], c3[9], c4[9], c9[9], c18[9], c81[9], c83[9], edwards_d[9];
  u32 range_sep[9], range_k[3][9];   // separation challenge s and kappa = s^2, kappa^2, kappa^3
  u32 logic_sep[9], logic_k[4][9];
  u32 fixed_sep[9], fixed_k[3][9];
  u32 var_sep[9], var_k[2][9];
};

// Widget arithmetic: every value in device form with normalised limbs and value < 1.05 r, closed
// under these three (fe_mul: < x y / 2^261 + r; the weak reduction after + and - brings the value
// back below r + r / 2^16).  Slower than the hand-scheduled lazy sums of the arithmetic identity, but
// the widget rows are a small part of real circuits and the kernel stays VALU-bound either way.
PM_DEV Fr wadd(const Fr& a, const Fr& b) { return fe_reduce_weak<FrP>(fe_add<FrP>(a, b)); }
PM_DEV Fr wsub(const Fr& a, const Fr& b) { return fe_reduce_weak<FrP>(fe_sub<FrP, 2, 1>(a, b)); }
PM_DEV Fr wmul(const Fr& a, const Fr& b) { return fe_mul<FrP>(a, b); }
// small multiples by additions (a product is ~225 instructions, an addition 9, a weak reduction ~40)
PM_DEV Fr wmul2(const Fr& a) { return fe_reduce_weak<FrP>(fe_add<FrP>(a, a)); }
PM_DEV Fr wmul3(const Fr& a) { return fe_reduce_weak<FrP>(fe_add<FrP>(fe_add<FrP>(a, a), a)); }
PM_DEV Fr wmul4(const Fr& a) {
  const Fr t = fe_add<FrP>(a, a);                        // limbs < 2^30 + ., value < 2.04 r
  return fe_reduce_weak<FrP>(fe_add<FrP>(t, t));         // limbs < 2^31 + ., value < 4.08 r
}
PM_DEV Fr wmul9(const Fr& a) {
  const Fr f = wmul4(a);
  return fe_reduce_weak<FrP>(fe_add<FrP>(fe_add<FrP>(f, f), a));   // 2 (4a) + a
}
PM_DEV Fr wsqr(const Fr& a) { return fe_sqr<FrP>(a); }
// f (f - 1)(f - 2)(f - 3) = u (u + 2) with u = f^2 - 3 f: zero exactly on the quads 0..3; one squaring and one
// product instead of three products
PM_DEV Fr wdelta(const Fr& f, const WidgetConsts& c) {
  const Fr u = wsub(wsqr(f), wmul3(f));
  return wmul(fe_add<FrP>(u, fr_limbs(c.c2)), u);          // first operand unreduced: limbs < 2^30 + ., value < 2.02 r
}

// Where the 4 m coset points of a launch sit.  Interleaved (one GPU, pm_plonk_quotient_dev): point 4 k + s = g w4^s w^k, the same
// polynomial at w X is index + 4 (wrapping at `wrap`).  PLANAR (the distributed prover, prover_dist.hip.h): four planes [s][m],
// plane s = the rank's rows of sub-coset g w4^s H in the block-transposed order of pm_fr_ntt_fourstep_batch_dev (position
// k1_local N2 + k2 holds k = k2 N1 + k1), so w X is index + N2 -- one row down -- and the row below the rank's last one
// is the halo row the transform delivered with it (four planes of N2 per polynomial; for the last rank it is rank 0's first
// row, where k + 1 means k2 + 1: `rot`).
struct QuotLayout {
  u32 log_m, n2, rot;
  const u32x4* halo_w[4];   // a, b, (unused), d
  const u32x4* halo_z;
};
template <bool WIDGETS, bool PLANAR>
__global__ void __launch_bounds__(256, 2) quotient_kernel(const QuotPtrs p, const RoundConsts kc, const WidgetConsts wc,
                                                       size_t n4, size_t wrap, const QuotLayout L) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;   // a multiple of 4: i mod 4 is fixed per thread
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 r4 = (u32)(t0 & 3);
  Fr zhi;
#pragma unroll
  for (int l = 0; l < 9; ++l)
    zhi.l[l] = r4 == 0 ? kc.zh_inv[0][l] : (r4 == 1 ? kc.zh_inv[1][l] : (r4 == 2 ? kc.zh_inv[2][l] : kc.zh_inv[3][l]));
  const Fr gamma = fr_limbs(kc.gamma);
  for (size_t i = t0; i < n4; i += stride) {
    size_t inext = i + 4 < wrap ? i + 4 : i + 4 - wrap;   // wrap = n4, or n4 + 4 when the rows carry a halo
    bool in_halo = false;
    if (PLANAR) {
      const u32 sp = (u32)(i >> L.log_m), pos = (u32)(i & (((size_t)1 << L.log_m) - 1)), m_ = 1u << L.log_m;
#pragma unroll
      for (int l = 0; l < 9; ++l)
        zhi.l[l] = sp == 0 ? kc.zh_inv[0][l] : (sp == 1 ? kc.zh_inv[1][l] : (sp == 2 ? kc.zh_inv[2][l] : kc.zh_inv[3][l]));
      in_halo = pos + L.n2 >= m_;
      inext = in_halo ? (size_t)sp * L.n2 + ((pos + L.n2 - m_ + L.rot) & (L.n2 - 1)) : i + L.n2;
    }
    Fr w[4], f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = to_dev(ld_canon(p.w[j], i));
    // arithmetic identity, ABI form: five (1, <2) products and one canonical load -> (6, <11)
    Fr g = fe_mul<FrP>(ld_canon(p.q_m, i), fe_mul<FrP>(w[0], w[1]));
    g = fe_add<FrP>(g, fe_mul<FrP>(ld_canon(p.q_l, i), w[0]));
    g = fe_add<FrP>(g, fe_mul<FrP>(ld_canon(p.q_r, i), w[1]));
    g = fe_add<FrP>(g, fe_mul<FrP>(ld_canon(p.q_o, i), w[2]));
    g = fe_add<FrP>(g, fe_mul<FrP>(ld_canon(p.q_4, i), w[3]));
    g = fe_add<FrP>(g, ld_canon(p.q_c, i));
    if (p.q_arith) g = fe_mul<FrP>(fe_norm<FrP>(g), to_dev(ld_canon(p.q_arith, i)));   // ABI x device -> ABI (1, <2)
    if (WIDGETS) {
      // the rows' other gate kinds; "next" = the same polynomial at w X = index + 4 on the 4n coset
      const Fr a = w[0], b = w[1], c = w[2], d = w[3];
      const Fr an = to_dev(ld_canon(PLANAR && in_halo ? L.halo_w[0] : p.w[0], inext)),
               bn = to_dev(ld_canon(PLANAR && in_halo ? L.halo_w[1] : p.w[1], inext)),
               dn = to_dev(ld_canon(PLANAR && in_halo ? L.halo_w[3] : p.w[3], inext));
      Fr wsum = fe_zero<FrP>();
      if (p.q_range) {
        Fr t = wdelta(wsub(c, wmul4(d)), wc);
        t = wadd(t, wmul(wdelta(wsub(b, wmul4(c)), wc), fr_limbs(wc.range_k[0])));
        t = wadd(t, wmul(wdelta(wsub(a, wmul4(b)), wc), fr_limbs(wc.range_k[1])));
        t = wadd(t, wmul(wdelta(wsub(dn, wmul4(a)), wc), fr_limbs(wc.range_k[2])));
        t = wmul(t, fr_limbs(wc.range_sep));
        wsum = wadd(wsum, wmul(to_dev(ld_canon(p.q_range, i)), t));
      }
      if (p.q_logic) {
        const Fr qa = wsub(an, wmul4(a)), qb = wsub(bn, wmul4(b)), qd = wsub(dn, wmul4(d));
        const Fr qc = to_dev(ld_canon(p.q_c, i));
        Fr t = wdelta(qa, wc);
        t = wadd(t, wmul(wdelta(qb, wc), fr_limbs(wc.logic_k[0])));
        t = wadd(t, wmul(wdelta(qd, wc), fr_limbs(wc.logic_k[1])));
        t = wadd(t, wmul(wsub(c, wmul(qa, qb)), fr_limbs(wc.logic_k[2])));
        // delta_xor_and(qa, qb, w = c, qd, q_c)
        const Fr s = wadd(qa, qb);
        Fr in = wadd(wsub(wmul4(c), wmul2(wmul9(s))), fr_limbs(wc.c81));                         // 4w - 18(a+b) + 81
        in = wadd(wmul(c, in), wmul2(wmul9(wadd(wsqr(qa), wsqr(qb)))));                          // w(..) + 18(a^2+b^2)
        in = wadd(wsub(in, wmul(s, fr_limbs(wc.c81))), fr_limbs(wc.c83));                        // - 81(a+b) + 83
        const Fr ff = wmul(c, in);
        const Fr e = wsub(wmul3(wadd(s, qd)), wadd(ff, ff));                                    // 3(a+b+c) - 2f
        const Fr bb = wmul(qc, wsub(wmul9(qd), wmul3(s)));                                      // q_c (9c - 3(a+b))
        t = wadd(t, wmul(wadd(bb, e), fr_limbs(wc.logic_k[3])));
        t = wmul(t, fr_limbs(wc.logic_sep));
        wsum = wadd(wsum, wmul(to_dev(ld_canon(p.q_logic, i)), t));
      }
      if (p.q_fixed) {
        const Fr xb = to_dev(ld_canon(p.q_l, i)), yb = to_dev(ld_canon(p.q_r, i)), xyb = to_dev(ld_canon(p.q_c, i));
        const Fr one = fr_limbs(wc.c1);
        const Fr bit = wsub(dn, wadd(d, d));
        Fr t = wmul(wmul(bit, wsub(bit, one)), wadd(bit, one));                                 // bit (bit-1)(bit+1)
        const Fr ya = wadd(wmul(wsqr(bit), wsub(yb, one)), one);
        const Fr xa = wmul(xb, bit);
        t = wadd(t, wmul(wsub(wmul(bit, xyb), c), fr_limbs(wc.fixed_k[0])));
        const Fr dxy = wmul(wmul(wmul(c, a), b), fr_limbs(wc.edwards_d));
        const Fr xacc = wsub(wadd(an, wmul(an, dxy)), wadd(wmul(a, ya), wmul(b, xa)));
        const Fr yacc = wsub(wsub(bn, wmul(bn, dxy)), wadd(wmul(b, ya), wmul(a, xa)));
        t = wadd(t, wmul(xacc, fr_limbs(wc.fixed_k[1])));
        t = wadd(t, wmul(yacc, fr_limbs(wc.fixed_k[2])));
        t = wmul(t, fr_limbs(wc.fixed_sep));
        wsum = wadd(wsum, wmul(to_dev(ld_canon(p.q_fixed, i)), t));
      }
      if (p.q_var) {
        const Fr y1x2 = wmul(b, c), y1y2 = wmul(b, d), x1x2 = wmul(a, c);
        Fr t = wsub(wmul(a, d), dn);                                                             // x1 y2 - x1y2
        const Fr dd = wmul(wmul(dn, y1x2), fr_limbs(wc.edwards_d));
        const Fr x3 = wsub(wadd(dn, y1x2), wadd(an, wmul(an, dd)));
        const Fr y3 = wsub(wadd(y1y2, x1x2), wsub(bn, wmul(bn, dd)));
        t = wadd(t, wmul(x3, fr_limbs(wc.var_k[0])));
        t = wadd(t, wmul(y3, fr_limbs(wc.var_k[1])));
        t = wmul(t, fr_limbs(wc.var_sep));
        wsum = wadd(wsum, wmul(to_dev(ld_canon(p.q_var, i)), t));
      }
      g = fe_add<FrP>(g, wmul(wsum, fr_limbs(kc.one_abi)));                                      // device x 2^256 -> ABI
    }
    g = fe_norm<FrP>(fe_add<FrP>(g, ld_canon(p.pi, i)));   // limbs back to (1)
    // permutation identity
    const Fr x = ld_canon(p.x, i);
    const Fr z = ld_canon(p.z, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = perm_factor(w[j], x, kc.beta_k[j], gamma);
    const Fr idz = fe_mul<FrP>(z, prod4(f[0], f[1], f[2], f[3]));                          // ABI (1, <2)
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = perm_factor(w[j], ld_canon(p.s[j], i), kc.beta, gamma);
    const Fr cpz = fe_mul<FrP>(ld_canon(PLANAR && in_halo ? L.halo_z : p.z, inext), prod4(f[0], f[1], f[2], f[3]));   // ABI (1, <2)
    // idz - cpz + 3r: (4, <5); times alpha -> ABI (1, <2)
    g = fe_add<FrP>(g, fe_mul<FrP>(fe_sub<FrP, 3, 1>(idz, cpz), fr_limbs(kc.alpha)));
    // (z - 1) l1 alpha^2:  z - 1 + 2r is (4, <3); product with ABI l1 is 2^251, alpha2 restores 2^256
    const Fr zm1 = fe_sub<FrP, 2, 1>(z, fr_limbs(kc.one_abi));
    g = fe_add<FrP>(g, fe_mul<FrP>(fe_mul<FrP>(zm1, ld_canon(p.l1, i)), fr_limbs(kc.alpha2)));
    // g: value < 18 r, limbs < 3 * 2^29 + 16
    st_canon(p.out, i, fe_mul<FrP>(g, zhi));
  }
}

// sigma_j(w^i) = k_j' w^i' for a slice of the copy permutation given as wire positions q = j' n + i' (preprocessing; r01 - r04
// gathered these on the host: 4 n field products or a 4 n x 32-byte round trip through host memory)
struct SigmaConsts {
  u32 ks[4][9];   // 1, k_1, k_2, k_3 in device form
};
__global__ void __launch_bounds__(256) sigma_eval_kernel(const long long* idx, size_t count, const u32x4* tlo, const u32x4* thi, u32 h,
                                                         u32 log_n, const SigmaConsts kc, u32x4* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t mask_n = ((size_t)1 << log_n) - 1, mask_lo = ((size_t)1 << h) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const size_t q = (size_t)idx[i], jj = (q >> log_n) & 3, ii = q & mask_n;
    Fr v = fe_mul<FrP>(ld_canon(tlo, ii & mask_lo), to_dev(ld_canon(thi, ii >> h)));   // ABI x device -> ABI, (1, <2)
    const Fr k = jj == 0 ? fr_limbs(kc.ks[0]) : (jj == 1 ? fr_limbs(kc.ks[1]) : (jj == 2 ? fr_limbs(kc.ks[2]) : fr_limbs(kc.ks[3])));
    st_canon(out, i, fe_mul<FrP>(v, k));
  }
}

// The distributed prover's way onto the 4n coset (prover_dist.hip.h): out[(4 j + s) m + i] = src_j[i] g_s^(lo + i) for the P
// polynomials of a batch and the four sub-cosets -- the inputs of 4 P size-n transforms over the ranks, one launch.
struct CosetExpandArgs {
  const u32x4* src[8];
  u32 count;
};
__global__ void __launch_bounds__(256) coset_expand_kernel(const CosetExpandArgs a, const u32x4* gs_pow /* [4][m] */, size_t m,
                                                           u32x4* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    Fr g[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) g[s_] = to_dev(ld_canon(gs_pow, (size_t)s_ * m + i));
    for (u32 j = 0; j < a.count; ++j) {
      const Fr v = ld_canon(a.src[j], i);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) st_canon(out, ((size_t)4 * j + s_) * m + i, fe_mul<FrP>(v, g[s_]));   // ABI x device -> ABI
    }
  }
}
static HFr load_fr(const uint64_t v[4]) {
  HFr r;
  memcpy(r.l, v, 32);
  return r;
}
static void fill_round_consts(RoundConsts& kc, const HFr& alpha, const HFr& beta, const HFr& gamma,
                              const uint64_t k[3][4], const uint64_t zh_inv[4][4]) {
  const host::Field<4>& F = host::FR();
  memset(&kc, 0, sizeof kc);
  to_limbs29_shift(kc.beta_k[0], beta, 2);
  for (int j = 0; j < 3; ++j) to_limbs29_shift(kc.beta_k[j + 1], host::mul(beta, load_fr(k[j]), F), 2);
  to_limbs29_shift(kc.beta, beta, 2);
  to_limbs29_shift(kc.gamma, gamma, 1);
  to_limbs29_shift(kc.alpha, alpha, 1);
  to_limbs29_shift(kc.alpha2, host::mul(alpha, alpha, F), 2);
  to_limbs29_shift(kc.one_abi, host::one(F), 0);
  if (zh_inv)
    for (int j = 0; j < 4; ++j) to_limbs29_shift(kc.zh_inv[j], load_fr(zh_inv[j]), 1);
}
static unsigned grid_for(const pm_ctx* ctx, size_t n) {
  return (unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cus * 16);
}

int coset_expand(pm_ctx* ctx, const void* const* d_src, uint32_t count, const void* d_gs_pow, size_t m, void* d_out) {
  if (!ctx || !d_src || count == 0 || count > 8 || !d_gs_pow || !d_out) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  CosetExpandArgs a;
  memset(&a, 0, sizeof a);
  a.count = count;
  for (uint32_t j = 0; j < count; ++j) a.src[j] = (const u32x4*)d_src[j];
  PM_HIP(ctx, hipSetDevice(ctx->device));
  ProfScope prof(ctx, ctx->stream, "plonk_coset_expand");
  hipLaunchKernelGGL(coset_expand_kernel, dim3(grid_for(ctx, m)), dim3(256), 0, ctx->stream, a, (const u32x4*)d_gs_pow, m,
                     (u32x4*)d_out);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

// out[p] = k_j' w^i' for p < count, q = idx[p] = j' n + i' (host indices, already range-checked by the caller): two-level
// tables of w (2 sqrt(n) entries) built on the device, one gather kernel.  Synchronises the context before it returns.
int sigma_evals_from_index(pm_ctx* ctx, const int64_t* idx, size_t count, uint32_t log_n, const uint64_t omega[4],
                           const uint64_t k[3][4], void* d_out) {
  if (!ctx || !idx || !omega || !k || !d_out) return PM_ERR_BAD_ARG;
  if (count == 0) return PM_OK;
  const host::Field<4>& F = host::FR();
  const uint32_t h = (log_n + 1) / 2;
  const size_t nlo = (size_t)1 << h, nhi = (size_t)1 << (log_n - h);
  void *d_idx = nullptr, *d_lo = nullptr, *d_hi = nullptr;
  struct Free3 {
    pm_ctx* c;
    void **a, **b, **d;
    ~Free3() {
      for (void** p : {a, b, d})
        if (*p) (void)pm_dev_free(c, *p);
    }
  } free3{ctx, &d_idx, &d_lo, &d_hi};
  int rc = pm_dev_alloc(ctx, count * 8, &d_idx);
  if (!rc) rc = pm_dev_alloc(ctx, nlo * 32, &d_lo);
  if (!rc) rc = pm_dev_alloc(ctx, nhi * 32, &d_hi);
  if (rc) return rc;
  const HFr w = load_fr(omega), one = host::one(F);
  host::u64 e[1] = {(host::u64)nlo};
  const HFr step = host::pow<4>(w, e, 1, F);
  rc = pm_fr_powers_dev(ctx, w.l, one.l, nlo, d_lo, nullptr);
  if (!rc) rc = pm_fr_powers_dev(ctx, step.l, one.l, nhi, d_hi, nullptr);
  if (!rc) rc = pm_dev_upload(ctx, d_idx, idx, count * 8);
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    SigmaConsts kc;
    to_limbs29_shift(kc.ks[0], one, 1);
    for (int j = 0; j < 3; ++j) to_limbs29_shift(kc.ks[j + 1], load_fr(k[j]), 1);
    PM_HIP(ctx, hipSetDevice(ctx->device));
    ProfScope prof(ctx, ctx->stream, "plonk_sigma_evals");
    hipLaunchKernelGGL(sigma_eval_kernel, dim3(grid_for(ctx, count)), dim3(256), 0, ctx->stream, (const long long*)d_idx, count,
                       (const u32x4*)d_lo, (const u32x4*)d_hi, h, log_n, kc, (u32x4*)d_out);
    PM_HIP(ctx, hipGetLastError());
  }
  return pm_sync(ctx);   // the temporaries go away when this returns
}


}  // namespace pm

using namespace pm;

extern "C" int pm_fr_powers_dev(pm_ctx* ctx, const uint64_t base[4], const uint64_t scale[4], size_t n, void* d_out,
                                void* hip_stream) {
  if (!ctx || !base || !scale) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  if (!d_out) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const unsigned blocks = grid_for(ctx, n);
  const u64 T = (u64)blocks * 256;
  NttConsts c;
  memset(&c, 0, sizeof c);
  const HFr b = load_fr(base);
  to_limbs29_shift(c.w8[0], b, 1);
  to_limbs29_shift(c.w8[1], hfr_pow_u64(b, T), 1);
  to_limbs29_shift(c.scale, load_fr(scale), 1);
  to_limbs29_shift(c.one, host::one(host::FR()), 0);
  ProfScope prof(ctx, st, "fr_powers");
  hipLaunchKernelGGL(powers_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)d_out, n, c);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_fr_lincomb_dev(pm_ctx* ctx, uint32_t k, const void* const* d_vecs, const uint64_t* coeffs, size_t n,
                                 void* d_out, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (k == 0 || k > PM_LINCOMB_MAX) return set_err(ctx, PM_ERR_BAD_ARG, "k must be in 1..PM_LINCOMB_MAX");
  if (n == 0) return PM_OK;
  if (!d_vecs || !coeffs || !d_out) return set_err(ctx, PM_ERR_BAD_ARG, "null pointer");
  LincombArgs a;
  memset(&a, 0, sizeof a);
  a.k = k;
  for (uint32_t j = 0; j < k; ++j) {
    if (!d_vecs[j]) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
    a.v[j] = (const u32x4*)d_vecs[j];
    to_limbs29_shift(a.c[j], load_fr(coeffs + 4 * j), 1);
  }
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  ProfScope prof(ctx, st, "fr_lincomb");
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, st, a, (u32x4*)d_out, n);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_plonk_perm_terms_dev(pm_ctx* ctx, const pm_plonk_perm_args* args, size_t n, void* d_num,
                                       void* d_den, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!args) return set_err(ctx, PM_ERR_BAD_ARG, "null args");
  if (n == 0) return PM_OK;
  PermPtrs p;
  for (int j = 0; j < 4; ++j) {
    p.w[j] = (const u32x4*)args->wires[j];
    p.s[j] = (const u32x4*)args->sigmas[j];
    if (!p.w[j] || !p.s[j]) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  }
  p.roots = (const u32x4*)args->roots;
  p.num = (u32x4*)d_num;
  p.den = (u32x4*)d_den;
  if (!p.roots || !p.num || !p.den) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  RoundConsts kc;
  fill_round_consts(kc, host::zero<4>(), load_fr(args->beta), load_fr(args->gamma), args->k, nullptr);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  ProfScope prof(ctx, st, "plonk_perm_terms");
  hipLaunchKernelGGL(perm_terms_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, st, p, kc, n);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_plonk_quotient_dev(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t n, void* d_out,
                                     void* hip_stream) {
  return pm::plonk_quotient_rows(ctx, args, n, false, d_out, hip_stream);
}
int pm::plonk_quotient_rows(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t n, bool halo, void* d_out,
                            void* hip_stream) {
  return plonk_quotient_layout(ctx, args, n, halo, nullptr, d_out, hip_stream);
}
int pm::plonk_quotient_layout(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t n, bool halo, const QuotPlanar* planar,
                              void* d_out, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!args) return set_err(ctx, PM_ERR_BAD_ARG, "null args");
  if (n == 0) return PM_OK;
  if (n & (n - 1)) return set_err(ctx, PM_ERR_LENGTH, "n must be a power of two");
  QuotPtrs p;
  for (int j = 0; j < 4; ++j) {
    p.w[j] = (const u32x4*)args->wires[j];
    p.s[j] = (const u32x4*)args->sigmas[j];
  }
  p.z = (const u32x4*)args->z;
  p.q_m = (const u32x4*)args->q_m;
  p.q_l = (const u32x4*)args->q_l;
  p.q_r = (const u32x4*)args->q_r;
  p.q_o = (const u32x4*)args->q_o;
  p.q_4 = (const u32x4*)args->q_4;
  p.q_c = (const u32x4*)args->q_c;
  p.q_arith = (const u32x4*)args->q_arith;
  p.q_range = (const u32x4*)args->q_range;
  p.q_logic = (const u32x4*)args->q_logic;
  p.q_fixed = (const u32x4*)args->q_fixed_group_add;
  p.q_var = (const u32x4*)args->q_variable_group_add;
  p.pi = (const u32x4*)args->pi;
  p.l1 = (const u32x4*)args->l1;
  p.x = (const u32x4*)args->x;
  p.out = (u32x4*)d_out;
  const void* all[] = {p.w[0], p.w[1], p.w[2], p.w[3], p.s[0], p.s[1], p.s[2], p.s[3], p.z,  p.q_m,
                       p.q_l,  p.q_r,  p.q_o,  p.q_4,  p.q_c,  p.pi,   p.l1,   p.x,    p.out};
  for (const void* q : all)
    if (!q) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  RoundConsts kc;
  fill_round_consts(kc, load_fr(args->alpha), load_fr(args->beta), load_fr(args->gamma), args->k, args->zh_inv);
  const bool widgets = p.q_range || p.q_logic || p.q_fixed || p.q_var;
  WidgetConsts wc;
  memset(&wc, 0, sizeof wc);
  if (widgets) {
    const host::Field<4>& F = host::FR();
    auto dev = [&](u32* dst, const HFr& v) { to_limbs29_shift(dst, v, 1); };
    const struct { u32* dst; u64 v; } small[] = {{wc.c1, 1}, {wc.c2, 2}, {wc.c3, 3}, {wc.c4, 4}, {wc.c9, 9},
                                                 {wc.c18, 18}, {wc.c81, 81}, {wc.c83, 83}};
    for (const auto& c : small) dev(c.dst, host::from_u64(c.v, F));
    // JubJub d = -(10240 / 10241)
    dev(wc.edwards_d, host::sub(host::zero<4>(), host::mul(host::from_u64(10240, F), host::inv(host::from_u64(10241, F), F), F), F));
    auto sep_powers = [&](const uint64_t sep[4], u32* s_out, u32 (*k_out)[9], int nk) {
      const HFr s_ = load_fr(sep), kappa = host::mul(s_, s_, F);
      dev(s_out, s_);
      HFr kp = kappa;
      for (int i = 0; i < nk; ++i) {
        dev(k_out[i], kp);
        kp = host::mul(kp, kappa, F);
      }
    };
    sep_powers(args->range_sep, wc.range_sep, wc.range_k, 3);
    sep_powers(args->logic_sep, wc.logic_sep, wc.logic_k, 4);
    sep_powers(args->fixed_sep, wc.fixed_sep, wc.fixed_k, 3);
    sep_powers(args->var_sep, wc.var_sep, wc.var_k, 2);
  }
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  ProfScope prof(ctx, st, "plonk_quotient");
  QuotLayout L;
  memset(&L, 0, sizeof L);
  const size_t wrap = 4 * n + (halo ? 4 : 0);
  if (planar) {
    while (((size_t)1 << L.log_m) < n) ++L.log_m;
    L.n2 = planar->n2;
    L.rot = planar->rot;
    for (int j = 0; j < 4; ++j) L.halo_w[j] = (const u32x4*)planar->halo_w[j];
    L.halo_z = (const u32x4*)planar->halo_z;
    if (!L.halo_z || !L.halo_w[0] || !L.halo_w[1] || !L.halo_w[3] || L.n2 == 0 || (L.n2 & (L.n2 - 1)) || L.n2 > n)
      return set_err(ctx, PM_ERR_BAD_ARG, "planar quotient layout: halo rows missing or row length not a power of two <= rows");
    if (widgets)
      hipLaunchKernelGGL((quotient_kernel<true, true>), dim3(grid_for(ctx, 4 * n)), dim3(256), 0, st, p, kc, wc, 4 * n, wrap, L);
    else
      hipLaunchKernelGGL((quotient_kernel<false, true>), dim3(grid_for(ctx, 4 * n)), dim3(256), 0, st, p, kc, wc, 4 * n, wrap, L);
  } else if (widgets) {
    hipLaunchKernelGGL((quotient_kernel<true, false>), dim3(grid_for(ctx, 4 * n)), dim3(256), 0, st, p, kc, wc, 4 * n, wrap, L);
  } else {
    hipLaunchKernelGGL((quotient_kernel<false, false>), dim3(grid_for(ctx, 4 * n)), dim3(256), 0, st, p, kc, wc, 4 * n, wrap, L);
  }
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}
