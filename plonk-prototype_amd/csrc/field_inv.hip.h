// Modular inversion in Fr / Fp by an optimised binary GCD (T. Pornin, "Optimized Binary GCD for Modular Inversion",
// 2020) in the limb form of fields.hip.h -- constant work, no data-dependent branches, so the lanes of a wave stay
// together.  x^(m-2) costs 255 (Fr) / 381 (Fp) dependent squarings: ~76 k instructions per Fr inversion and the
// latency floor of every kernel that needs one (r02: 57 % of the batch-inversion kernel).  Here: ROUNDS =
// ceil((2 bits(m) - 1) / W) rounds (18 / 28), each W plain binary-GCD steps on (2 W + 2)-bit approximations of a and b
// (their low W bits, exact, and their top W + 2 bits) that accumulate a 2 x 2 matrix of 32-bit factors, followed by
// one limb-wise update of (a, b) -- an exact division by 2^W = one limb -- and of (u, v) modulo m (one
// Montgomery-style limb reduction).  ~16 k instructions for Fr.
//
// Invariants (integers, y the input): a = y u, b = y v (mod m); start a = y, b = m, u = 1, v = 0; after all rounds
// a = 0, b = gcd = 1 and v = 1 / y; y = 0 gives v = 0 ("zeros stay zero", util::batch_inversion).
#pragma once
#include "fields.hip.h"

namespace pm {

template <class P>
struct InvParams;
template <>
struct InvParams<FrP> {
  static constexpr int BITS = 255;
};
template <>
struct InvParams<FpP> {
  static constexpr int BITS = 381;
};

// value < 2 m, any limbs < 2^32 -> the canonical representative, every limb < 2^W
template <class P>
PM_DEV Fe<P> fe_canon_limbs(const Fe<P>& a) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  Fe<P> x = fe_norm_full<P>(a), d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    const int32_t t = (int32_t)x.l[i] - (int32_t)M.v[i] + c;
    d.l[i] = (u32)t & MASK;
    c = t >> W;
  }
  const int32_t top = (int32_t)x.l[N - 1] - (int32_t)M.v[N - 1] + c;
  d.l[N - 1] = (u32)top;
#pragma unroll
  for (int i = 0; i < N; ++i) x.l[i] = top < 0 ? x.l[i] : d.l[i];
  return x;
}

// y canonical (fe_canon_limbs).  Returns v >= 0 with v y = 1 (mod m) as plain integers (no Montgomery factor
// involved), limbs < 2^W except the top one, value < 64 m; 0 -> 0.  A caller whose y carries a factor F (y = x F)
// gets x^-1 / F and multiplies by what its form needs (fe_inv_dev below).
template <class P>
PM_DEV Fe<P> fe_inv_int(const Fe<P>& y) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  constexpr u32 NINV = Consts<P>::neg_inv();
  constexpr int ROUNDS = (2 * InvParams<P>::BITS - 1 + W - 1) / W;
  constexpr int S = 64 - 2 * W;   // bits of the third limb that fit the 64-bit window
  static_assert(N >= 3 && W <= 30 && S >= 2, "limb geometry");
  u32 a[N], b[N];
  int32_t u[N], v[N];   // limbs 0 .. N-2 in [0, 2^W), the top limb carries the sign
#pragma unroll
  for (int i = 0; i < N; ++i) {
    a[i] = y.l[i];
    b[i] = M.v[i];
    u[i] = i == 0 ? 1 : 0;
    v[i] = 0;
  }
  for (int round = 0; round < ROUNDS; ++round) {
    // ---- a == 0 in every lane of the wave: the remaining rounds only multiply b's row of the matrix by 2^W and divide
    // it out again (b and v keep their values, v up to a multiple of m), so the wave leaves together.  ROUNDS covers the
    // worst case (2 bits(m) - 1 steps); random inputs need ~1.41 bits(m) steps (Fr: 13 of the 18 rounds, r06).  The vote
    // is wave-uniform: no divergence, and the result does not depend on which lanes share the wave (the skipped rounds
    // change the representative of v, not its class; callers reduce it)
    {
      u32 any_a = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) any_a |= a[i];
      if (__all(any_a == 0)) break;
    }
    // ---- approximations: low W bits | top W + 2 bits of max(a, b) (both numbers cut at the same position)
    u32 a2 = 0, a1 = 0, a0 = 0, b2 = 0, b1 = 0, b0 = 0;
    bool found = false;
#pragma unroll
    for (int i = N - 1; i >= 2; --i) {
      const bool nz = (a[i] | b[i]) != 0;
      const bool take = nz && !found;
      a2 = take ? a[i] : a2;
      a1 = take ? a[i - 1] : a1;
      a0 = take ? a[i - 2] : a0;
      b2 = take ? b[i] : b2;
      b1 = take ? b[i - 1] : b1;
      b0 = take ? b[i - 2] : b0;
      found = found || nz;
    }
    // both below 2^(2 W + 2): the approximation is the number itself
    u32 above = 0;
#pragma unroll
    for (int i = 3; i < N; ++i) above |= a[i] | b[i];
    const bool exact = above == 0 && (a[2] | b[2]) < 4u;
    u64 xa = ((u64)a2 << (W + S)) | ((u64)a1 << S) | (u64)(a0 >> (W - S));
    u64 xb = ((u64)b2 << (W + S)) | ((u64)b1 << S) | (u64)(b0 >> (W - S));
    const int lz = __clzll((long long)(xa | xb | 1ull));
    u64 ah = ((xa << lz) >> (64 - (W + 2)) << W) | a[0];
    u64 bh = ((xb << lz) >> (64 - (W + 2)) << W) | b[0];
    const u64 ae = (u64)a[0] | ((u64)a[1] << W) | ((u64)(a[2] & 3u) << (2 * W));
    const u64 be = (u64)b[0] | ((u64)b[1] << W) | ((u64)(b[2] & 3u) << (2 * W));
    ah = exact ? ae : ah;
    bh = exact ? be : bh;
    // ---- W binary-GCD steps on the approximations, recorded as (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^W
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const bool odd = (ah & 1ull) != 0;
      const bool sw = odd && ah < bh;
      const u64 ta = sw ? bh : ah, tb = sw ? ah : bh;
      const int32_t tf0 = sw ? f1 : f0, tf1 = sw ? f0 : f1, tg0 = sw ? g1 : g0, tg1 = sw ? g0 : g1;
      ah = (ta - (odd ? tb : 0ull)) >> 1;
      bh = tb;
      f0 = tf0 - (odd ? tf1 : 0);
      g0 = tg0 - (odd ? tg1 : 0);
      f1 = tf1 << 1;
      g1 = tg1 << 1;
    }
    // ---- (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^W: the low limb of both sums is zero by construction
    u32 na[N], nb[N];
    {
      long long ca = 0, cb = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        ca += (long long)f0 * (long long)a[i] + (long long)g0 * (long long)b[i];
        cb += (long long)f1 * (long long)a[i] + (long long)g1 * (long long)b[i];
        if (i > 0) {
          na[i - 1] = (u32)ca & MASK;
          nb[i - 1] = (u32)cb & MASK;
        }
        ca >>= W;
        cb >>= W;
      }
      na[N - 1] = (u32)ca;
      nb[N - 1] = (u32)cb;
      // a negative result is negated together with its row of the matrix (the approximations can get a comparison wrong)
      const bool sa = ca < 0, sb = cb < 0;
      const u32 ma = sa ? 0xffffffffu : 0u, mb = sb ? 0xffffffffu : 0u;
      u32 cya = sa ? 1u : 0u, cyb = sb ? 1u : 0u;
#pragma unroll
      for (int i = 0; i < N - 1; ++i) {
        const u32 ta = ((na[i] ^ ma) & MASK) + cya, tb = ((nb[i] ^ mb) & MASK) + cyb;
        a[i] = ta & MASK;
        b[i] = tb & MASK;
        cya = ta >> W;
        cyb = tb >> W;
      }
      a[N - 1] = (na[N - 1] ^ ma) + cya;
      b[N - 1] = (nb[N - 1] ^ mb) + cyb;
      f0 = sa ? -f0 : f0;
      g0 = sa ? -g0 : g0;
      f1 = sb ? -f1 : f1;
      g1 = sb ? -g1 : g1;
    }
    // ---- (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^W mod m: add the multiple of m that clears the low limb
    {
      const u32 lu = ((u32)f0 * (u32)u[0] + (u32)g0 * (u32)v[0]) & MASK, lv = ((u32)f1 * (u32)u[0] + (u32)g1 * (u32)v[0]) & MASK;
      const u32 qu = (lu * NINV) & MASK, qv = (lv * NINV) & MASK;
      long long cu = 0, cv = 0;
      int32_t nu[N], nv[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        cu += (long long)f0 * (long long)u[i] + (long long)g0 * (long long)v[i] + (long long)qu * (long long)M.v[i];
        cv += (long long)f1 * (long long)u[i] + (long long)g1 * (long long)v[i] + (long long)qv * (long long)M.v[i];
        if (i > 0) {
          nu[i - 1] = (int32_t)((u32)cu & MASK);
          nv[i - 1] = (int32_t)((u32)cv & MASK);
        }
        cu >>= W;
        cv >>= W;
      }
      nu[N - 1] = (int32_t)cu;
      nv[N - 1] = (int32_t)cv;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        u[i] = nu[i];
        v[i] = nv[i];
      }
    }
  }
  // |v| < (ROUNDS + 1) m: add 32 m and carry
  constexpr Limbs<N> M32 = Consts<P>::k_mod(32);
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.l[i] = (u32)v[i] + M32.v[i];
  return fe_norm_full<P>(r);
}

// x in the device Montgomery form (x 2^(W N), any representative < 2 m) -> 1 / x in the same form; 0 -> 0
template <class P>
PM_DEV Fe<P> fe_inv_dev(const Fe<P>& x) {
  // integer inverse of x R' is x^-1 / R'; times R'^3 / R' (one product with the constant R'^3) = x^-1 R'
  return fe_mul<P>(fe_inv_int<P>(fe_canon_limbs<P>(x)), fe_pow2<P, 3 * P::W * P::N>());
}

}  // namespace pm
