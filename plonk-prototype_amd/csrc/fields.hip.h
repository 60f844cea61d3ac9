// BLS12-381 Fr / Fp Montgomery arithmetic for gfx950 (device side), carry-free form.
//
// Why unsaturated limbs.  Measured on MI355X (tools/ubench.hip, profiles/r01_ubench.txt):
// v_mad_u64_u32 issues at the same ~4.4 cycles per wave-instruction as v_add_u32 -- 32-bit
// integer multiply is full rate on CDNA4 -- but gfx950 needs two wait states between a VALU
// that writes a carry (VCC / SGPR pair) and the VALU that consumes it (hipcc pads every
// v_addc with `s_nop 1`).  A classic saturated 8 x 32-bit Montgomery product is one mad plus
// one dependent v_addc per limb product, i.e. a hazard on every second instruction.  With
// radix 2^29 (Fr, 9 limbs) / 2^28 (Fp, 14 limbs) a 64-bit column accumulator absorbs every
// limb product of a column without overflow: the product is pure `acc += (u64)a*b`
// (v_mad_u64_u32, no carry-out), additions are limb-wise, and no instruction ever reads a
// carry flag.  Plain C++ -- no inline asm, the compiler schedules everything.
//
// Value domain: elements are kept in a redundant range (value < V*m, limbs < B*2^W); the
// Montgomery radix R = 2^(W*N) exceeds the modulus by 2^6 (Fr) / 2^11 (Fp), so
// fe_mul(a, b) < a*b/R + m lands below 2m without any conditional subtraction for every
// operand bound used in this library.  Canonical form (the ABI: saturated 32-bit limbs,
// fully reduced, bit-identical to dusk_bls12_381::{Scalar, Fp} memory, ref:Cargo.toml:20;
// Scalar used at ref:allocated_scalar.rs:19) exists only at kernel boundaries:
// fe_unpack() on load, fe_canon_pack() on store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

typedef uint32_t u32;
typedef unsigned long long u64;

#define PM_DEV __device__ __forceinline__
#define PM_HD __host__ __device__ constexpr

// ------------------------------------------------------------------ parameters
// SAT[] = modulus in saturated 32-bit limbs; everything else is derived at compile time.
struct FrP {
  static constexpr int W = 29;   // limb bits
  static constexpr int N = 9;    // limbs, R = 2^261
  static constexpr int NS = 8;   // saturated 32-bit limbs at the ABI
  static constexpr u32 SAT[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
};
struct FpP {
  static constexpr int W = 28;
  static constexpr int N = 14;   // R = 2^392
  static constexpr int NS = 12;
  static constexpr u32 SAT[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                  0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                  0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
};

template <int N>
struct Limbs {
  u32 v[N];
};

// bits [lo, lo+w) of a saturated little-endian limb array
template <int NS>
PM_HD u32 sat_bits(const u32 (&s)[NS], int lo, int w) {
  u64 x = 0;
  int i = lo / 32, sh = lo % 32;
  if (i < NS) x = s[i] >> sh;
  if (i + 1 < NS && sh) x |= (u64)s[i + 1] << (32 - sh);
  return (u32)(x & ((w >= 32) ? 0xffffffffull : ((1ull << w) - 1)));
}

template <class P>
struct Consts {
  static constexpr int W = P::W, N = P::N, NS = P::NS;
  static constexpr u32 MASK = (1u << W) - 1;
  // modulus in radix 2^W
  static PM_HD Limbs<N> mod_limbs() {
    Limbs<N> r{};
    for (int i = 0; i < N; ++i) r.v[i] = sat_bits<NS>(P::SAT, i * W, W);
    return r;
  }
  // -m^-1 mod 2^W (Newton)
  static PM_HD u32 neg_inv() {
    u32 m0 = sat_bits<NS>(P::SAT, 0, W);
    u32 x = 1;
    for (int i = 0; i < 6; ++i) x = x * (2u - m0 * x);
    return (0u - x) & MASK;
  }
  // k*m in radix 2^W (normalised; top limb unbounded)
  static PM_HD Limbs<N> k_mod(u32 k) {
    Limbs<N> m = mod_limbs(), r{};
    u64 c = 0;
    for (int i = 0; i < N; ++i) {
      u64 t = (u64)m.v[i] * k + c;
      r.v[i] = (i == N - 1) ? (u32)t : (u32)(t & MASK);
      c = t >> W;
    }
    return r;
  }
  // 2^e mod m in radix 2^W (compile-time double-and-reduce)
  static PM_HD Limbs<N> pow2_mod(int e) {
    Limbs<N> m = mod_limbs(), x{};
    x.v[0] = 1;
    for (int it = 0; it < e; ++it) {
      u32 c = 0;
      for (int i = 0; i < N; ++i) {  // x = 2x (normalised; the top limb keeps the overflow)
        u32 t = (x.v[i] << 1) | c;
        c = (i == N - 1) ? 0 : (t >> W);
        x.v[i] = (i == N - 1) ? t : (t & MASK);
      }
      bool ge = true;  // x >= m ?
      for (int i = N - 1; i >= 0; --i) {
        if (x.v[i] != m.v[i]) {
          ge = x.v[i] > m.v[i];
          break;
        }
      }
      if (ge) {
        u32 bw = 0;
        for (int i = 0; i < N; ++i) {
          u32 t = x.v[i] - m.v[i] - bw;
          bw = (i == N - 1) ? 0 : ((t >> W) & 1);
          x.v[i] = (i == N - 1) ? t : (t & MASK);
        }
      }
    }
    return x;
  }
  // 2^(W N) - m in radix 2^W (all limbs < 2^W)
  static PM_HD Limbs<N> rbar() {
    Limbs<N> m = mod_limbs(), r{};
    for (int i = 0; i < N; ++i) r.v[i] = MASK - m.v[i];
    r.v[0] += 1;  // m is odd, so this cannot carry
    return r;
  }
  // k*m with every limb but the top biased by 2^(W+E) - 2^E so that a limb-wise
  // `a + bias - b` cannot go negative for b limbs < 2^(W+E) - 2^E (top limb: < top of k*m - 2^E)
  static PM_HD Limbs<N> sub_bias(u32 k, int E) {
    Limbs<N> r = k_mod(k);
    for (int i = 0; i < N - 1; ++i) {
      r.v[i] += (1u << (W + E));
      r.v[i + 1] -= (1u << E);
    }
    return r;
  }
};

template <class P>
struct Fe {
  u32 l[P::N];
};
typedef Fe<FrP> Fr;
typedef Fe<FpP> Fp;

// ------------------------------------------------------------------ trivial ops
template <class P>
PM_DEV Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::N; ++i) r.l[i] = 0;
  return r;
}
// limb-wise sum: value and limb bounds add
template <class P>
PM_DEV Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::N; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
// a - b + K*m.  Requires b limbs < 2^(W+E) - 2^E and b's top limb < top(K*m) - 2^E.
// Result: value < a + K*m, limbs < a + 2^(W+E) + 2^W.
template <class P, int K, int E>
PM_DEV Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  constexpr Limbs<P::N> bias = Consts<P>::sub_bias(K, E);
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::N; ++i) r.l[i] = a.l[i] + (bias.v[i] - b.l[i]);
  return r;
}
// one parallel carry step: limbs < 2^32 in -> limbs < 2^W + 2^(32-W) out (top limb keeps the rest)
template <class P>
PM_DEV Fe<P> fe_norm(const Fe<P>& a) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  Fe<P> r;
  r.l[0] = a.l[0] & MASK;
#pragma unroll
  for (int i = 1; i < N - 1; ++i) r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> W);
  r.l[N - 1] = a.l[N - 1] + (a.l[N - 2] >> W);
  return r;
}
// full sequential carry: every limb but the top < 2^W
template <class P>
PM_DEV Fe<P> fe_norm_full(const Fe<P>& a) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    u32 t = a.l[i] + c;
    r.l[i] = t & MASK;
    c = t >> W;
  }
  r.l[N - 1] = a.l[N - 1] + c;
  return r;
}

// ------------------------------------------------------------------ Montgomery product
// Column-wise (product scanning) with a single 64-bit accumulator and no carries.
// Requires  N * max(a limb) * max(b limb) + (N-1) * 2^(2W) + 2^(64-W) < 2^64, i.e. with
// b normalised (limbs <= 2^W + 8):  a limbs < 6 * 2^29 (Fr) / < 13 * 2^28 (Fp).
// Result: limbs < 2^W (normalised), value < a*b/R + m.
//
// r05: ONE dependent chain per product.  Written as plain `acc += a * b`, hipcc's Reassociate pass adds the late
// operand of every column -- the shifted carry -- last: each column becomes a fresh chain from 0 plus a 64-bit
// addition to fold the carry in (16 extra v_lshl_add_u64 per Fr product, 26 per Fp product, + moves).  Dependent
// and independent v_mad_u64_u32 issue at the same rate at every occupancy (tools/ubench.hip,
// profiles/r05_ubench.txt), so those additions buy nothing.  PM_KEEP gives every partial sum a second use (an
// empty asm: no instruction), Reassociate can no longer linearise the column, and instruction selection folds each
// product with the running accumulator into one v_mad_u64_u32 whose addend is the carry.  The asm is NOT volatile
// (an input-only asm would be: ~150 ordering barriers per product, the scheduler could no longer sink loads to
// their uses and several kernels went from 120 to 350+ VGPRs): it threads a token register `tok` (tied in / out)
// that the product's last statement ties to the result's top limb, so the chain stays live and side-effect free.
// Fr 228 -> 205 VALU instructions per product, Fp 507 -> 460; measured -7.5 % .. -10 % shader cycles per product
// at 1, 2 and 4 waves per SIMD (tools/fe_mul_chain_ab.hip, profiles/r05_fe_mul_chain_ab.txt).  The mads as inline
// asm (clang has no builtin) lose: the hazard recogniser pads every asm -> asm pair with s_nop.  Also measured and
// not adopted there: the shift split into v_alignbit_b32 + 32-bit shift (VOP3 issues like the 64-bit shift), "+q"
// through a mad, and for Fr the subtractive digit q' = acc mod 2^W with a signed accumulator (-2 % more at four
// waves per SIMD, but it needs Ba * Bb < 3.5 where the NTT butterflies hand fe_mul limbs up to 5 * 2^29).
// The r04 form (no chain, q added as a value) lives only where the A/B does: tools/fe_chain.hip.h, tools/fe_mul_chain_ab.hip.
#define PM_KEEP(tok, x) asm("" : "+v"(tok) : "v"(x))
#define PM_KEEP_INIT(tok) asm("" : "=v"(tok))
PM_DEV void fe_mac(u64& acc, u32 a, u32 b, u32& tok) {
  acc += (u64)a * b;
  PM_KEEP(tok, acc);
}

// The reduction half shared by every product routine: K accumulators walk the 2N-1 columns in lock step;
// `prod(k, acc)` adds the limb products of column k (in the order the caller wants them issued), this adds the
// q * m terms, derives the quotient digit and shifts.  Statements of the K chains alternate, so two products of one
// wave interleave instruction by instruction (kernels that run one or two waves per SIMD: DESIGN.md section 4).
template <class P, int K, class F>
PM_DEV void fe_mont_cols(F&& prod, Fe<P>* const (&r)[K]) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  constexpr u32 NINV = Consts<P>::neg_inv();
  u32 q[K][N];
  u64 acc[K];
  u32 tok;
  PM_KEEP_INIT(tok);
#pragma unroll
  for (int c = 0; c < K; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    prod(k, acc, tok);
    if (k < N) {
#pragma unroll
      for (int i = 0; i < k; ++i)
#pragma unroll
        for (int c = 0; c < K; ++c) fe_mac(acc[c], q[c][i], M.v[k - i], tok);
#pragma unroll
      for (int c = 0; c < K; ++c) {
        if (M.v[0] == 1u) {  // Fr: m = 1 mod 2^W, -m^-1 = -1
          q[c][k] = (0u - (u32)acc[c]) & MASK;
          acc[c] += (u64)MASK;  // (acc + q) >> W == (acc + 2^W - 1) >> W: the carry does not wait for q
        } else {
          q[c][k] = ((u32)acc[c] * NINV) & MASK;
          fe_mac(acc[c], q[c][k], M.v[0], tok);
        }
        acc[c] >>= W;
      }
    } else {
#pragma unroll
      for (int i = k - N + 1; i < N; ++i)
#pragma unroll
        for (int c = 0; c < K; ++c) fe_mac(acc[c], q[c][i], M.v[k - i], tok);
#pragma unroll
      for (int c = 0; c < K; ++c) {
        r[c]->l[k - N] = (u32)acc[c] & MASK;
        acc[c] >>= W;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < K; ++c) r[c]->l[N - 1] = (u32)acc[c];
  asm("" : "+v"(r[0]->l[N - 1]) : "v"(tok));  // the token chain ends in a live value
}
// limb products a_i b_(k-i) of column k
template <class P>
PM_DEV void fe_col_mul(int k, u64& acc, const Fe<P>& a, const Fe<P>& b, u32& tok) {
  constexpr int N = P::N;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int j = k - i;
    if (j < 0 || j >= N) continue;
    fe_mac(acc, a.l[i], b.l[j], tok);
  }
}
// ... of a^2 with the cross terms taken once against the pre-doubled copy d = 2a
template <class P>
PM_DEV void fe_col_sqr(int k, u64& acc, const Fe<P>& a, const u32 (&d)[P::N], u32& tok) {
  constexpr int N = P::N;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int j = k - i;
    if (j < 0 || j >= N || i > j) continue;
    fe_mac(acc, a.l[i], (i == j) ? a.l[i] : d[j], tok);
  }
}

template <class P>
PM_DEV Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> t;
  Fe<P>* const out[1] = {&t};
  fe_mont_cols<P, 1>([&](int k, u64* acc, u32& tok) { fe_col_mul<P>(k, acc[0], a, b, tok); }, out);
  return t;
}
// a * b0 / R for a one-limb b0 (< 2^W): the reduction half of fe_mul only (N + N(N-1) limb products instead of
// 2 N^2 - N).  Same bounds and result class as fe_mul.  Used for Montgomery -> integer: x 2^256 * 2^5 / 2^261.
template <class P>
PM_DEV Fe<P> fe_mul_limb(const Fe<P>& a, u32 b0) {
  Fe<P> t;
  Fe<P>* const out[1] = {&t};
  fe_mont_cols<P, 1>(
      [&](int k, u64* acc, u32& tok) {
        if (k < P::N) fe_mac(acc[0], a.l[k], b0, tok);
      },
      out);
  return t;
}
// a^2 with the cross terms taken once against a pre-doubled copy: N(N+1)/2 products instead of
// N^2 for the a*a part (the reduction part is unchanged).  Same bounds as fe_mul(a, a).
template <class P>
PM_DEV Fe<P> fe_sqr(const Fe<P>& a) {
  u32 d[P::N];
#pragma unroll
  for (int i = 0; i < P::N; ++i) d[i] = a.l[i] << 1;
  Fe<P> t;
  Fe<P>* const out[1] = {&t};
  fe_mont_cols<P, 1>([&](int k, u64* acc, u32& tok) { fe_col_sqr<P>(k, acc[0], a, d, tok); }, out);
  return t;
}

// Two independent products / squarings in lock step: the same arithmetic as fe_mul / fe_sqr, the statements of
// the two interleaved.  A wave issues in order and one VALU instruction per ~5.5 cycles when alone on its SIMD
// (profiles/r05_ubench.txt): kernels that run one or two waves per SIMD (the MSM's) want two chains per wave
// (measured: DESIGN.md section 4).
template <class P>
PM_DEV void fe_mul2(const Fe<P>& a0, const Fe<P>& b0, const Fe<P>& a1, const Fe<P>& b1, Fe<P>& r0, Fe<P>& r1) {
  constexpr int N = P::N;
  Fe<P> t0, t1;  // the outputs may alias the inputs
  Fe<P>* const out[2] = {&t0, &t1};
  fe_mont_cols<P, 2>(
      [&](int k, u64* acc, u32& tok) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int j = k - i;
          if (j < 0 || j >= N) continue;
          fe_mac(acc[0], a0.l[i], b0.l[j], tok);
          fe_mac(acc[1], a1.l[i], b1.l[j], tok);
        }
      },
      out);
  r0 = t0;
  r1 = t1;
}
// Three independent products in lock step (the third chain of a round that has three products to offer).
template <class P>
PM_DEV void fe_mul3(const Fe<P>& a0, const Fe<P>& b0, const Fe<P>& a1, const Fe<P>& b1, const Fe<P>& a2, const Fe<P>& b2,
                    Fe<P>& r0, Fe<P>& r1, Fe<P>& r2) {
  constexpr int N = P::N;
  Fe<P> t0, t1, t2;
  Fe<P>* const out[3] = {&t0, &t1, &t2};
  fe_mont_cols<P, 3>(
      [&](int k, u64* acc, u32& tok) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int j = k - i;
          if (j < 0 || j >= N) continue;
          fe_mac(acc[0], a0.l[i], b0.l[j], tok);
          fe_mac(acc[1], a1.l[i], b1.l[j], tok);
          fe_mac(acc[2], a2.l[i], b2.l[j], tok);
        }
      },
      out);
  r0 = t0;
  r1 = t1;
  r2 = t2;
}
// r0 = (a0 b0 + c0 d0) / R with ONE reduction for the two products (both go into the same column accumulators: a
// difference of products costs 3 N^2 limb products instead of 4 N^2 when the subtrahend is negated limb-wise first),
// in lock step with an ordinary product r1 = a1 b1 / R.  Column bound: N (Ba Bb + Bc Bd) 2^(2W) + (N-1) 2^(2W) +
// 2^(64-W) < 2^64, i.e. Ba Bb + Bc Bd < 17 for Fp (limb bounds B in units of 2^W); the value is < (a0 b0 + c0 d0) / R + m.
template <class P>
PM_DEV void fe_mma2(const Fe<P>& a0, const Fe<P>& b0, const Fe<P>& c0, const Fe<P>& d0, const Fe<P>& a1, const Fe<P>& b1,
                    Fe<P>& r0, Fe<P>& r1) {
  constexpr int N = P::N;
  Fe<P> t0, t1;
  Fe<P>* const out[2] = {&t0, &t1};
  fe_mont_cols<P, 2>(
      [&](int k, u64* acc, u32& tok) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int j = k - i;
          if (j < 0 || j >= N) continue;
          fe_mac(acc[0], a0.l[i], b0.l[j], tok);
          fe_mac(acc[1], a1.l[i], b1.l[j], tok);
          fe_mac(acc[0], c0.l[i], d0.l[j], tok);
        }
      },
      out);
  r0 = t0;
  r1 = t1;
}
template <class P>
PM_DEV void fe_sqr2(const Fe<P>& a0, const Fe<P>& a1, Fe<P>& r0, Fe<P>& r1) {
  constexpr int N = P::N;
  u32 d0[N], d1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    d0[i] = a0.l[i] << 1;
    d1[i] = a1.l[i] << 1;
  }
  Fe<P> t0, t1;
  Fe<P>* const out[2] = {&t0, &t1};
  fe_mont_cols<P, 2>(
      [&](int k, u64* acc, u32& tok) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int j = k - i;
          if (j < 0 || j >= N || i > j) continue;
          fe_mac(acc[0], a0.l[i], (i == j) ? a0.l[i] : d0[j], tok);
          fe_mac(acc[1], a1.l[i], (i == j) ? a1.l[i] : d1[j], tok);
        }
      },
      out);
  r0 = t0;
  r1 = t1;
}

// Cheap reduction without a Montgomery product: x (limbs < 2^32, value < 2^(W N)) -> normalised
// limbs, same residue, value < m + m/2^16.  One-limb Barrett: q = floor(top(x) / (top(m)+1)) never
// exceeds floor(x/m) and misses it by at most one; x - q m is formed as the low W N bits of
// x + q (2^(W N) - m), so no borrow chain is needed.  ~4 VALU ops per limb.
template <class P>
PM_DEV Fe<P> fe_reduce_weak(const Fe<P>& x) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  constexpr Limbs<N> RB = Consts<P>::rbar();
  constexpr u64 MAGIC = ((u64)1 << 52) / ((u64)M.v[N - 1] + 1);
  const u32 top = x.l[N - 1] + (x.l[N - 2] >> W);
  const u32 q = (u32)(((u64)top * MAGIC) >> 52);
  Fe<P> r;
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    acc += (u64)q * RB.v[i];
    acc += x.l[i];
    r.l[i] = (u32)acc & MASK;
    acc >>= W;
  }
  return r;
}

// compile-time constant 2^e mod m as an element (canonical limbs)
template <class P, int E>
PM_DEV Fe<P> fe_pow2() {
  constexpr Limbs<P::N> c = Consts<P>::pow2_mod(E);
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::N; ++i) r.l[i] = c.v[i];
  return r;
}
// 1 in the device Montgomery form (R' = 2^(W N)): multiplying by it reduces without changing the value
template <class P>
PM_DEV Fe<P> fe_one() {
  return fe_pow2<P, P::W * P::N>();
}
// x in ABI Montgomery form (R = 2^(32 NS)) <-> device form (R' = 2^(W N)):  x * 2^(+-(W N - 32 NS))
template <class P>
PM_DEV Fe<P> fe_abi_to_dev(const Fe<P>& a) {  // multiply by 2^(WN - 32NS): fe_mul by 2^(2WN - 32NS)
  return fe_mul<P>(a, fe_pow2<P, 2 * P::W * P::N - 32 * P::NS>());
}

// ------------------------------------------------------------------ canonical boundary
// saturated canonical (NS x 32 bit) -> radix 2^W, normalised
template <class P>
PM_DEV Fe<P> fe_unpack(const u32* s) {
  constexpr int N = P::N, W = P::W, NS = P::NS;
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int lo = i * W, j = lo / 32, sh = lo % 32;
    u32 x = 0;
    if (j < NS) x = s[j] >> sh;
    if (j + 1 < NS && sh + W > 32) x |= s[j + 1] << (32 - sh);
    r.l[i] = x & Consts<P>::MASK;
  }
  return r;
}
// fully normalised limbs (value < 2^(32 NS)) -> saturated
template <class P>
PM_DEV void fe_pack_raw(u32* s, const Fe<P>& a) {
  constexpr int N = P::N, W = P::W, NS = P::NS;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    // bits [32j, 32j+32) come from limbs i0 = floor(32j / W) and following
    const int lo = 32 * j;
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l0 = i * W;  // limb i covers [l0, l0 + W) (top limb: beyond)
      if (l0 < lo + 32 && l0 + 32 > lo) {
        if (l0 >= lo)
          x |= a.l[i] << (l0 - lo);
        else
          x |= a.l[i] >> (lo - l0);
      }
    }
    s[j] = x;
  }
}
// value < 2m (limbs < 2^32) -> canonical saturated limbs
template <class P>
PM_DEV void fe_canon_pack(u32* s, const Fe<P>& a) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  Fe<P> x = fe_norm_full<P>(a);
  // d = x - m with a signed borrow chain
  Fe<P> d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    int32_t t = (int32_t)x.l[i] - (int32_t)M.v[i] + c;
    d.l[i] = (u32)t & MASK;
    c = t >> W;  // arithmetic shift: 0 or -1
  }
  int32_t top = (int32_t)x.l[N - 1] - (int32_t)M.v[N - 1] + c;
  d.l[N - 1] = (u32)top;
  const bool neg = top < 0;  // x < m: keep x
#pragma unroll
  for (int i = 0; i < N; ++i) x.l[i] = neg ? x.l[i] : d.l[i];
  fe_pack_raw<P>(s, x);
}

// ------------------------------------------------------------------ 16-byte vector IO
struct __attribute__((aligned(16))) u32x4 {
  u32 x, y, z, w;
};

// canonical element in memory (NS/4 x 16 bytes) -> limbs
template <class P>
PM_DEV Fe<P> fe_load(const void* p) {
  u32 s[P::NS];
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < P::NS / 4; ++i) {
    u32x4 v = q[i];
    s[4 * i] = v.x;
    s[4 * i + 1] = v.y;
    s[4 * i + 2] = v.z;
    s[4 * i + 3] = v.w;
  }
  return fe_unpack<P>(s);
}
// limbs (value < 2m) -> canonical element in memory
template <class P>
PM_DEV void fe_store(void* p, const Fe<P>& a) {
  u32 s[P::NS];
  fe_canon_pack<P>(s, a);
  u32x4* q = reinterpret_cast<u32x4*>(p);
#pragma unroll
  for (int i = 0; i < P::NS / 4; ++i) q[i] = u32x4{s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]};
}

}  // namespace pm
