// Host-side (x86-64) Montgomery arithmetic for the few scalar jobs the library does on the
// CPU: domain constants (group_gen, size_inv, coset generator powers -- the fields of
// dusk_plonk::fft::EvaluationDomain::new), and the final
// window fold / projective->affine normalisation of an MSM result (a handful of Fp
// operations on 144 bytes, where one CPU thread beats a single GPU lane by >20x).
// 4 x u64 (Fr) / 6 x u64 (Fp) little-endian Montgomery limbs == the Rust memory layout.
#pragma once
#include <stdint.h>
#include <string.h>

namespace pm {
namespace host {

typedef unsigned __int128 u128;
typedef uint64_t u64;

template <int N>
struct Field {
  u64 m[N], one[N], r2[N], inv;
};

inline const Field<4>& FR() {
  static const Field<4> f = {
      {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
      {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL},
      {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL},
      0xfffffffeffffffffULL};
  return f;
}
inline const Field<6>& FP() {
  static const Field<6> f = {
      {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
       0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL},
      {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
       0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL},
      {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
       0x9a793e85b519952dULL, 0x11988fe592cae3aaULL},
      0x89f3fffcfffcfffdULL};
  return f;
}

template <int N>
struct El {
  u64 l[N];
};
typedef El<4> HFr;
typedef El<6> HFp;

template <int N>
inline bool geq(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; --i) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}
template <int N>
inline u64 sub_n(u64* r, const u64* a, const u64* b) {
  u64 bw = 0;
  for (int i = 0; i < N; ++i) {
    u128 t = (u128)a[i] - b[i] - bw;
    r[i] = (u64)t;
    bw = (u64)(t >> 64) & 1;
  }
  return bw;
}
template <int N>
inline u64 add_n(u64* r, const u64* a, const u64* b) {
  u64 c = 0;
  for (int i = 0; i < N; ++i) {
    u128 t = (u128)a[i] + b[i] + c;
    r[i] = (u64)t;
    c = (u64)(t >> 64);
  }
  return c;
}
template <int N>
inline El<N> add(const El<N>& a, const El<N>& b, const Field<N>& F) {
  El<N> r;
  u64 c = add_n<N>(r.l, a.l, b.l);
  if (c || geq<N>(r.l, F.m)) sub_n<N>(r.l, r.l, F.m);
  return r;
}
template <int N>
inline El<N> sub(const El<N>& a, const El<N>& b, const Field<N>& F) {
  El<N> r;
  if (sub_n<N>(r.l, a.l, b.l)) add_n<N>(r.l, r.l, F.m);
  return r;
}
// coarsely integrated operand scanning (CIOS): one pass per limb of b -- multiply-accumulate, then one reduction round;
// N + 2 words of state, every loop bound a template constant (fully unrolled by the compiler)
template <int N>
inline El<N> mul(const El<N>& a, const El<N>& b, const Field<N>& F) {
  u64 t[N + 2] = {0};
#pragma GCC unroll 8
  for (int i = 0; i < N; ++i) {
    u64 c = 0;
    const u64 bi = b.l[i];
#pragma GCC unroll 8
    for (int j = 0; j < N; ++j) {
      u128 p = (u128)a.l[j] * bi + t[j] + c;
      t[j] = (u64)p;
      c = (u64)(p >> 64);
    }
    u128 s = (u128)t[N] + c;
    t[N] = (u64)s;
    t[N + 1] = (u64)(s >> 64);
    const u64 q = t[0] * F.inv;
    u128 p0 = (u128)q * F.m[0] + t[0];
    c = (u64)(p0 >> 64);
#pragma GCC unroll 8
    for (int j = 1; j < N; ++j) {
      u128 p = (u128)q * F.m[j] + t[j] + c;
      t[j - 1] = (u64)p;
      c = (u64)(p >> 64);
    }
    s = (u128)t[N] + c;
    t[N - 1] = (u64)s;
    t[N] = t[N + 1] + (u64)(s >> 64);
  }
  El<N> r;
  memcpy(r.l, t, 8 * N);
  if (t[N] || geq<N>(r.l, F.m)) sub_n<N>(r.l, r.l, F.m);
  return r;
}
template <int N>
inline El<N> one(const Field<N>& F) {
  El<N> r;
  memcpy(r.l, F.one, 8 * N);
  return r;
}
template <int N>
inline El<N> zero() {
  El<N> r;
  memset(r.l, 0, 8 * N);
  return r;
}
template <int N>
inline bool is_zero(const El<N>& a) {
  u64 x = 0;
  for (int i = 0; i < N; ++i) x |= a.l[i];
  return x == 0;
}
template <int N>
inline bool eq(const El<N>& a, const El<N>& b) {
  return memcmp(a.l, b.l, 8 * N) == 0;
}
template <int N>
inline El<N> from_u64(u64 v, const Field<N>& F) {
  El<N> t = zero<N>(), r2;
  t.l[0] = v;
  memcpy(r2.l, F.r2, 8 * N);
  return mul<N>(t, r2, F);
}
// a^e, e = little-endian limbs of a plain integer
template <int N>
inline El<N> pow(const El<N>& a, const u64* e, int elimbs, const Field<N>& F) {
  El<N> acc = one<N>(F), b = a;
  for (int i = 0; i < 64 * elimbs; ++i) {
    if ((e[i / 64] >> (i % 64)) & 1) acc = mul<N>(acc, b, F);
    b = mul<N>(b, b, F);
  }
  return acc;
}
// x / 2 mod m for odd m: (x + m) / 2 when x is odd (the carry of the sum is the top bit)
template <int N>
inline void half_mod(u64* x, const u64* m) {
  u64 top = 0;
  if (x[0] & 1) top = add_n<N>(x, x, m);
  for (int i = 0; i < N - 1; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
  x[N - 1] = (x[N - 1] >> 1) | (top << 63);
}
template <int N>
inline void shr1(u64* x) {
  for (int i = 0; i < N - 1; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
  x[N - 1] >>= 1;
}
template <int N>
inline bool is_one(const u64* x) {
  u64 r = x[0] ^ 1;
  for (int i = 1; i < N; ++i) r |= x[i];
  return r == 0;
}
// 1 / a in the field (Montgomery form in and out; 0 -> 0) by the binary extended Euclid on plain integers (HAC 14.61's
// shape: a few hundred shift / subtract steps on N limbs, ~4 us for Fp) instead of a^(m - 2) (575 products, ~40 us -- it
// was the longest single piece of host work behind every commitment batch of a small proof).  The representative a R is
// inverted as an integer, (a R)^-1 = a^-1 R^-1, and two products by R^2 bring it back: a^-1 R.  Checked against the
// exponentiation in tests/test_abi.py (pm_test_host_inverse).
template <int N>
inline El<N> inv(const El<N>& a, const Field<N>& F) {
  if (is_zero(a)) return a;
  u64 u[N], v[N], x1[N], x2[N];
  memcpy(u, a.l, sizeof u);
  memcpy(v, F.m, sizeof v);
  // a representative >= m (never produced by this library: every El is canonical) is reduced first; a multiple of m is
  // 0 in the field -> 0, like the exponentiation gave.  Without this u reaches 0 below and the halving loop never ends
  // (ADVICE r05).
  while (geq<N>(u, v)) sub_n<N>(u, u, v);
  {
    u64 any = 0;
    for (int i = 0; i < N; ++i) any |= u[i];
    if (!any) return El<N>{};
  }
  memset(x1, 0, sizeof x1);
  memset(x2, 0, sizeof x2);
  x1[0] = 1;
  while (!is_one<N>(u) && !is_one<N>(v)) {
    while (!(u[0] & 1)) {
      shr1<N>(u);
      half_mod<N>(x1, F.m);
    }
    while (!(v[0] & 1)) {
      shr1<N>(v);
      half_mod<N>(x2, F.m);
    }
    if (geq<N>(u, v)) {
      sub_n<N>(u, u, v);
      if (sub_n<N>(x1, x1, x2)) add_n<N>(x1, x1, F.m);
    } else {
      sub_n<N>(v, v, u);
      if (sub_n<N>(x2, x2, x1)) add_n<N>(x2, x2, F.m);
    }
  }
  El<N> y, r2;
  memcpy(y.l, is_one<N>(u) ? x1 : x2, sizeof y.l);
  memcpy(r2.l, F.r2, sizeof r2.l);
  return mul<N>(mul<N>(y, r2, F), r2, F);
}
// the same by exponentiation (kept as the independent check of inv)
template <int N>
inline El<N> inv_fermat(const El<N>& a, const Field<N>& F) {
  u64 e[N], two[N];
  memset(two, 0, sizeof two);
  two[0] = 2;
  sub_n<N>(e, F.m, two);
  return pow<N>(a, e, N, F);
}

// ROOT_OF_UNITY = 7^((r-1)/2^32), Montgomery (order 2^32) -- SURVEY.md section 8c
inline HFr fr_root_of_unity() {
  return HFr{{0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL,
              0x5bf3adda19e9b27bULL}};
}
static const unsigned FR_TWO_ADICITY = 32;
static const u64 FR_GENERATOR = 7;  // multiplicative generator = coset shift

// ------------------------------------------------------------------ G1 on the host
// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2); identity <=> ZZ == 0.
struct XYZZ {
  HFp x, y, zz, zzz;
};
inline XYZZ xyzz_identity() {
  XYZZ r;
  r.x = zero<6>(); r.y = zero<6>(); r.zz = zero<6>(); r.zzz = zero<6>();
  return r;
}
inline XYZZ xyzz_double(const XYZZ& p) {
  const Field<6>& F = FP();
  if (is_zero(p.zz)) return p;
  // dbl-2008-s-1
  HFp U = add(p.y, p.y, F);
  if (is_zero(U)) return xyzz_identity();  // y == 0 cannot occur in G1 (odd order); defensive
  HFp V = mul(U, U, F), W = mul(U, V, F), S = mul(p.x, V, F);
  HFp xx = mul(p.x, p.x, F);
  HFp M = add(add(xx, xx, F), xx, F);
  XYZZ r;
  r.x = sub(sub(mul(M, M, F), S, F), S, F);
  r.y = sub(mul(M, sub(S, r.x, F), F), mul(W, p.y, F), F);
  r.zz = mul(V, p.zz, F);
  r.zzz = mul(W, p.zzz, F);
  return r;
}
inline XYZZ xyzz_add(const XYZZ& p, const XYZZ& q) {
  const Field<6>& F = FP();
  if (is_zero(p.zz)) return q;
  if (is_zero(q.zz)) return p;
  // add-2008-s
  HFp U1 = mul(p.x, q.zz, F), U2 = mul(q.x, p.zz, F);
  HFp S1 = mul(p.y, q.zzz, F), S2 = mul(q.y, p.zzz, F);
  HFp P = sub(U2, U1, F), R = sub(S2, S1, F);
  if (is_zero(P)) {
    if (is_zero(R)) return xyzz_double(p);
    return xyzz_identity();
  }
  HFp PP = mul(P, P, F), PPP = mul(P, PP, F), Q = mul(U1, PP, F);
  XYZZ r;
  r.x = sub(sub(sub(mul(R, R, F), PPP, F), Q, F), Q, F);
  r.y = sub(mul(R, sub(Q, r.x, F), F), mul(S1, PPP, F), F);
  r.zz = mul(mul(p.zz, q.zz, F), PP, F);
  r.zzz = mul(mul(p.zzz, q.zzz, F), PPP, F);
  return r;
}
// -> affine (x, y); identity -> returns false and leaves out untouched
inline bool xyzz_to_affine(const XYZZ& p, HFp& x, HFp& y) {
  const Field<6>& F = FP();
  if (is_zero(p.zz)) return false;
  HFp zi = inv(p.zzz, F);             // 1/ZZZ
  HFp zz_i = mul(mul(zi, zi, F), mul(p.zz, p.zz, F), F);  // ZZ^2/ZZZ^2 = ZZ^2/ZZ^3 = 1/ZZ
  x = mul(p.x, zz_i, F);
  y = mul(p.y, zi, F);
  return true;
}

}  // namespace host
}  // namespace pm
