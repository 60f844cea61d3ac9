// Candidate Montgomery products for the r05 A/B (tools/fe_mul_chain_ab.hip): the library's column-wise product
// (csrc/fields.hip.h, fe_mul) restated so that EVERY limb product of a column accumulates through the C operand
// of v_mad_u64_u32 -- one dependent chain per product, the carry of column k-1 is the chain's first addend.
//
// Why: hipcc reassociates `acc += a*b` so that the late operand (the shifted carry) is added last -- each column
// becomes a fresh chain starting at 0 plus one v_lshl_add_u64 to fold the carry in (fe_mul<FrP>: 24 v_lshl_add_u64,
// 16 for that reason alone).  r01's microbenchmark already showed dependent and independent v_mad_u64_u32 issuing at
// the same rate, so the extra additions buy nothing when other waves (or a second chain) fill the pipe.
// How the chain is forced (no builtin for v_mad_u64_u32 exists in clang 22):
//   * default: plain C++ `acc += (u64)a * b` followed by an input-only empty asm that USES acc (PM_KEEP).  Every partial
//     sum then has two uses, LLVM's Reassociate pass cannot linearise the column's additions, and instruction selection
//     folds each product with the running accumulator into one v_mad_u64_u32.  No instruction is emitted for the asm.
//   * MODE bit 8: the mads themselves as inline asm.  Measured for the record: the hazard recogniser cannot see inside an
//     asm statement and pads every asm -> asm pair with `s_nop 0` (~130 per Fr product).
//
// MODE bits:  1 = split the 64-bit right shift into v_alignbit_b32 + a 32-bit shift
//             2 = Fr only: subtractive quotient digits q' = acc mod 2^W, T - q' m, signed accumulator (v_mad_i64_i32 by -M),
//                 +m folded into the high columns (needs N * Ba * Bb * 2^(2W) < 2^63: Ba * Bb < 3.5 for Fr)
//             4 = "+q" of the additive form through v_mad (q * 1) instead of a 64-bit add of MASK before the shift
//             8 = mads as inline asm instead of C++ + PM_KEEP
#pragma once
#include "fields.hip.h"

namespace pm {

#undef PM_KEEP   // (fields.hip.h has the token-threaded two-argument form of the adopted product)
#define PM_KEEP(x) asm volatile("" ::"v"(x))
template <bool ASM>
PM_DEV void mad_vv(u64& acc, u32 a, u32 b) {
  if constexpr (ASM) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  else { acc += (u64)a * b; PM_KEEP(acc); }
}
template <bool ASM>
PM_DEV void mad_vs(u64& acc, u32 a, u32 s) {
  if constexpr (ASM) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(s) : "vcc");
  else { acc += (u64)a * s; PM_KEEP(acc); }
}
template <bool ASM>
PM_DEV void madi_vs(u64& acc, u32 a, int s) {  // a < 2^31
  if constexpr (ASM) asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(s) : "vcc");
  else { acc += (u64)((long long)(int)a * (long long)s); PM_KEEP(acc); }
}

template <int W, bool SPLIT, bool ARITH>
PM_DEV u64 shr_w(u64 acc) {
  if constexpr (!SPLIT) {
    if constexpr (ARITH) return (u64)((long long)acc >> W);
    return acc >> W;
  } else {
    const u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    const u32 nlo = __builtin_amdgcn_alignbit(hi, lo, W);
    const u32 nhi = ARITH ? (u32)((int)hi >> W) : (hi >> W);
    return ((u64)nhi << 32) | nlo;
  }
}

// K independent products in lock step, every one a single chain
template <class P, int K, int MODE>
PM_DEV void fe_mul_chain(const Fe<P>* a, const Fe<P>* b, Fe<P>* r) {
  constexpr int N = P::N, W = P::W;
  constexpr u32 MASK = Consts<P>::MASK;
  constexpr Limbs<N> M = Consts<P>::mod_limbs();
  constexpr u32 NINV = Consts<P>::neg_inv();
  constexpr bool SPLIT = MODE & 1, SUBT = (MODE & 2) && M.v[0] == 1u, QMAD = MODE & 4, ASM = MODE & 8;
  u32 q[K][N];
  u64 acc[K];
#pragma unroll
  for (int c = 0; c < K; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i)
#pragma unroll
      for (int c = 0; c < K; ++c) mad_vv<ASM>(acc[c], a[c].l[i], b[c].l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i)
#pragma unroll
      for (int c = 0; c < K; ++c) {
        if constexpr (SUBT) madi_vs<ASM>(acc[c], q[c][i], -(int)M.v[k - i]);
        else mad_vs<ASM>(acc[c], q[c][i], M.v[k - i]);
      }
#pragma unroll
    for (int c = 0; c < K; ++c) {
      if constexpr (M.v[0] == 1u) {
        if constexpr (SUBT) {
          q[c][k] = (u32)acc[c] & MASK;  // T - q' m: the low W bits cancel, nothing to add
        } else {
          q[c][k] = (0u - (u32)acc[c]) & MASK;
          if constexpr (QMAD) mad_vs<ASM>(acc[c], q[c][k], 1u);
          else acc[c] += (u64)MASK;  // (acc + q) >> W == (acc + MASK) >> W: the carry does not wait for q
        }
      } else {
        q[c][k] = ((u32)acc[c] * NINV) & MASK;
        mad_vs<ASM>(acc[c], q[c][k], M.v[0]);
      }
      acc[c] = shr_w<W, SPLIT, SUBT>(acc[c]);
    }
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
    if constexpr (SUBT) {  // + m R: limbs (k-N, k-N+1) of m as one 58-bit addend on every second high column
      if ((k - N) % 2 == 0) {
        const u64 two = (u64)M.v[k - N] + ((k - N + 1 < N) ? ((u64)M.v[k - N + 1] << W) : 0ull);
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] += two;
      }
    }
#pragma unroll
    for (int i = k - N + 1; i < N; ++i)
#pragma unroll
      for (int c = 0; c < K; ++c) mad_vv<ASM>(acc[c], a[c].l[i], b[c].l[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; ++i)
#pragma unroll
      for (int c = 0; c < K; ++c) {
        if constexpr (SUBT) madi_vs<ASM>(acc[c], q[c][i], -(int)M.v[k - i]);
        else mad_vs<ASM>(acc[c], q[c][i], M.v[k - i]);
      }
#pragma unroll
    for (int c = 0; c < K; ++c) {
      r[c].l[k - N] = (u32)acc[c] & MASK;
      acc[c] = shr_w<W, SPLIT, SUBT>(acc[c]);
    }
  }
  if constexpr (SUBT && ((N - 1) % 2 == 0)) {
#pragma unroll
    for (int c = 0; c < K; ++c) acc[c] += (u64)M.v[N - 1];
  }
#pragma unroll
  for (int c = 0; c < K; ++c) r[c].l[N - 1] = (u32)acc[c];
}

template <class P, int MODE>
PM_DEV Fe<P> fe_mul_c(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  fe_mul_chain<P, 1, MODE>(&a, &b, &r);
  return r;
}

}  // namespace pm
