// r05 A/B: the library's Montgomery product (compiler-scheduled C++, csrc/fields.hip.h) against
// single-chain restatements whose limb products accumulate through the C operand of v_mad_u64_u32 (tools/fe_chain.hip.h),
// with the 64-bit shift kept or split into 32-bit forms, and for Fr the subtractive quotient digit.
// Every variant is first compared with fe_mul on random operands (canonical bytes; exit code 2 on a mismatch), then timed
// as dependent chains x <- x*y, y <- y*x with in-kernel s_memtime stamps (true shader cycles, waves placed per SIMD as in
// tools/ubench.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -I../plonk-prototype_amd/csrc fe_mul_chain_ab.hip -o fe_mul_chain_ab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "fe_chain.hip.h"

using namespace pm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// VAR: -1 = library fe_mul / fe_mul2; otherwise MODE bits of fe_mul_chain
template <class P, int VAR, int K>
PM_DEV void prod(const Fe<P>* a, const Fe<P>* b, Fe<P>* r) {
  if constexpr (VAR < 0) {
    if constexpr (K == 1) r[0] = fe_mul<P>(a[0], b[0]);
    else fe_mul2<P>(a[0], b[0], a[1], b[1], r[0], r[1]);
  } else {
    fe_mul_chain<P, K, VAR>(a, b, r);
  }
}

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// ------------------------------------------------------------------ correctness
template <class P, int VAR, int K>
__global__ void k_check(u32* bad, u32 seed, u32 a_bound) {
  const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  Fe<P> a[K], b[K], r[K];
  for (int c = 0; c < K; ++c)
    for (int i = 0; i < P::N; ++i) {
      // a: limbs up to a_bound * 2^W (the lazily reduced left operand), b: normalised
      const u32 ra = mix(tid * 977u + seed + 31u * i + 7919u * c), rb = mix(tid * 1301u + seed * 3u + 17u * i + 104729u * c);
      a[c].l[i] = (u32)(((u64)ra * ((u64)a_bound << P::W)) >> 32);
      b[c].l[i] = rb & Consts<P>::MASK;
      if ((tid & 15) == 1) a[c].l[i] = (a_bound << P::W) - 1;  // extremes
      if ((tid & 15) == 2) b[c].l[i] = Consts<P>::MASK;
      if ((tid & 15) == 3) a[c].l[i] = 0;
      if ((tid & 15) == 4) { a[c].l[i] = (a_bound << P::W) - 1; b[c].l[i] = Consts<P>::MASK; }
    }
  // keep the VALUE below the bound fe_mul needs (a*b/R + m < 2m): top limbs small
  for (int c = 0; c < K; ++c) { a[c].l[P::N - 1] &= 0xffff; b[c].l[P::N - 1] &= 0xffff; }
  prod<P, VAR, K>(a, b, r);
  for (int c = 0; c < K; ++c) {
    const Fe<P> e = fe_mul<P>(a[c], b[c]);
    u32 s0[P::NS], s1[P::NS];
    fe_canon_pack<P>(s0, e);
    fe_canon_pack<P>(s1, r[c]);
    bool ok = true;
    for (int i = 0; i < P::NS; ++i) ok &= s0[i] == s1[i];
    for (int i = 0; i < P::N - 1; ++i) ok &= r[c].l[i] <= Consts<P>::MASK;  // normalised output limbs
    if (!ok) atomicAdd(bad, 1u);
  }
}

// ------------------------------------------------------------------ timing
#define ITER 128
template <class P, int VAR, int K, int WPB>
__global__ void __launch_bounds__(256 * WPB) k_time(u32* out, unsigned long long* stamps, u32 seed) {
  extern __shared__ u32 lds_dummy[];
  const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  Fe<P> x[2], y[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
      x[c].l[i] = mix(tid * 2654435761u + seed + i + 100 * c) & Consts<P>::MASK;
      y[c].l[i] = mix(tid * 40503u + 7 * i + seed + 1000 * c) & Consts<P>::MASK;
    }
  x[0].l[P::N - 1] &= 0xffff; y[0].l[P::N - 1] &= 0xffff; x[1].l[P::N - 1] &= 0xffff; y[1].l[P::N - 1] &= 0xffff;
  if (seed == 0xffffffffu) lds_dummy[threadIdx.x] = tid;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  for (int it = 0; it < ITER; ++it) {  // 4 products per iteration: two dependent chains
    if constexpr (K == 1) {
      prod<P, VAR, 1>(&x[0], &y[0], &x[0]);
      prod<P, VAR, 1>(&x[1], &y[1], &x[1]);
      prod<P, VAR, 1>(&y[0], &x[0], &y[0]);
      prod<P, VAR, 1>(&y[1], &x[1], &y[1]);
    } else {
      Fe<P> t[2];
      prod<P, VAR, 2>(x, y, t);
      x[0] = t[0]; x[1] = t[1];
      prod<P, VAR, 2>(y, x, t);
      y[0] = t[0]; y[1] = t[1];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::N; ++i) o ^= x[0].l[i] ^ y[0].l[i] ^ x[1].l[i] ^ y[1].l[i];
  out[tid] = o;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * (tid >> 6)] = t1 - t0;
    stamps[2 * (tid >> 6) + 1] = r1 - r0;
  }
}

static int g_cus = 0;
static u32* g_out = nullptr;
static unsigned long long* g_st = nullptr;
static std::vector<unsigned long long> g_h;

template <class P, int VAR, int K, int WPB>
int time_one(const char* field, const char* name, double* cyc_out) {
  const int blocks = g_cus, threads = 256 * WPB;
  auto kern = &k_time<P, VAR, K, WPB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  double best = 1e30, clk = 0;
  for (int r = 0; r < 4; ++r) {
    kern<<<blocks, threads, 100 * 1024>>>(g_out, g_st, (u32)r + 1);
    CK(hipDeviceSynchronize());
    const int nw = blocks * threads / 64;
    CK(hipMemcpy(g_h.data(), g_st, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0;
    for (int w = 0; w < nw; ++w) { sc += (double)g_h[2 * w]; sr += (double)g_h[2 * w + 1]; }
    const double cyc = sc / nw / ((double)ITER * 4 * WPB);
    if (r > 0 && cyc < best) { best = cyc; clk = (sc / nw) / ((sr / nw) * 10e-9) / 1e9; }
  }
  printf("%-3s %-44s K=%d waves/SIMD=%d  %7.1f cycles per wave-product per SIMD  (%3d VGPRs, %u B scratch, clock %.2f GHz)\n", field, name, K, WPB, best,
         fa.numRegs, (unsigned)fa.localSizeBytes, clk);
  if (cyc_out) *cyc_out = best;
  return 0;
}

template <class P, int VAR, int K>
int check_one(const char* field, const char* name, u32 a_bound, u32* d_bad) {
  CK(hipMemset(d_bad, 0, 4));
  k_check<P, VAR, K><<<256, 256>>>(d_bad, 12345u, a_bound);
  k_check<P, VAR, K><<<256, 256>>>(d_bad, 999u, 1u);
  CK(hipDeviceSynchronize());
  u32 bad; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("check %-3s %-44s K=%d a-limbs < %u * 2^W: %s (%u mismatches of %d)\n", field, name, K, a_bound, bad ? "MISMATCH" : "ok", bad, 2 * 65536 * K);
  return bad ? 2 : 0;
}

#define VARIANT(P, F, VAR, NAME, ABOUND)                                   \
  if (int e = check_one<P, VAR, 1>(F, NAME, ABOUND, d_bad)) return e;      \
  if (int e = check_one<P, VAR, 2>(F, NAME, ABOUND, d_bad)) return e;
#define TIME(P, F, VAR, NAME)                                              \
  if (time_one<P, VAR, 1, 1>(F, NAME, nullptr)) return 1;                  \
  if (time_one<P, VAR, 2, 1>(F, NAME, nullptr)) return 1;                  \
  if (time_one<P, VAR, 1, 2>(F, NAME, nullptr)) return 1;                  \
  if (time_one<P, VAR, 2, 2>(F, NAME, nullptr)) return 1;                  \
  if (time_one<P, VAR, 1, 4>(F, NAME, nullptr)) return 1;

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  g_cus = p.multiProcessorCount;
  printf("device %s CUs=%d\n", p.gcnArchName, g_cus);
  CK(hipMalloc(&g_out, sizeof(u32) * 1024 * g_cus));
  CK(hipMalloc(&g_st, sizeof(unsigned long long) * 2 * 16 * g_cus));
  g_h.resize(2 * 16 * g_cus);
  u32* d_bad; CK(hipMalloc(&d_bad, 4));
  // a-limb bounds: the library allows the left operand up to 6 * 2^29 (Fr) / 13 * 2^28 (Fp); the signed form 3 * 2^29
  VARIANT(FrP, "Fr", 0, "chain, add MASK, shr64", 5)
  VARIANT(FrP, "Fr", 1, "chain, add MASK, alignbit+shr32", 5)
  VARIANT(FrP, "Fr", 4, "chain, +q by mad, shr64", 5)
  VARIANT(FrP, "Fr", 5, "chain, +q by mad, alignbit+shr32", 5)
  VARIANT(FrP, "Fr", 2, "chain, subtractive signed, ashr64", 3)
  VARIANT(FrP, "Fr", 3, "chain, subtractive signed, alignbit+ashr32", 3)
  VARIANT(FrP, "Fr", 8, "asm-mad chain, add MASK, shr64", 5)
  VARIANT(FpP, "Fp", 0, "chain, shr64", 12)
  VARIANT(FpP, "Fp", 1, "chain, alignbit+shr32", 12)
  TIME(FrP, "Fr", -1, "library fe_mul / fe_mul2")
  TIME(FrP, "Fr", 0, "chain, add MASK, shr64")
  TIME(FrP, "Fr", 1, "chain, add MASK, alignbit+shr32")
  TIME(FrP, "Fr", 4, "chain, +q by mad, shr64")
  TIME(FrP, "Fr", 5, "chain, +q by mad, alignbit+shr32")
  TIME(FrP, "Fr", 2, "chain, subtractive signed, ashr64")
  TIME(FrP, "Fr", 3, "chain, subtractive signed, alignbit+ashr32")
  TIME(FrP, "Fr", 8, "asm-mad chain, add MASK, shr64")
  TIME(FpP, "Fp", -1, "library fe_mul / fe_mul2")
  TIME(FpP, "Fp", 0, "chain, shr64")
  TIME(FpP, "Fp", 1, "chain, alignbit+shr32")
  return 0;
}
