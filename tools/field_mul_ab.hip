// A/B of two ways to form a BLS12-381 Montgomery product on gfx950:
//   A  the library's integer product: unsaturated 29/28-bit limbs, one v_mad_u64_u32 per limb
//      product with a 64-bit column accumulator (csrc/fields.hip.h, fe_mul / fe_sqr)
//   B  a double-precision-FMA product on 52-bit limbs (Emmart/Zheng/Weems style): each limb
//      product is split exactly into hi = fma_rz(a, b, 2^104) and lo = fma_rz(a, b, 2^104 + 2^52 - hi),
//      and the mantissa bit patterns are summed into 64-bit integer columns
// Both run as dependent chains x <- x*y, y <- y*x (two independent chains per thread), so the
// number measured is products/s as a kernel would see them.  The FMA product is checked against
// __int128 arithmetic on the host for the first lanes (exit code 2 on a mismatch).
// Build: hipcc --offload-arch=gfx950 -O3 -I../plonk-prototype_amd/csrc field_mul_ab.hip -o field_mul_ab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "fields.hip.h"

using namespace pm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------ B: FMA product
static constexpr u64 M52 = (1ull << 52) - 1;
template <class P>
struct DpfConsts {
  static constexpr int N = (32 * P::NS + 51 + 4) / 52;  // Fr: 5 limbs (R = 2^260), Fp: 8 (R = 2^416)
  static PM_HD u64 mod_limb(int i) {
    u64 lo = sat_bits<P::NS>(P::SAT, 52 * i, 26), hi = sat_bits<P::NS>(P::SAT, 52 * i + 26, 26);
    return lo | (hi << 26);
  }
  static PM_HD u64 neg_inv() {  // -m^-1 mod 2^52
    u64 m0 = mod_limb(0), x = 1;
    for (int i = 0; i < 7; ++i) x = x * (2ull - m0 * x);
    return (0ull - x) & M52;
  }
};
template <int N>
struct Dpf {
  double l[N];
};

PM_DEV u64 dbits(double d) { return (u64)__double_as_longlong(d); }
PM_DEV double bitsd(u64 b) { return __longlong_as_double((long long)b); }

// The wave must run with f64 rounding = toward zero (set_rz()).
template <class P>
PM_DEV Dpf<DpfConsts<P>::N> dpf_mul(const Dpf<DpfConsts<P>::N>& a, const Dpf<DpfConsts<P>::N>& b) {
  constexpr int N = DpfConsts<P>::N;
  constexpr u64 EXP_HI = 0x467ull << 52, EXP_LO = 0x433ull << 52;  // 2^104, 2^52 exponent fields
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  u64 col[2 * N + 1];
  // every column starts at minus the exponent fields it is going to receive (product + reduction)
#pragma unroll
  for (int k = 0; k <= 2 * N; ++k) {
    int nlo = 0, nhi = 0;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        if (i + j == k) nlo += 2;       // a_i b_j and q_i m_j
        if (i + j + 1 == k) nhi += 2;
      }
    col[k] = 0ull - ((u64)nlo * EXP_LO + (u64)nhi * EXP_HI);
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double hi = __builtin_fma(a.l[i], b.l[j], C1);
      const double lo = __builtin_fma(a.l[i], b.l[j], C2 - hi);
      col[i + j + 1] += dbits(hi);
      col[i + j] += dbits(lo);
    }
  constexpr u64 NINV = DpfConsts<P>::neg_inv();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 q;
    if (NINV == ((0ull - (1ull + (1ull << 32))) & M52))  // Fr: m = 1 - 2^32 mod 2^52
      q = (0ull - (col[i] + (col[i] << 32))) & M52;
    else
      q = (col[i] * NINV) & M52;
    const double qd = bitsd(q | EXP_LO) - 0x1p52;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double mj = (double)DpfConsts<P>::mod_limb(j);
      const double hi = __builtin_fma(qd, mj, C1);
      const double lo = __builtin_fma(qd, mj, C2 - hi);
      col[i + j + 1] += dbits(hi);
      col[i + j] += dbits(lo);
    }
    col[i + 1] += col[i] >> 52;
  }
  Dpf<N> r;
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const u64 t = col[N + k] + carry;
    carry = (k == N - 1) ? 0 : (t >> 52);
    const u64 limb = (k == N - 1) ? t : (t & M52);
    r.l[k] = bitsd(limb | EXP_LO) - 0x1p52;
  }
  return r;
}

PM_DEV void set_rz() {
  // MODE register, FP_ROUND for f64/f16 = bits [3:2]; 3 = round toward zero.  Inline asm on purpose:
  // after the __builtin_amdgcn_s_setreg intrinsic LLVM's mode-register pass re-asserts the default
  // rounding mode in front of the first f64 instruction.
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}

// ------------------------------------------------------------------ kernels
#define ITER 256
template <class P, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_int_mul(u32* out, u32 seed) {
  const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  Fe<P> x, y, u, v;
#pragma unroll
  for (int i = 0; i < P::N; ++i) {
    x.l[i] = (tid * 2654435761u + seed + i) & Consts<P>::MASK;
    y.l[i] = (tid * 40503u + 7 * i + seed) & Consts<P>::MASK;
    u.l[i] = x.l[i] ^ 0x5555u;
    v.l[i] = y.l[i] ^ 0x3333u;
  }
  x.l[P::N - 1] &= 0xffff; y.l[P::N - 1] &= 0xffff; u.l[P::N - 1] &= 0xffff; v.l[P::N - 1] &= 0xffff;
  for (int it = 0; it < ITER; ++it) {
    x = fe_mul<P>(x, y);
    u = fe_mul<P>(u, v);
    y = fe_mul<P>(y, x);
    v = fe_mul<P>(v, u);
  }
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::N; ++i) o ^= x.l[i] ^ y.l[i] ^ u.l[i] ^ v.l[i];
  out[tid] = o;
}
template <class P, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_int_sqr(u32* out, u32 seed) {
  const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  Fe<P> x, u;
#pragma unroll
  for (int i = 0; i < P::N; ++i) {
    x.l[i] = (tid * 2654435761u + seed + i) & Consts<P>::MASK;
    u.l[i] = x.l[i] ^ 0x5555u;
  }
  x.l[P::N - 1] &= 0xffff; u.l[P::N - 1] &= 0xffff;
  for (int it = 0; it < ITER; ++it) {
    x = fe_sqr<P>(x); u = fe_sqr<P>(u);
    x = fe_sqr<P>(x); u = fe_sqr<P>(u);
  }
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::N; ++i) o ^= x.l[i] ^ u.l[i];
  out[tid] = o;
}
template <class P, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_dpf_mul(u32* out, u32 seed, u64* dump, int iters) {
  constexpr int N = DpfConsts<P>::N;
  set_rz();
  const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  Dpf<N> x, y, u, v;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 a = ((u64)(tid * 2654435761u + seed + i) << 20 | (tid * 977u + i)) & M52;
    u64 b = ((u64)(tid * 40503u + 7 * i + seed) << 21 | (tid * 31u + 3 * i)) & M52;
    if (i == N - 1) { a &= 0xffff; b &= 0xffff; }
    x.l[i] = (double)a; y.l[i] = (double)b;
    u.l[i] = (double)(a ^ 0x5555u); v.l[i] = (double)(b ^ 0x3333u);
  }
  if (dump && tid < 64) {  // inputs of the first product, for the host check
#pragma unroll
    for (int i = 0; i < N; ++i) { dump[tid * 3 * N + i] = (u64)x.l[i]; dump[tid * 3 * N + N + i] = (u64)y.l[i]; }
  }
  for (int it = 0; it < iters; ++it) {
    x = dpf_mul<P>(x, y);
    if (dump && it == 0 && tid < 64) {
#pragma unroll
      for (int i = 0; i < N; ++i) dump[tid * 3 * N + 2 * N + i] = (u64)x.l[i];
    }
    u = dpf_mul<P>(u, v);
    y = dpf_mul<P>(y, x);
    v = dpf_mul<P>(v, u);
  }
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) o ^= (u32)dbits(x.l[i]) ^ (u32)dbits(y.l[i]) ^ (u32)dbits(u.l[i]) ^ (u32)dbits(v.l[i]);
  out[tid] = o;
}

// ------------------------------------------------------------------ host check of the FMA product
typedef unsigned __int128 u128;
struct Big { std::vector<u64> w; };  // little-endian 64-bit words
static Big big_from_limbs52(const u64* l, int n, int words) {
  Big r; r.w.assign(words, 0);
  for (int i = 0; i < n; ++i) {
    int lo = 52 * i, j = lo / 64, sh = lo % 64;
    u128 v = (u128)l[i] << sh;
    u128 c = v;
    for (int k = j; k < words && c; ++k) { u128 t = (u128)r.w[k] + (u64)c; r.w[k] = (u64)t; c = (c >> 64) + (t >> 64); }
  }
  return r;
}
static Big big_mul(const Big& a, const Big& b) {
  Big r; r.w.assign(a.w.size() + b.w.size(), 0);
  for (size_t i = 0; i < a.w.size(); ++i) {
    u64 c = 0;
    for (size_t j = 0; j < b.w.size(); ++j) { u128 t = (u128)a.w[i] * b.w[j] + r.w[i + j] + c; r.w[i + j] = (u64)t; c = (u64)(t >> 64); }
    r.w[i + b.w.size()] += c;
  }
  return r;
}
static int big_cmp(const Big& a, const Big& b) {
  size_t n = a.w.size() > b.w.size() ? a.w.size() : b.w.size();
  for (size_t i = n; i-- > 0;) {
    u64 x = i < a.w.size() ? a.w[i] : 0, y = i < b.w.size() ? b.w[i] : 0;
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}
static void big_sub(Big& a, const Big& b) {
  u64 bw = 0;
  for (size_t i = 0; i < a.w.size(); ++i) { u64 y = i < b.w.size() ? b.w[i] : 0; u128 t = (u128)a.w[i] - y - bw; a.w[i] = (u64)t; bw = (u64)(t >> 64) & 1; }
}
// r * 2^(52 N) == a * b (mod m), checked as r * 2^(52N) - a*b == 0 mod m by long reduction
static Big big_mod(Big x, const Big& m) {  // shift-subtract, slow but tiny
  int bits = (int)x.w.size() * 64;
  Big r; r.w.assign(m.w.size() + 1, 0);
  for (int i = bits - 1; i >= 0; --i) {
    u64 c = (x.w[i / 64] >> (i % 64)) & 1;
    for (size_t k = 0; k < r.w.size(); ++k) { u64 n = r.w[k] >> 63; r.w[k] = (r.w[k] << 1) | c; c = n; }
    if (big_cmp(r, m) >= 0) big_sub(r, m);
  }
  return r;
}
template <class P>
static int check_dpf(const std::vector<u64>& dump) {
  constexpr int N = DpfConsts<P>::N;
  u64 ml[N];
  for (int i = 0; i < N; ++i) ml[i] = DpfConsts<P>::mod_limb(i);
  const int words = (52 * N + 63) / 64 + 1;
  Big m = big_from_limbs52(ml, N, words);
  for (int t = 0; t < 64; ++t) {
    Big a = big_from_limbs52(&dump[t * 3 * N], N, words), b = big_from_limbs52(&dump[t * 3 * N + N], N, words);
    Big r = big_from_limbs52(&dump[t * 3 * N + 2 * N], N, words);
    Big R; R.w.assign(words, 0); R.w[(52 * N) / 64] = 1ull << ((52 * N) % 64);
    Big lhs = big_mod(big_mul(r, R), m), rhs = big_mod(big_mul(a, b), m);
    if (big_cmp(lhs, rhs) != 0) { printf("FMA product MISMATCH at lane %d\n", t); return 2; }
    Big two; two.w = {2};
    Big twice = big_mul(m, two);
    if (big_cmp(r, twice) >= 0) { printf("FMA product not below 2m at lane %d\n", t); return 2; }
  }
  return 0;
}

// ------------------------------------------------------------------ driver
template <typename F>
static float time_launch(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(0); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0); launch(r + 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
static void report(const char* name, int waves, int cus, float ms, double products_per_thread, double ref) {
  const double total = products_per_thread * 256.0 * cus * waves;
  const double rate = total / (ms * 1e-3);
  // cycles per wave-product per SIMD at 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / (products_per_thread * waves);
  printf("%-22s waves/SIMD=%d  %8.3f ms  %.3e products/s  (%7.1f cycles per wave-product per SIMD)%s", name, waves, ms, rate, cyc,
         ref > 0 ? "" : "\n");
  if (ref > 0) printf("  x%.2f vs integer\n", rate / ref);
}

template <int W>
static int run_all(int cus, u32* d_out, u64* d_dump) {
  const int blocks = cus * W;
  const double ppt = 4.0 * ITER;
  float t;
  t = time_launch([&](int s) { k_int_mul<FrP, W><<<blocks, 256>>>(d_out, (u32)s); });
  const double fr_int = ppt * 256.0 * cus * W / (t * 1e-3);
  report("Fr  int 9x29 fe_mul", W, cus, t, ppt, 0);
  t = time_launch([&](int s) { k_int_sqr<FrP, W><<<blocks, 256>>>(d_out, (u32)s); });
  report("Fr  int 9x29 fe_sqr", W, cus, t, ppt, 0);
  t = time_launch([&](int s) { k_dpf_mul<FrP, W><<<blocks, 256>>>(d_out, (u32)s, nullptr, ITER); });
  report("Fr  f64 5x52 fma", W, cus, t, ppt, fr_int);
  t = time_launch([&](int s) { k_int_mul<FpP, W><<<blocks, 256>>>(d_out, (u32)s); });
  const double fp_int = ppt * 256.0 * cus * W / (t * 1e-3);
  report("Fp  int 14x28 fe_mul", W, cus, t, ppt, 0);
  t = time_launch([&](int s) { k_int_sqr<FpP, W><<<blocks, 256>>>(d_out, (u32)s); });
  report("Fp  int 14x28 fe_sqr", W, cus, t, ppt, 0);
  t = time_launch([&](int s) { k_dpf_mul<FpP, W><<<blocks, 256>>>(d_out, (u32)s, nullptr, ITER); });
  report("Fp  f64 8x52 fma", W, cus, t, ppt, fp_int);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s CUs=%d\n", p.gcnArchName, cus);
  u32* d_out;
  u64* d_dump;
  CK(hipMalloc(&d_out, 256 * cus * 8 * sizeof(u32)));
  CK(hipMalloc(&d_dump, 64 * 3 * 8 * sizeof(u64)));
  // correctness of the FMA product first
  {
    std::vector<u64> h(64 * 3 * 8);
    k_dpf_mul<FrP, 1><<<1, 256>>>(d_out, 1, d_dump, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_dump, 64 * 3 * 5 * 8, hipMemcpyDeviceToHost));
    if (int rc = check_dpf<FrP>(h)) return rc;
    k_dpf_mul<FpP, 1><<<1, 256>>>(d_out, 1, d_dump, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_dump, 64 * 3 * 8 * 8, hipMemcpyDeviceToHost));
    if (int rc = check_dpf<FpP>(h)) return rc;
    printf("FMA products check against 128-bit host arithmetic: ok (64 lanes each, Fr and Fp)\n");
  }
  run_all<1>(cus, d_out, d_dump);
  run_all<2>(cus, d_out, d_dump);
  run_all<4>(cus, d_out, d_dump);
  return 0;
}
