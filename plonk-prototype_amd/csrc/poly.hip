// Device-resident polynomial helpers around the NTT / MSM calls of a PLONK prover round
// (SURVEY.md section 8f rows N1 / N2 -- the callers either side of the hot path):
//   pm_fr_vec_op_dev         Evaluations / Polynomial coefficient-wise  + - *   (dusk_plonk::fft)
//   pm_fr_poly_evaluate_dev  Polynomial::evaluate          (Horner at a point)
//   pm_fr_poly_ruffini_dev   Polynomial::ruffini           (division by X - z)
//   pm_fr_batch_inverse_dev  util::batch_inversion         (zeros stay zero)
// dusk-plonk 0.8.2 is pinned at ref:Cargo.toml:19; none of it is in the reference tree.
// All vectors are canonical ABI Fr (4 x u64 Montgomery, R = 2^256) in device memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "context.h"
#include "host_field.h"
#include "ntt_kernels.cuh"
#include "poly_common.cuh"

namespace pm {

using host::HFr;

// ------------------------------------------------------------------ coefficient-wise ops
template <int OP>
__global__ void __launch_bounds__(256) vec_op_kernel(const u32x4* a, const u32x4* b, size_t b_len, u32x4* out,
                                                      size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  Fr bb = fe_zero<FrP>();
  if (b_len == 1) bb = ld_canon(b, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    Fr x = ld_canon(a, i);
    Fr y = b_len == 1 ? bb : ld_canon(b, i);
    Fr r;
    if (OP == 0) r = fe_add<FrP>(x, y);                                   // (2, <2)
    if (OP == 1) r = fe_reduce_weak<FrP>(fe_sub<FrP, 2, 1>(x, y));         // x - y + 2r -> (1, <1.01)
    if (OP == 2) r = fr_abi_mul(x, y);
    st_canon(out, i, r);
  }
}

// ------------------------------------------------------------------ evaluate
struct EvalConsts {
  u32 x[9];      // the point, device Montgomery form (x * 2^261)
  u32 xrow[9];   // x^256
  u32 xseg[9];   // x^(256 L)
  u32 one[9];
};
// sum of 256 lazily reduced values through LDS; result in thread 0
PM_DEV Fr block_sum_256(Fr v, u32* sh /* 256 * 9 words */) {
  const u32 t = threadIdx.x;
  for (u32 s = 128; s > 0; s >>= 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[i * 256 + t] = v.l[i];
    __syncthreads();
    if (t < s) {
      Fr o;
#pragma unroll
      for (int i = 0; i < 9; ++i) o.l[i] = sh[i * 256 + t + s];
      v = fe_reduce_weak<FrP>(fe_add<FrP>(v, o));
    }
    __syncthreads();
  }
  return v;
}
// partial[b] = x^(b SEG) * sum_{i in segment b} c_i x^(i - b SEG),  SEG = 256 L, strided Horner
struct EvalPolys {
  const u32x4* p[PM_LINCOMB_MAX];   // blockIdx.y selects the polynomial
};
__global__ void __launch_bounds__(256) poly_eval_kernel(const EvalPolys polys, size_t n, u32 L, const EvalConsts kc,
                                                         const u32x4* xpow /* x^t, t < 256 */,
                                                         const u32x4* xblk /* x^(b SEG) */, u32x4* partial_all) {
  __shared__ u32 sh[256 * 9];
  const u32 t = threadIdx.x, b = blockIdx.x;
  const u32x4* coeffs = polys.p[blockIdx.y];
  u32x4* partial = partial_all + 3 * (size_t)blockIdx.y * gridDim.x;
  const size_t base = (size_t)b * 256 * L;
  const Fr xrow = fr_limbs(kc.xrow);
  Fr acc = fe_zero<FrP>();
  for (u32 j = L; j-- > 0;) {
    const size_t idx = base + (size_t)j * 256 + t;
    acc = fe_mul<FrP>(acc, xrow);                       // (1, <2)
    if (idx < n) acc = fe_add<FrP>(acc, ld_canon(coeffs, idx));   // (2, <3)
  }
  acc = fe_mul<FrP>(acc, ld_tw(xpow, t));
  acc = block_sum_256(acc, sh);
  if (t == 0) {
    acc = fe_mul<FrP>(acc, ld_tw(xblk, b));
    st_tw(partial, b, acc);
  }
}
__global__ void __launch_bounds__(256) poly_eval_final_kernel(const u32x4* partial_all, u32 count, u32x4* out) {
  __shared__ u32 sh[256 * 9];
  const u32 t = threadIdx.x;
  const u32x4* partial = partial_all + 3 * (size_t)blockIdx.x * count;
  Fr acc = fe_zero<FrP>();
  for (u32 i = t; i < count; i += 256) acc = fe_reduce_weak<FrP>(fe_add<FrP>(acc, ld_tw(partial, i)));
  acc = block_sum_256(acc, sh);
  if (t == 0) st_canon(out, blockIdx.x, acc);
}

// ------------------------------------------------------------------ Ruffini
// q_{i-1} = c_i + z q_i.  With k = n-1-i, d_k = c_{n-1-k}: y_k = d_k + z y_{k-1} = z^k sum_{s<=k} d_s z^-s,
// so the recurrence becomes an element-wise scaling, a prefix SUM, and a scaling back.
struct ScanArgs {
  const u32x4* coeffs;   // canonical, n
  u32x4* tmp;            // canonical, n-1: block-local inclusive prefix sums of d_s z^-s
  u32x4* block_tot;      // 48-byte entries, one per block
  const u32x4* zi_hi;    // z^-(x << lh), 48-byte entries
  const u32x4* zi_lo;    // z^-x
  const u32x4* z_hi;     // z^(x << lh)
  const u32x4* z_lo;
  u32x4* out;            // canonical, n-1
  size_t n;              // coefficients
  u32 lh;
  u32 L;                 // tiles of 256 per block
};
PM_DEV Fr fr_shfl_up(const Fr& v, int d) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = __shfl_up(v.l[i], d);
  return r;
}
// inclusive prefix sum across the 256 threads of a block (values (1, <1.01) in, same out)
PM_DEV Fr block_scan_256(Fr v, u32* sh /* 4 * 9 words */) {
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  for (int d = 1; d < 64; d <<= 1) {
    Fr o = fr_shfl_up(v, d);
    if (lane >= (u32)d) v = fe_reduce_weak<FrP>(fe_add<FrP>(v, o));
  }
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = v.l[i];
  }
  __syncthreads();
  Fr carry = fe_zero<FrP>();
  for (u32 w = 0; w < wave; ++w) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[w * 9 + i];
    carry = fe_add<FrP>(carry, o);                       // <= (3, <3.1)
  }
  v = fe_reduce_weak<FrP>(fe_add<FrP>(v, carry));
  __syncthreads();
  return v;
}
__global__ void __launch_bounds__(256) ruffini_local_kernel(const ScanArgs a) {
  __shared__ u32 sh[4 * 9 + 9];
  const u32 t = threadIdx.x, b = blockIdx.x;
  const size_t m = a.n - 1;  // outputs
  Fr run = fe_zero<FrP>();   // sum of the earlier tiles of this block
  for (u32 j = 0; j < a.L; ++j) {
    const size_t k = ((size_t)b * a.L + j) * 256 + t;
    Fr e = fe_zero<FrP>();
    if (k < m) e = fe_mul<FrP>(ld_canon(a.coeffs, a.n - 1 - k), two_level(a.zi_hi, a.zi_lo, (u32)k, a.lh));
    e = fe_reduce_weak<FrP>(e);
    Fr s = block_scan_256(e, sh);
    s = fe_reduce_weak<FrP>(fe_add<FrP>(s, run));
    if (k < m) st_canon(a.tmp, k, s);
    if (t == 255) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[36 + i] = s.l[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 9; ++i) run.l[i] = sh[36 + i];
    __syncthreads();
  }
  if (t == 0) st_tw(a.block_tot, b, run);
}
// exclusive prefix sum of the block totals (one block; serial over groups of 256)
__global__ void __launch_bounds__(256) ruffini_carry_kernel(u32x4* block_tot, u32 nblocks) {
  __shared__ u32 sh[4 * 9 + 9];
  const u32 t = threadIdx.x;
  Fr run = fe_zero<FrP>();
  for (u32 base = 0; base < nblocks; base += 256) {
    const u32 i = base + t;
    Fr v = i < nblocks ? ld_tw(block_tot, i) : fe_zero<FrP>();
    Fr s = block_scan_256(v, sh);                         // inclusive
    Fr incl = fe_reduce_weak<FrP>(fe_add<FrP>(s, run));
    // exclusive value for block i = inclusive - own = run + (s - v): recompute as run + scan of the others
    Fr excl = fe_reduce_weak<FrP>(fe_sub<FrP, 3, 1>(incl, v));
    if (i < nblocks) st_tw(block_tot, i, excl);
    if (t == 255) {
#pragma unroll
      for (int k = 0; k < 9; ++k) sh[36 + k] = incl.l[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) run.l[k] = sh[36 + k];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) ruffini_final_kernel(const ScanArgs a) {
  const size_t m = a.n - 1;
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= m) return;
  const size_t b = k / ((size_t)256 * a.L);
  Fr s = fe_add<FrP>(ld_canon(a.tmp, k), ld_tw(a.block_tot, b));              // (2, <2.1)
  Fr y = fe_mul<FrP>(s, two_level(a.z_hi, a.z_lo, (u32)k, a.lh));              // * z^k
  st_canon(a.out, m - 1 - k, fe_reduce_weak<FrP>(y));                          // q_{n-2-k}
}
// z == 0: q_{i-1} = c_i
__global__ void ruffini_shift_kernel(const u32x4* coeffs, u32x4* out, size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[2 * i] = coeffs[2 * (i + 1)];
  out[2 * i + 1] = coeffs[2 * (i + 1) + 1];
}

// ------------------------------------------------------------------ batch inversion
// Montgomery's trick per thread over `L` elements taken with stride T (coalesced), one Fermat
// inversion per thread.  Works in the device Montgomery domain: u = a * 2^5 (ABI -> device form),
// prefix products in device form, inverse in device form, result shifted back to ABI form.
__global__ void __launch_bounds__(256) batch_inverse_kernel(u32x4* v, size_t n, u32 L, u32x4* scratch) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Fr one = fe_one<FrP>();
  const Fr to_dev = fe_pow2<FrP, 2 * 261 - 256>();   // a * this / 2^261 = a * 2^5
  const Fr to_abi = fe_pow2<FrP, 256>();             // d * this / 2^261 = d / 2^5
  Fr acc = one;
  for (u32 j = 0; j < L; ++j) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    Fr a = fe_mul<FrP>(ld_canon(v, i), to_dev);
    u32 s[8];
    fe_canon_pack<FrP>(s, a);
    u32 nz = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) nz |= s[q];
    st_tw(scratch, i, acc);                    // product of the earlier non-zero elements
    if (nz) acc = fe_mul<FrP>(acc, a);
  }
  // acc^(r-2)
  Fr inv = one, base = acc;
  {
    constexpr u32 E[8] = {0xffffffffu, 0xfffffffeu, 0xfffe5bfeu, 0x53bda402u,
                          0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};  // r - 2
    for (int w = 0; w < 8; ++w) {
      for (int bit = 0; bit < 32; ++bit) {
        if ((E[w] >> bit) & 1) inv = fe_mul<FrP>(inv, base);
        base = fe_mul<FrP>(base, base);
      }
    }
  }
  for (u32 j = L; j-- > 0;) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) continue;
    Fr a = fe_mul<FrP>(ld_canon(v, i), to_dev);
    u32 s[8];
    fe_canon_pack<FrP>(s, a);
    u32 nz = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) nz |= s[q];
    if (!nz) continue;                         // zero stays zero
    Fr r = fe_mul<FrP>(inv, ld_tw(scratch, i));   // a_i^-1 in device form
    inv = fe_mul<FrP>(inv, a);
    st_canon(v, i, fe_mul<FrP>(r, to_abi));
  }
}

// ------------------------------------------------------------------ prefix product
// out[0] = 1, out[i] = prod_{j<i} a[j]: the grand-product accumulator z of the permutation
// argument.  Work-efficient three-kernel scan; the products run in the device Montgomery domain.
//   local : a tile of 256*PP_L elements is staged in LDS (coalesced), every thread multiplies its PP_L
//           consecutive elements, the 256 thread totals are scanned (shuffles + LDS), and a second
//           serial sweep writes the tile-local exclusive prefixes (48-byte entries) + the tile total
//   carry : exclusive scan of the tile totals, pre-multiplied by the device->ABI constant
//   final : out = local prefix * carry  (one product per element, lands in ABI form)
constexpr int PP_L = 8;
PM_DEV Fr fr_shfl_up_mul_scan(Fr v, u32 lane) {  // inclusive product scan inside a wave
  for (int d = 1; d < 64; d <<= 1) {
    Fr o = fr_shfl_up(v, d);
    if (lane >= (u32)d) v = fe_mul<FrP>(v, o);
  }
  return v;
}
__global__ void __launch_bounds__(256) prefix_prod_local_kernel(const u32x4* in, size_t n, u32x4* tmp, u32x4* tile_tot) {
  extern __shared__ u32 sh[];  // [9][256 * PP_L] limbs, then 4 * 9 words of wave totals
  constexpr int TILE = 256 * PP_L;
  u32* wave_tot = sh + 9 * TILE;
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const size_t base = (size_t)blockIdx.x * TILE;
  const Fr to_dev = fe_pow2<FrP, 2 * 261 - 256>();
  const Fr one = fe_one<FrP>();
#pragma unroll
  for (int j = 0; j < PP_L; ++j) {
    const u32 e = j * 256 + t;
    Fr u = one;                                            // past the end: neutral element
    if (base + e < n) u = fe_mul<FrP>(ld_canon(in, base + e), to_dev);
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[i * TILE + e] = u.l[i];
  }
  __syncthreads();
  Fr tot = one;
#pragma unroll
  for (int j = 0; j < PP_L; ++j) {
    Fr u;
#pragma unroll
    for (int i = 0; i < 9; ++i) u.l[i] = sh[i * TILE + t * PP_L + j];
    tot = j == 0 ? u : fe_mul<FrP>(tot, u);
  }
  Fr incl = fr_shfl_up_mul_scan(tot, lane);
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) wave_tot[wave * 9 + i] = incl.l[i];
  }
  Fr excl = fr_shfl_up(incl, 1);                           // exclusive inside the wave
  if (lane == 0) excl = one;
  __syncthreads();
  Fr carry = one;
  for (u32 w = 0; w < wave; ++w) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = wave_tot[w * 9 + i];
    carry = fe_mul<FrP>(carry, o);
  }
  Fr run = fe_mul<FrP>(excl, carry);                       // product of everything before this thread
#pragma unroll
  for (int j = 0; j < PP_L; ++j) {
    const u32 e = t * PP_L + j;
    Fr u;
#pragma unroll
    for (int i = 0; i < 9; ++i) u.l[i] = sh[i * TILE + e];
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[i * TILE + e] = run.l[i];   // exclusive prefix of element e
    run = fe_mul<FrP>(run, u);
  }
  if (t == 255) st_tw(tile_tot, blockIdx.x, run);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PP_L; ++j) {                           // coalesced write-out
    const u32 e = j * 256 + t;
    if (base + e < n) {
      Fr v;
#pragma unroll
      for (int i = 0; i < 9; ++i) v.l[i] = sh[i * TILE + e];
      st_tw(tmp, base + e, v);
    }
  }
}
__global__ void __launch_bounds__(256) prefix_prod_carry_kernel(u32x4* tile_tot, u32 ntiles) {
  __shared__ u32 sh[4 * 9 + 9];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const Fr one = fe_one<FrP>();
  const Fr to_abi = fe_pow2<FrP, 256>();
  Fr run = one;
  for (u32 base = 0; base < ntiles; base += 256) {
    const u32 i = base + t;
    Fr v = i < ntiles ? ld_tw(tile_tot, i) : one;
    Fr incl = fr_shfl_up_mul_scan(v, lane);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 9; ++k) sh[wave * 9 + k] = incl.l[k];
    }
    Fr excl = fr_shfl_up(incl, 1);
    if (lane == 0) excl = one;
    __syncthreads();
    Fr carry = run;
    for (u32 w = 0; w < wave; ++w) {
      Fr o;
#pragma unroll
      for (int k = 0; k < 9; ++k) o.l[k] = sh[w * 9 + k];
      carry = fe_mul<FrP>(carry, o);
    }
    excl = fe_mul<FrP>(excl, carry);
    if (i < ntiles) st_tw(tile_tot, i, fr_canon(fe_mul<FrP>(excl, to_abi)));
    if (t == 255) {
      Fr all = fe_mul<FrP>(excl, v);                       // inclusive through the last tile of the group
#pragma unroll
      for (int k = 0; k < 9; ++k) sh[36 + k] = all.l[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) run.l[k] = sh[36 + k];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) prefix_prod_final_kernel(const u32x4* tmp, const u32x4* tile_carry, size_t n,
                                                                 u32x4* out) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  st_canon(out, k, fe_mul<FrP>(ld_tw(tmp, k), ld_tw(tile_carry, k / (256 * PP_L))));
}

// ------------------------------------------------------------------ host helpers
// out[i] = base^(i * stride), `count` device-form entries of 48 bytes at `out` (caller-provided memory)
static void build_pow(u32x4* out, const HFr& base, u32 count, u32 stride, hipStream_t st) {
  NttConsts c;
  memset(&c, 0, sizeof c);
  to_limbs29(c.w8[0], base);
  to_limbs29(c.scale, host::one(host::FR()));
  to_limbs29(c.one, host::one(host::FR()));
  hipLaunchKernelGGL(pow_table_kernel, dim3((count + 255) / 256), dim3(256), 0, st, out, c, count, stride);
}

}  // namespace pm

using namespace pm;

// ------------------------------------------------------------------ C ABI
extern "C" int pm_dev_alloc(pm_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  PM_HIP(ctx, hipMalloc(out, bytes ? bytes : 16));
  return PM_OK;
}
extern "C" int pm_dev_free(pm_ctx* ctx, void* p) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (p) PM_HIP(ctx, hipFree(p));
  return PM_OK;
}
extern "C" int pm_dev_upload(pm_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !src))) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  if (bytes) PM_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PM_OK;
}
extern "C" int pm_dev_download(pm_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !d_src))) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  if (bytes) PM_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PM_OK;
}

extern "C" int pm_fr_vec_op_dev(pm_ctx* ctx, int op, const void* d_a, const void* d_b, size_t b_len, void* d_out,
                                size_t n, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (op < 0 || op > 2) return set_err(ctx, PM_ERR_BAD_ARG, "op must be 0 (add), 1 (sub) or 2 (mul)");
  if (b_len != 1 && b_len != n) return set_err(ctx, PM_ERR_LENGTH, "b must have n elements or one (broadcast)");
  if (n == 0) return PM_OK;
  if (!d_a || !d_b || !d_out) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cus * 32);
  const u32x4 *a = (const u32x4*)d_a, *b = (const u32x4*)d_b;
  u32x4* o = (u32x4*)d_out;
  ProfScope prof(ctx, st, op == 0 ? "fr_vec_add" : (op == 1 ? "fr_vec_sub" : "fr_vec_mul"));
  if (op == 0) hipLaunchKernelGGL((vec_op_kernel<0>), dim3(blocks), dim3(256), 0, st, a, b, b_len, o, n);
  if (op == 1) hipLaunchKernelGGL((vec_op_kernel<1>), dim3(blocks), dim3(256), 0, st, a, b, b_len, o, n);
  if (op == 2) hipLaunchKernelGGL((vec_op_kernel<2>), dim3(blocks), dim3(256), 0, st, a, b, b_len, o, n);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_fr_poly_evaluate_many_dev(pm_ctx* ctx, uint32_t k, const void* const* d_polys, size_t n,
                                            const uint64_t point[4], uint64_t* out, void* hip_stream) {
  if (!ctx || !point || !out) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (k == 0) return PM_OK;
  if (k > PM_LINCOMB_MAX) return set_err(ctx, PM_ERR_BAD_ARG, "k must be in 1..PM_LINCOMB_MAX");
  if (n == 0) {
    memset(out, 0, 32 * (size_t)k);
    return PM_OK;
  }
  if (!d_polys) return set_err(ctx, PM_ERR_BAD_ARG, "null pointer");
  EvalPolys polys;
  memset(&polys, 0, sizeof polys);
  for (uint32_t j = 0; j < k; ++j) {
    if (!d_polys[j]) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
    polys.p[j] = (const u32x4*)d_polys[j];
  }
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  int rc = order_on(ctx, ctx->ord_poly, st);
  if (rc) return rc;
  const u32 L = (u32)std::max<size_t>(1, std::min<size_t>(64, n / ((size_t)256 * 1024)));
  const size_t seg = (size_t)256 * L;
  const u32 nblocks = (u32)((n + seg - 1) / seg);
  HFr x;
  memcpy(x.l, point, 32);
  EvalConsts kc;
  to_limbs29(kc.x, x);
  to_limbs29(kc.xrow, hfr_pow_u64(x, 256));
  to_limbs29(kc.xseg, hfr_pow_u64(x, seg));
  to_limbs29(kc.one, host::one(host::FR()));
  rc = ensure_buffer(ctx, ctx->poly_ws, (size_t)(256 + (1 + (size_t)k) * nblocks) * 48 + 32 * (size_t)k + 64);
  if (rc) return rc;
  u32x4* xpow = (u32x4*)ctx->poly_ws.ptr;
  u32x4* xblk = xpow + 3 * 256;
  u32x4* partial = xblk + 3 * (size_t)nblocks;
  u32x4* d_out = partial + 3 * (size_t)nblocks * k;
  {
    NttConsts c;
    memset(&c, 0, sizeof c);
    memcpy(c.w8[0], kc.x, sizeof kc.x);
    memcpy(c.scale, kc.one, sizeof kc.one);
    hipLaunchKernelGGL(pow_table_kernel, dim3(1), dim3(256), 0, st, xpow, c, 256u, 1u);
    hipLaunchKernelGGL(pow_table_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, st, xblk, c, nblocks, (u32)seg);
  }
  {
    ProfScope prof(ctx, st, "fr_poly_evaluate");
    hipLaunchKernelGGL(poly_eval_kernel, dim3(nblocks, k), dim3(256), 0, st, polys, n, L, kc, (const u32x4*)xpow,
                       (const u32x4*)xblk, partial);
    hipLaunchKernelGGL(poly_eval_final_kernel, dim3(k), dim3(256), 0, st, (const u32x4*)partial, nblocks, d_out);
  }
  PM_HIP(ctx, hipGetLastError());
  PM_HIP(ctx, hipMemcpyAsync(out, d_out, 32 * (size_t)k, hipMemcpyDeviceToHost, st));
  PM_HIP(ctx, hipStreamSynchronize(st));
  return PM_OK;
}

extern "C" int pm_fr_poly_evaluate_dev(pm_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t point[4],
                                       uint64_t out[4], void* hip_stream) {
  if (!ctx || !point || !out) return PM_ERR_BAD_ARG;
  if (n && !d_coeffs) return PM_ERR_BAD_ARG;
  const void* one[1] = {d_coeffs};
  return pm_fr_poly_evaluate_many_dev(ctx, 1, one, n, point, out, hip_stream);
}

extern "C" int pm_fr_poly_ruffini_dev(pm_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t z[4], void* d_out,
                                      void* hip_stream) {
  if (!ctx || !z) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n <= 1) return PM_OK;  // the quotient of a constant is the zero polynomial (no coefficients)
  if (!d_coeffs || !d_out) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  if (d_coeffs == d_out) return set_err(ctx, PM_ERR_BAD_ARG, "ruffini is not in place");
  if (n > ((size_t)1 << 31)) return set_err(ctx, PM_ERR_LENGTH, "n > 2^31");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const size_t m = n - 1;
  const host::Field<4>& F = host::FR();
  HFr zz;
  memcpy(zz.l, z, 32);
  if (host::is_zero(zz)) {
    hipLaunchKernelGGL(ruffini_shift_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
                       (const u32x4*)d_coeffs, (u32x4*)d_out, m);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
  }
  u32 lg = 0;
  while (((size_t)1 << lg) < m) ++lg;
  const u32 lh = (lg + 1) / 2, n_lo = 1u << lh, n_hi = (u32)((m >> lh) + 1);
  const u32 L = (u32)std::max<size_t>(1, std::min<size_t>(16, m / ((size_t)256 * 1024)));
  const u32 nblocks = (u32)((m + (size_t)256 * L - 1) / ((size_t)256 * L));
  // the four power tables live in a context buffer: no allocator calls (and no device-wide
  // synchronisation from hipFree) inside a proving loop
  HFr zinv = host::inv(zz, F);
  const size_t tab_entries = 2 * ((size_t)n_lo + n_hi);
  int rc = order_on(ctx, ctx->ord_poly, st);
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_tab, tab_entries * 48);
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, m * 32 + (size_t)nblocks * 48 + 64);
  if (rc) return rc;
  u32x4* tab = (u32x4*)ctx->poly_tab.ptr;
  u32x4 *zi_lo = tab, *zi_hi = zi_lo + 3 * (size_t)n_lo, *z_lo = zi_hi + 3 * (size_t)n_hi, *z_hi = z_lo + 3 * (size_t)n_lo;
  build_pow(zi_lo, zinv, n_lo, 1, st);
  build_pow(zi_hi, zinv, n_hi, n_lo, st);
  build_pow(z_lo, zz, n_lo, 1, st);
  build_pow(z_hi, zz, n_hi, n_lo, st);
  ScanArgs a;
  a.coeffs = (const u32x4*)d_coeffs;
  a.tmp = (u32x4*)ctx->poly_ws.ptr;
  a.block_tot = a.tmp + 2 * m;
  a.zi_hi = zi_hi;
  a.zi_lo = zi_lo;
  a.z_hi = z_hi;
  a.z_lo = z_lo;
  a.out = (u32x4*)d_out;
  a.n = n;
  a.lh = lh;
  a.L = L;
  ProfScope prof(ctx, st, "fr_poly_ruffini");
  hipLaunchKernelGGL(ruffini_local_kernel, dim3(nblocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ruffini_carry_kernel, dim3(1), dim3(256), 0, st, a.block_tot, nblocks);
  hipLaunchKernelGGL(ruffini_final_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_fr_prefix_product_dev(pm_ctx* ctx, const void* d_in, size_t n, void* d_out, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  if (!d_in || !d_out) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const size_t tile = (size_t)256 * PP_L;
  const u32 ntiles = (u32)((n + tile - 1) / tile);
  int rc = order_on(ctx, ctx->ord_poly, st);
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, (n + ntiles) * 48 + 64);
  if (rc) return rc;
  u32x4* tmp = (u32x4*)ctx->poly_ws.ptr;
  u32x4* tile_tot = tmp + 3 * n;
  const size_t lds = (9 * tile + 36) * 4;
  PM_HIP(ctx, hipFuncSetAttribute((const void*)prefix_prod_local_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  ProfScope prof(ctx, st, "fr_prefix_product");
  hipLaunchKernelGGL(prefix_prod_local_kernel, dim3(ntiles), dim3(256), lds, st, (const u32x4*)d_in, n, tmp, tile_tot);
  hipLaunchKernelGGL(prefix_prod_carry_kernel, dim3(1), dim3(256), 0, st, tile_tot, ntiles);
  hipLaunchKernelGGL(prefix_prod_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32x4*)tmp,
                     (const u32x4*)tile_tot, n, (u32x4*)d_out);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_fr_batch_inverse_dev(pm_ctx* ctx, void* d_inout, size_t n, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  if (!d_inout) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  // enough threads to fill the chip, at most 64 elements per thread
  // one Fermat inversion (~380 products) per thread: 64 elements per thread amortise it to ~6
  // products per element; below 2^22 elements keep at least one wave per SIMD instead
  const size_t want_threads = std::max<size_t>((n + 63) / 64, std::min<size_t>(n, (size_t)ctx->num_cus * 256));
  const unsigned blocks = (unsigned)((want_threads + 255) / 256);
  const size_t T = (size_t)blocks * 256;
  const u32 L = (u32)((n + T - 1) / T);
  int rc = order_on(ctx, ctx->ord_poly, st);
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, n * 48);
  if (rc) return rc;
  ProfScope prof(ctx, st, "fr_batch_inverse");
  hipLaunchKernelGGL(batch_inverse_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)d_inout, n, L,
                     (u32x4*)ctx->poly_ws.ptr);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}
