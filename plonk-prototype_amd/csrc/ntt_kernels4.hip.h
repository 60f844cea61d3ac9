// Radix-4 variant of the Stockham pass kernel (see ntt_kernels.hip.h for the algorithm): every
// thread keeps FOUR elements (36 VGPRs of data instead of 72) and a wave owns 9 KiB of the LDS
// tile instead of 18, so four waves per SIMD fit where the radix-8 kernel is limited to two by
// LDS.  Same passes, same tables for the inter-pass twiddles, its own step-twiddle tables.
#pragma once
#include "ntt_kernels.hip.h"

namespace pm {

__host__ __device__ constexpr int step4_radix_log(int S, int s) { return (S - 2 * s) >= 2 ? 2 : 1; }
__host__ __device__ constexpr int num_steps4(int S) { return (S + 1) / 2; }
__host__ __device__ constexpr int step4_tw_offset(int S, int s) {
  int off = 0;
  for (int i = 0; i < s; ++i) off += ((1 << step4_radix_log(S, i)) - 1) << (2 * i);
  return off;
}
__host__ __device__ constexpr int step4_tw_total(int S) { return step4_tw_offset(S, num_steps4(S)); }
__host__ __device__ constexpr size_t pass4_lds_bytes(int S, int LT) {
  return num_steps4(S) > 1 ? ((size_t)36 << (S + LT)) : 0;
}

template <int S, int LT, int STEP, bool OUT_UFAST, bool OUT_WIDE>
PM_DEV void ntt_step4(Fr (&x)[4], const NttPassArgs& a, const NttConsts& kc, u32x4* lds0, u32x4* lds1, u32* lds2,
                      const Fr& w4, u32 tid, size_t j0) {
  constexpr int R = 1 << S;
  constexpr int T = 1 << LT;
  constexpr int U = R / 4;
  constexpr int NSTEPS = num_steps4(S);
  constexpr int LQ = step4_radix_log(S, STEP);
  constexpr u32 nsp = 1u << (2 * STEP);  // Ns'
  constexpr bool last = (STEP == NSTEPS - 1);
  constexpr bool ufast = OUT_UFAST && last;
  const u32 c = ufast ? tid / U : tid & (T - 1);
  const u32 u = ufast ? tid % U : tid >> LT;
  if (STEP > 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const u32 e = (u + m * U) * T + c;
      u32x4 lo = lds0[e], hi = lds1[e];
      x[m].l[0] = lo.x; x[m].l[1] = lo.y; x[m].l[2] = lo.z; x[m].l[3] = lo.w;
      x[m].l[4] = hi.x; x[m].l[5] = hi.y; x[m].l[6] = hi.z; x[m].l[7] = hi.w;
      x[m].l[8] = lds2[e];
    }
  }
  const u32x4* stw = a.step_tw + 3 * step4_tw_offset(S, STEP);
  if constexpr (LQ == 2) {
    const u32 kp = u & (nsp - 1);
    if (STEP > 0) {
      x[0] = fe_reduce_weak<FrP>(x[0]);
      x[1] = fe_mul<FrP>(x[1], ld_tw(stw, 0 * nsp + kp));
      x[2] = fe_mul<FrP>(x[2], ld_tw(stw, 1 * nsp + kp));
      x[3] = fe_mul<FrP>(x[3], ld_tw(stw, 2 * nsp + kp));
    }
    dft4(x[0], x[1], x[2], x[3], w4);  // X[t] in x[t]
    if constexpr (!last) {
      __syncthreads();
      const u32 base = (u - kp) * 4 + kp;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const u32 e = (base + t * nsp) * T + c;
        lds0[e] = u32x4{x[t].l[0], x[t].l[1], x[t].l[2], x[t].l[3]};
        lds1[e] = u32x4{x[t].l[4], x[t].l[5], x[t].l[6], x[t].l[7]};
        lds2[e] = x[t].l[8];
      }
      __syncthreads();
    }
    // last: Ns' = R/4, k' = u: X[t] belongs to row u + t U = x[t] already
  } else {  // radix 2, always the last step: Ns' = R/2, k' = v = u + i U, pairs (x[i], x[i+2])
    x[0] = fe_norm<FrP>(x[0]);
    x[1] = fe_norm<FrP>(x[1]);
    x[2] = fe_mul<FrP>(x[2], ld_tw(stw, u));
    x[3] = fe_mul<FrP>(x[3], ld_tw(stw, u + U));
    BFLY(3, x[0], x[2]);
    BFLY(3, x[1], x[3]);
  }
  if constexpr (last) {
    const size_t n = (size_t)1 << a.log_n;
    const size_t j = j0 + c;
    const size_t ns = (size_t)1 << a.log_ns;
    const size_t k = j & (ns - 1);
    const size_t obase = (j - k) * R + k;
    if constexpr (OUT_WIDE) {
      const WidePtr wout = wide_ptrs(a.out, a.wide_glog);
      const size_t boff = (size_t)blockIdx.y * n;
#pragma unroll
      for (int m = 0; m < 4; ++m) st_wide(wout, boff + obase + (size_t)(u + m * U) * ns, x[m]);
    } else {
      u32x4* gout = reinterpret_cast<u32x4*>(a.out) + 2 * (size_t)blockIdx.y * a.batch_stride_out;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const size_t go = obase + (size_t)(u + m * U) * ns;
        Fr v = x[m];
        if (a.flags & PASS_POST_SCALE) v = fe_mul<FrP>(v, fr_limbs(kc.scale));
        if (a.flags & PASS_POST_COSET) v = fe_mul<FrP>(v, two_level(a.cs_hi, a.cs_lo, (u32)go, a.lh));
        if (!(a.flags & (PASS_POST_SCALE | PASS_POST_COSET))) v = fe_reduce_weak<FrP>(v);
        fe_store<FrP>(gout + 2 * go, v);
      }
    }
  }
}

template <int S, int LT, bool OUT_UFAST, bool IN_WIDE, bool OUT_WIDE>
__global__ void __launch_bounds__((1 << (S + LT)) / 4 < 64 ? 64 : (1 << (S + LT)) / 4)
    ntt_pass4_kernel(const NttPassArgs a, const NttConsts kc) {
  constexpr int R = 1 << S;
  constexpr int T = 1 << LT;
  constexpr int U = R / 4;
  constexpr int NTHREADS = U * T;
  constexpr int NSTEPS = num_steps4(S);
  extern __shared__ u32x4 lds[];
  u32x4* lds0 = lds;
  u32x4* lds1 = lds + R * T;
  u32* lds2 = reinterpret_cast<u32*>(lds + 2 * R * T);

  const u32 tid = threadIdx.x;
  if (NTHREADS < 64 && tid >= NTHREADS) return;
  const u32 log_n = a.log_n;
  const size_t n = (size_t)1 << log_n;
  const size_t n_cols = (size_t)1 << (log_n - S);
  const size_t j0 = (size_t)xcd_tile(blockIdx.x, gridDim.x, a.flags) * T;
  const Fr w4 = fr_limbs(kc.w8[1]);

  Fr x[4];
  {
    constexpr bool ufast = OUT_UFAST && NSTEPS == 1;
    const u32 c = ufast ? tid / U : tid & (T - 1);
    const u32 u = ufast ? tid % U : tid >> LT;
    const size_t j = j0 + c;
    if constexpr (IN_WIDE) {
      const WidePtr win = wide_ptrs(const_cast<void*>(a.in), a.wide_glog);
      const size_t boff = (size_t)blockIdx.y * n;
      const u32 k = (u32)(j & (((size_t)1 << a.log_ns) - 1));
      const u32 tw_shift = log_n - a.log_ns - S;
      if (a.flags & PASS_DIRECT_TW) {
        const WidePtr wtw = wide_ptrs(const_cast<void*>(a.pass_tw), a.wide_glog);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const size_t idx = j + (size_t)(u + m * U) * n_cols;
          x[m] = fe_mul<FrP>(ld_wide(win, boff + idx), ld_wide(wtw, idx));
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const u32 row = u + m * U;
          Fr v = ld_wide(win, boff + j + (size_t)row * n_cols);
          x[m] = fe_mul<FrP>(v, fr_canon(two_level(a.tw_hi, a.tw_lo, (k * row) << tw_shift, a.lh)));
        }
      }
    } else {
      const u32x4* gin = reinterpret_cast<const u32x4*>(a.in) + 2 * (size_t)blockIdx.y * a.batch_stride_in;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const size_t gi = j + (size_t)(u + m * U) * n_cols;
        if (gi < a.in_len) {
          x[m] = fe_load<FrP>(gin + 2 * gi);
          if (a.flags & PASS_PRE_COSET)
            x[m] = fe_mul<FrP>(x[m], two_level(a.cs_hi, a.cs_lo, (u32)gi, a.lh));
        } else {
          x[m] = fe_zero<FrP>();
        }
      }
    }
  }
  ntt_step4<S, LT, 0, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
  if constexpr (NSTEPS > 1) ntt_step4<S, LT, 1, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
  if constexpr (NSTEPS > 2) ntt_step4<S, LT, 2, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
  if constexpr (NSTEPS > 3) ntt_step4<S, LT, 3, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
  if constexpr (NSTEPS > 4) ntt_step4<S, LT, 4, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
  if constexpr (NSTEPS > 5) ntt_step4<S, LT, 5, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w4, tid, j0);
}

// step twiddles of the radix-4 kernel: block s, entry [(t-1)*Ns' + k'] = wR^(k' t R/(Ns' q)), Ns' = 4^s
static __global__ void step4_tw_kernel(u32x4* out, const NttConsts c, u32 S) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 off = 0;
  for (u32 s = 0; 2 * s < S; ++s) {
    const u32 lq = (S - 2 * s) >= 2 ? 2 : 1;
    const u32 nsp = 1u << (2 * s);
    const u32 cnt = ((1u << lq) - 1) * nsp;
    if (i >= off && i < off + cnt) {
      const u32 t = (i - off) / nsp + 1, kp = (i - off) % nsp;
      const u32 e = (kp * t) << (S - 2 * s - lq);
      st_tw(out, i, fr_canon(fr_pow(fr_limbs(c.w8[0]), e, fr_limbs(c.one))));
      return;
    }
    off += cnt;
  }
}

}  // namespace pm
