// Stockham autosort NTT over BLS12-381 Fr for gfx950: natural order in, natural order out.
//
// Replaces the body of dusk_plonk::fft::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
// -> best_fft -> serial_fft (dusk-plonk 0.8.2, pinned at ref:Cargo.toml:19).
//
// One launch = one Stockham PASS of radix R = 2^S over the whole vector:
//     j in [0, N/R), k = j mod Ns (Ns = product of earlier pass radices)
//     x[t]  = in[j + t*N/R] * w_{Ns*R}^(k*t)            t in [0, R)      (strided columns)
//     X     = DFT_R(x)
//     out[(j-k)*R + k + t*Ns] = X[t]
// A workgroup owns a tile of T = 2^LT adjacent columns j (so every global access is a run of
// T contiguous elements, or of R elements in the first pass) and computes the R-point DFTs
// inside the tile with the same recurrence at radix 8 (then 4 or 2): every thread keeps
// 8 Fr elements (72 VGPRs of 29-bit limbs) in registers, the first in-tile step is fed
// straight from HBM, the last one stores straight to HBM, and the steps in between exchange
// through LDS (limb planes of 16+16+4 bytes so consecutive lanes hit consecutive banks).
//
// Number representation (fields.hip.h): 9 x 29-bit limbs, lazily reduced.  Between passes the
// vector lives in HBM in the same 9-limb "wide" form (36 B/element, blocked by 4), so only
// the first load unpacks canonical data and only the last store canonicalises.
//
// Bounds are written (B, V): limbs < B*2^29, value < V*r.  fe_mul needs its left operand at
// B < 6 and returns (1, <2) for every V < 68 (R = 2^261 = 68.6 r).
//
// Cost: 13 Montgomery products per radix-8 butterfly (8 twiddles incl. a reducing multiply
// by one on the untwiddled input, 5 constants); the kernel is integer-ALU bound.
#pragma once
#include "fields.hip.h"

namespace pm {

struct NttConsts {
  u32 w8[3][9];  // w8^1, w8^2 (= w4), w8^3 for this direction; limbs, Montgomery R' = 2^261
  u32 one[9];    // 1 in R' form (the reducing multiplier)
  u32 scale[9];  // n^-1 in R' form (ifft); == one for the forward transform
};

struct NttPassArgs {
  const void* in;         // canonical: N x 32 B.   wide: planes (see wide_ptrs)
  void* out;
  const u32x4* step_tw;   // in-tile step twiddles for this S, 48 B per entry
  const u32x4* tw_hi;     // w_N^(x << lh)        x < N >> lh        (48 B entries)
  const u32x4* tw_lo;     // w_N^x                x < 1 << lh
  const u32x4* cs_hi;     // coset powers g^(x << lh) (fwd) or g^-(x << lh) (inv)
  const u32x4* cs_lo;     // g^x / g^-x           x < 1 << lh
  const void* pass_tw;    // PASS_DIRECT_TW: this pass's input twiddles as wide planes, element order
  unsigned long long batch_stride_in;   // elements between batch vectors (canonical side)
  unsigned long long batch_stride_out;
  unsigned long long wide_total;        // elements in one wide buffer (batch * N)
  u32 wide_glog;          // log2 of the wide layout's block size (2 or 3)
  u32 in_len;             // elements present in `in`; the rest of the domain reads as zero
  u32 log_n;
  u32 log_ns;             // log2(Ns)
  u32 lh;                 // split point of the two-level tables
  u32 flags;
};
enum : u32 {
  PASS_DIRECT_TW = 1u,   // input twiddles come from pass_tw (one load) instead of hi*lo
  PASS_PRE_COSET = 2u,   // multiply input i by g^i (coset_fft)
  PASS_POST_SCALE = 4u,  // multiply outputs by consts.scale (single-pass ifft; otherwise n^-1 is
                         // folded into the last pass's twiddle table)
  PASS_POST_COSET = 8u,  // multiply output i by cs_hi/lo (coset_ifft)
  PASS_XCD_REMAP = 16u,  // blockIdx -> tile so that each XCD walks a contiguous range of tiles
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Giving XCD x
// the tiles [x*nb/8, (x+1)*nb/8) keeps neighbouring tiles -- which share the cache lines at their
// common border in the blocked wide layout and read neighbouring twiddle-table lines -- behind one
// L2.  Placement only changes speed, never results.
PM_DEV u32 xcd_tile(u32 b, u32 nb, u32 flags) {
  if (!(flags & PASS_XCD_REMAP) || (nb & 7u)) return b;
  return (b & 7u) * (nb >> 3) + (b >> 3);
}

// ---- table entries: 9 limbs in 48 bytes -------------------------------------------------
PM_DEV Fr ld_tw(const u32x4* tab, size_t idx) {
  const u32x4* p = tab + 3 * idx;
  u32x4 a = p[0], b = p[1];
  u32 c = reinterpret_cast<const u32*>(p + 2)[0];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c;
  return r;
}
PM_DEV void st_tw(u32x4* tab, size_t idx, const Fr& v) {
  u32x4* p = tab + 3 * idx;
  p[0] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
  p[1] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
  p[2] = u32x4{v.l[8], 0u, 0u, 0u};
}
// ---- wide vectors: 9 limbs per element, 36 B/element, blocked by G = 4 or 8 elements: each group
// of G consecutive elements is one 36 G-byte block [G x limbs 0-3 | G x limbs 4-7 | G x limb 8]
// (G = the smallest tile width T of the plan, so a lane group reads whole blocks).
// A pass touches groups of T >= 4 adjacent columns, so the three 16/16/4-byte pieces of a lane
// quad land in the same one or two cache lines.  (Measured alternatives at 2^20 / 2^24: three
// separate limb planes fetch 2.3x the ideal bytes and make the 2^24 passes HBM-bound; 48-byte
// records coalesce but move 33 % more bytes and lose 10 % at 2^24.)
struct WidePtr {
  u32x4* p;
  u32 glog;  // log2 of the block's element count (2 or 3)
};
PM_DEV WidePtr wide_ptrs(void* base, u32 glog) {
  WidePtr w;
  w.p = reinterpret_cast<u32x4*>(base);
  w.glog = glog;
  return w;
}
PM_DEV Fr ld_wide(const WidePtr& w, size_t i) {
  const u32 G = 1u << w.glog;
  const u32x4* q = w.p + (size_t)(9 * G / 4) * (i >> w.glog);
  const u32 e = (u32)i & (G - 1);
  u32x4 a = q[e], b = q[G + e];
  u32 c = reinterpret_cast<const u32*>(q + 2 * G)[e];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c;
  return r;
}
PM_DEV void st_wide(const WidePtr& w, size_t i, const Fr& v) {
  const u32 G = 1u << w.glog;
  u32x4* q = w.p + (size_t)(9 * G / 4) * (i >> w.glog);
  const u32 e = (u32)i & (G - 1);
  q[e] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
  q[G + e] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
  reinterpret_cast<u32*>(q + 2 * G)[e] = v.l[8];
}

PM_DEV Fr fr_limbs(const u32* c) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = c[i];
  return r;
}

// w^e from the two-level table: hi[e >> lh] * lo[e & mask]   -> (1, <2)
PM_DEV Fr two_level(const u32x4* hi, const u32x4* lo, u32 e, u32 lh) {
  return fe_mul<FrP>(ld_tw(hi, e >> lh), ld_tw(lo, e & ((1u << lh) - 1u)));
}

// canonical limbs (value < r) for table entries: reduce a (1, <2) product fully
PM_DEV Fr fr_canon(const Fr& a) {
  u32 s[8];
  fe_canon_pack<FrP>(s, a);
  return fe_unpack<FrP>(s);
}
PM_DEV Fr fr_pow(Fr b, unsigned long long e, Fr acc) {
  while (e) {
    if (e & 1) acc = fe_mul<FrP>(acc, b);
    b = fe_mul<FrP>(b, b);
    e >>= 1;
  }
  return acc;
}

// (x, y) <- (x + y, x - y + K r);  y limbs <= 2^30 - 2, y value < (K-1) r
#define BFLY(K, A, B)                       \
  {                                         \
    Fr _s = fe_add<FrP>(A, B);              \
    Fr _d = fe_sub<FrP, K, 1>(A, B);        \
    A = _s;                                 \
    B = _d;                                 \
  }

// 8-point DIF network.  In: x[1..7] = (1, <2) products, x[0] = (<=1+, <V0) with V0 <= 24.
// Out: x[p] = X[bitrev3(p)], every output (B < 5, V < 40).
PM_DEV void dft8(Fr* x, const Fr& w1, const Fr& w2, const Fr& w3) {
  BFLY(3, x[0], x[4]);  // sums (2,.), diffs (4,.)
  BFLY(3, x[1], x[5]);
  BFLY(3, x[2], x[6]);
  BFLY(3, x[3], x[7]);
  x[5] = fe_mul<FrP>(x[5], w1);
  x[6] = fe_mul<FrP>(x[6], w2);
  x[7] = fe_mul<FrP>(x[7], w3);
  BFLY(5, x[0], x[2]);  // x0 (4,.) x2 (5,.)
  BFLY(5, x[1], x[3]);
  BFLY(3, x[4], x[6]);  // x4 (5,.) x6 (7,.)
  BFLY(3, x[5], x[7]);  // x5 (2,4) x7 (4,5)
  x[3] = fe_mul<FrP>(x[3], w2);
  x[7] = fe_mul<FrP>(x[7], w2);
  x[0] = fe_norm<FrP>(x[0]);
  x[1] = fe_norm<FrP>(x[1]);
  x[2] = fe_norm<FrP>(x[2]);
  x[4] = fe_norm<FrP>(x[4]);
  x[6] = fe_norm<FrP>(x[6]);
  BFLY(9, x[0], x[1]);
  BFLY(3, x[2], x[3]);
  BFLY(5, x[4], x[5]);
  BFLY(3, x[6], x[7]);
}
// 4-point DIF; a0 = (<=1+, <24) untwiddled, a1..a3 = (1, <2).  Outputs X0..X3 in a0..a3.
PM_DEV void dft4(Fr& a0, Fr& a1, Fr& a2, Fr& a3, const Fr& w4) {
  BFLY(3, a0, a2);  // a0 (2+, <26)  a2 (4+, <27)
  BFLY(3, a1, a3);  // a1 (2, 4)     a3 (4, 5)
  a3 = fe_mul<FrP>(a3, w4);
  a2 = fe_norm<FrP>(a2);
  BFLY(5, a0, a1);  // a0 = X0, a1 = X2
  BFLY(3, a2, a3);  // a2 = X1, a3 = X3
  Fr t = a1;
  a1 = a2;
  a2 = t;
}

// Offsets (in entries) of each in-tile step's twiddle block inside step_tw for radix 2^S:
// step s has sub-size Ns' = 8^s and radix q; block = (q-1) * Ns' entries laid out [t-1][k'].
__host__ __device__ constexpr int step_radix_log(int S, int s) { return (S - 3 * s) >= 3 ? 3 : (S - 3 * s); }
__host__ __device__ constexpr int num_steps(int S) { return (S + 2) / 3; }
__host__ __device__ constexpr int step_tw_offset(int S, int s) {
  int off = 0;
  for (int i = 0; i < s; ++i) off += ((1 << step_radix_log(S, i)) - 1) << (3 * i);
  return off;
}
__host__ __device__ constexpr int step_tw_total(int S) { return step_tw_offset(S, num_steps(S)); }

// LDS bytes per tile: 36 B per element
__host__ __device__ constexpr size_t pass_lds_bytes(int S, int LT) {
  return num_steps(S) > 1 ? ((size_t)36 << (S + LT)) : 0;
}

// One in-tile Stockham step (compile-time step index so every x[] index is a constant).
template <int S, int LT, int STEP, bool OUT_UFAST, bool OUT_WIDE>
PM_DEV void ntt_step(Fr (&x)[8], const NttPassArgs& a, const NttConsts& kc, u32x4* lds0, u32x4* lds1,
                     u32* lds2, const Fr& w8_1, const Fr& w8_2, const Fr& w8_3, u32 tid, size_t j0) {
  constexpr int R = 1 << S;
  constexpr int T = 1 << LT;
  constexpr int U = R / 8;
  constexpr int NSTEPS = num_steps(S);
  constexpr int LQ = step_radix_log(S, STEP);
  constexpr int Q = 1 << LQ;
  constexpr u32 nsp = 1u << (3 * STEP);  // Ns'
  constexpr bool last = (STEP == NSTEPS - 1);
  constexpr bool ufast = OUT_UFAST && last;
  const u32 c = ufast ? tid / U : tid & (T - 1);
  const u32 u = ufast ? tid % U : tid >> LT;
  if (STEP > 0) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const u32 e = (u + m * U) * T + c;
      u32x4 lo = lds0[e], hi = lds1[e];
      x[m].l[0] = lo.x; x[m].l[1] = lo.y; x[m].l[2] = lo.z; x[m].l[3] = lo.w;
      x[m].l[4] = hi.x; x[m].l[5] = hi.y; x[m].l[6] = hi.z; x[m].l[7] = hi.w;
      x[m].l[8] = lds2[e];
    }
  }
  const u32x4* stw = a.step_tw + 3 * step_tw_offset(S, STEP);
  // sub-butterfly i (i < 8/Q) has index v = u + i*U and works on x[i + (8/Q)*t], t < Q
  if constexpr (Q == 8) {
    const u32 kp = u & (nsp - 1);
    if (STEP > 0) {
      x[0] = fe_reduce_weak<FrP>(x[0]);  // the untwiddled input: (<6, <40) -> (1, <1.01)
      x[1] = fe_mul<FrP>(x[1], ld_tw(stw, 0 * nsp + kp));
      x[2] = fe_mul<FrP>(x[2], ld_tw(stw, 1 * nsp + kp));
      x[3] = fe_mul<FrP>(x[3], ld_tw(stw, 2 * nsp + kp));
      x[4] = fe_mul<FrP>(x[4], ld_tw(stw, 3 * nsp + kp));
      x[5] = fe_mul<FrP>(x[5], ld_tw(stw, 4 * nsp + kp));
      x[6] = fe_mul<FrP>(x[6], ld_tw(stw, 5 * nsp + kp));
      x[7] = fe_mul<FrP>(x[7], ld_tw(stw, 6 * nsp + kp));
    }
    dft8(x, w8_1, w8_2, w8_3);
    // x[p] = X[bitrev3(p)]
    if constexpr (!last) {
      __syncthreads();
      const u32 base = (u - kp) * 8 + kp;
#define PM_ST(t, p)                                               \
  {                                                               \
    const u32 e = (base + (t) * nsp) * T + c;                     \
    lds0[e] = u32x4{x[p].l[0], x[p].l[1], x[p].l[2], x[p].l[3]};  \
    lds1[e] = u32x4{x[p].l[4], x[p].l[5], x[p].l[6], x[p].l[7]};  \
    lds2[e] = x[p].l[8];                                          \
  }
      PM_ST(0, 0) PM_ST(1, 4) PM_ST(2, 2) PM_ST(3, 6) PM_ST(4, 1) PM_ST(5, 5) PM_ST(6, 3) PM_ST(7, 7)
#undef PM_ST
      __syncthreads();
    } else {
      // X[t] belongs to row u + t*U: put it into x[t]  (swap 1<->4, 3<->6)
      Fr t1 = x[1];
      x[1] = x[4];
      x[4] = t1;
      Fr t3 = x[3];
      x[3] = x[6];
      x[6] = t3;
    }
  } else if constexpr (Q == 4) {
    // always the last step (the odd radix goes last): Ns' = R/4, k' = v
    const u32 v0 = u, v1 = u + U;
    x[0] = fe_norm<FrP>(x[0]);
    x[1] = fe_norm<FrP>(x[1]);
    x[2] = fe_mul<FrP>(x[2], ld_tw(stw, 0 * nsp + v0));
    x[4] = fe_mul<FrP>(x[4], ld_tw(stw, 1 * nsp + v0));
    x[6] = fe_mul<FrP>(x[6], ld_tw(stw, 2 * nsp + v0));
    x[3] = fe_mul<FrP>(x[3], ld_tw(stw, 0 * nsp + v1));
    x[5] = fe_mul<FrP>(x[5], ld_tw(stw, 1 * nsp + v1));
    x[7] = fe_mul<FrP>(x[7], ld_tw(stw, 2 * nsp + v1));
    dft4(x[0], x[2], x[4], x[6], w8_2);  // X[t] -> row v + t*R/4 = u + (i + 2t) U = x[i + 2t]
    dft4(x[1], x[3], x[5], x[7], w8_2);
  } else {  // Q == 2, last step, Ns' = R/2, k' = v = u + i U
    x[0] = fe_norm<FrP>(x[0]);
    x[1] = fe_norm<FrP>(x[1]);
    x[2] = fe_norm<FrP>(x[2]);
    x[3] = fe_norm<FrP>(x[3]);
    x[4] = fe_mul<FrP>(x[4], ld_tw(stw, u));
    x[5] = fe_mul<FrP>(x[5], ld_tw(stw, u + U));
    x[6] = fe_mul<FrP>(x[6], ld_tw(stw, u + 2 * U));
    x[7] = fe_mul<FrP>(x[7], ld_tw(stw, u + 3 * U));
    BFLY(3, x[0], x[4]);
    BFLY(3, x[1], x[5]);
    BFLY(3, x[2], x[6]);
    BFLY(3, x[3], x[7]);
  }
  if constexpr (last) {
    // -------------------------------------------------------------- store
    const size_t n = (size_t)1 << a.log_n;
    const size_t j = j0 + c;
    const size_t ns = (size_t)1 << a.log_ns;
    const size_t k = j & (ns - 1);
    const size_t obase = (j - k) * R + k;
    if constexpr (OUT_WIDE) {
      const WidePtr wout = wide_ptrs(a.out, a.wide_glog);
      const size_t boff = (size_t)blockIdx.y * n;
#pragma unroll
      for (int m = 0; m < 8; ++m) st_wide(wout, boff + obase + (size_t)(u + m * U) * ns, x[m]);
    } else {
      u32x4* gout = reinterpret_cast<u32x4*>(a.out) + 2 * (size_t)blockIdx.y * a.batch_stride_out;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
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
__global__ void __launch_bounds__((1 << (S + LT)) / 8 < 64 ? 64 : (1 << (S + LT)) / 8)
    ntt_pass_kernel(const NttPassArgs a, const NttConsts kc) {
  constexpr int R = 1 << S;
  constexpr int T = 1 << LT;
  constexpr int U = R / 8;  // threads per column
  constexpr int NTHREADS = U * T;
  constexpr int NSTEPS = num_steps(S);
  extern __shared__ u32x4 lds[];
  u32x4* lds0 = lds;                                         // limbs 0..3
  u32x4* lds1 = lds + R * T;                                 // limbs 4..7
  u32* lds2 = reinterpret_cast<u32*>(lds + 2 * R * T);       // limb 8

  const u32 tid = threadIdx.x;
  if (NTHREADS < 64 && tid >= NTHREADS) return;  // single-wave workgroup: barriers stay legal
  const u32 log_n = a.log_n;
  const size_t n = (size_t)1 << log_n;
  const size_t n_cols = (size_t)1 << (log_n - S);  // N / R
  const size_t j0 = (size_t)xcd_tile(blockIdx.x, gridDim.x, a.flags) * T;

  const Fr w8_1 = fr_limbs(kc.w8[0]);
  const Fr w8_2 = fr_limbs(kc.w8[1]);
  const Fr w8_3 = fr_limbs(kc.w8[2]);

  Fr x[8];
  // ---------------------------------------------------------------- load (step 0 input)
  {
    constexpr bool ufast = OUT_UFAST && NSTEPS == 1;
    const u32 c = ufast ? tid / U : tid & (T - 1);
    const u32 u = ufast ? tid % U : tid >> LT;
    const size_t j = j0 + c;
    if constexpr (IN_WIDE) {
      const WidePtr win = wide_ptrs(const_cast<void*>(a.in), a.wide_glog);
      const size_t boff = (size_t)blockIdx.y * n;
      const u32 k = (u32)(j & (((size_t)1 << a.log_ns) - 1));
      const u32 tw_shift = log_n - a.log_ns - S;  // exponent stride N / (Ns R)
      if (a.flags & PASS_DIRECT_TW) {
        const WidePtr wtw = wide_ptrs(const_cast<void*>(a.pass_tw), a.wide_glog);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const size_t idx = j + (size_t)(u + m * U) * n_cols;
          x[m] = fe_mul<FrP>(ld_wide(win, boff + idx), ld_wide(wtw, idx));     // (<6, <40) * (1, <1)
        }
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const u32 row = u + m * U;
          Fr v = ld_wide(win, boff + j + (size_t)row * n_cols);                // (<6, <40)
          x[m] = fe_mul<FrP>(v, fr_canon(two_level(a.tw_hi, a.tw_lo, (k * row) << tw_shift, a.lh)));
        }
      }
    } else {
      const u32x4* gin = reinterpret_cast<const u32x4*>(a.in) + 2 * (size_t)blockIdx.y * a.batch_stride_in;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const size_t gi = j + (size_t)(u + m * U) * n_cols;
        if (gi < a.in_len) {
          x[m] = fe_load<FrP>(gin + 2 * gi);                                   // (1, <1)
          if (a.flags & PASS_PRE_COSET)
            x[m] = fe_mul<FrP>(x[m], two_level(a.cs_hi, a.cs_lo, (u32)gi, a.lh));
        } else {
          x[m] = fe_zero<FrP>();
        }
      }
    }
  }
  // ---------------------------------------------------------------- in-tile steps
  ntt_step<S, LT, 0, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w8_1, w8_2, w8_3, tid, j0);
  if constexpr (NSTEPS > 1)
    ntt_step<S, LT, 1, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w8_1, w8_2, w8_3, tid, j0);
  if constexpr (NSTEPS > 2)
    ntt_step<S, LT, 2, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w8_1, w8_2, w8_3, tid, j0);
  if constexpr (NSTEPS > 3)
    ntt_step<S, LT, 3, OUT_UFAST, OUT_WIDE>(x, a, kc, lds0, lds1, lds2, w8_1, w8_2, w8_3, tid, j0);
}

// out[i] = mult * base^(i * stride)      (table builder; one thread per entry)
static __global__ void pow_table_kernel(u32x4* out, const NttConsts c, u32 count, u32 stride) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fr r = fr_pow(fr_limbs(c.w8[0]), (unsigned long long)i * stride, fr_limbs(c.scale));
  st_tw(out, i, fr_canon(r));
}

// step twiddles for radix 2^S: block s, entry [(t-1)*Ns' + k'] = w_{Ns' q}^(k' t) = wR^(k' t R/(Ns' q))
static __global__ void step_tw_kernel(u32x4* out, const NttConsts c, u32 S) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 off = 0;
  for (u32 s = 0; 3 * s < S; ++s) {
    const u32 lq = (S - 3 * s) >= 3 ? 3 : (S - 3 * s);
    const u32 nsp = 1u << (3 * s);
    const u32 cnt = ((1u << lq) - 1) * nsp;
    if (i >= off && i < off + cnt) {
      const u32 t = (i - off) / nsp + 1, kp = (i - off) % nsp;
      const u32 e = (kp * t) << (S - 3 * s - lq);
      st_tw(out, i, fr_canon(fr_pow(fr_limbs(c.w8[0]), e, fr_limbs(c.one))));
      return;
    }
    off += cnt;
  }
}

// twiddles of one pass in element order: out[idx] = mult * w_N^((j mod Ns) * row * N/(Ns R)),
// idx = j + row * N/R  (exactly the index the pass loads its input element with)
static __global__ void pass_tw_kernel(void* out, const NttConsts c, u32 log_n, u32 log_ns, u32 S, const u32x4* tw_hi,
                               const u32x4* tw_lo, u32 lh, u32 glog) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)1 << log_n;
  if (idx >= n) return;
  const u32 j = (u32)(idx & ((n >> S) - 1)), row = (u32)(idx >> (log_n - S));
  const u32 k = j & ((1u << log_ns) - 1);
  const u32 e = (k * row) << (log_n - log_ns - S);
  Fr v = fe_mul<FrP>(two_level(tw_hi, tw_lo, e, lh), fr_limbs(c.scale));
  st_wide(wide_ptrs(out, glog), idx, fr_canon(v));
}

// log_n < 3: direct evaluation, one thread per output
static __global__ void ntt_tiny_kernel(const NttPassArgs a, const NttConsts kc) {
  const u32 n = 1u << a.log_n;
  const u32 j = threadIdx.x;
  const u32x4* gin = reinterpret_cast<const u32x4*>(a.in) + 2 * (size_t)blockIdx.y * a.batch_stride_in;
  u32x4* gout = reinterpret_cast<u32x4*>(a.out) + 2 * (size_t)blockIdx.y * a.batch_stride_out;
  Fr acc = fe_zero<FrP>();
  if (j < n) {
    for (u32 i = 0; i < n && i < a.in_len; ++i) {
      Fr v = fe_load<FrP>(gin + 2 * i);
      if (a.flags & PASS_PRE_COSET) v = fe_mul<FrP>(v, two_level(a.cs_hi, a.cs_lo, i, a.lh));
      v = fe_mul<FrP>(v, two_level(a.tw_hi, a.tw_lo, (i * j) & (n - 1), a.lh));  // (1, <2)
      acc = fe_add<FrP>(acc, v);                                                  // <= (4, 8)
    }
  }
  __syncthreads();  // in-place safe: every read precedes every write (one block per vector)
  if (j < n) {
    acc = fe_mul<FrP>(acc, fr_limbs(kc.scale));  // n^-1 (inverse) or one (forward): also the reduction
    if (a.flags & PASS_POST_COSET) acc = fe_mul<FrP>(acc, two_level(a.cs_hi, a.cs_lo, j, a.lh));
    fe_store<FrP>(gout + 2 * j, acc);
  }
}

}  // namespace pm
