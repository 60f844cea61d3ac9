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
#include <cstdlib>
#include <cstring>
#include <vector>

#include "context.h"
#include "field_inv.hip.h"
#include "host_field.h"
#include "ntt_kernels.hip.h"
#include "poly_common.hip.h"

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
// the two power tables of one evaluation in ONE launch (two launches of ~12 us each were a third of a small circuit's
// opening round): xpow[t] = x^t for t < 256 and xblk[b] = x^(b SEG)
__global__ void __launch_bounds__(256) eval_tables_kernel(u32x4* xpow, u32x4* xblk, const NttConsts c, u32 nblocks, u32 seg) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 256u) {
    st_tw(xpow, i, fr_canon(fr_pow(fr_limbs(c.w8[0]), i, fr_limbs(c.scale))));
  } else if (i - 256u < nblocks) {
    const u32 b = i - 256u;
    st_tw(xblk, b, fr_canon(fr_pow(fr_limbs(c.w8[0]), (unsigned long long)b * seg, fr_limbs(c.scale))));
  }
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

// ------------------------------------------------------------------ chunked scans (prefix product, Ruffini)
// Both are first-order recurrences y_k = op(a_k, y_{k-1}) over the whole vector:
//   prefix product   out_k = y_{k-1},  y_k = y_{k-1} a_k           (the grand product z of the permutation argument)
//   Ruffini          y_k = d_k + z y_{k-1},  out = y_k             (q_{i-1} = c_i + z q_i read from the top coefficient down)
// and both are computed the same way (r02; replaces the LDS-staged three-kernel scans of r01, which moved
// 160 B and ~5 products per element):
//   totals  every thread runs the recurrence over its SC_K consecutive elements from the neutral start and
//           keeps only the chunk's total: 1 product per element, n / SC_K values written
//   scan    the totals obey the SAME recurrence one level up (Ruffini: with z^SC_K in place of z): recurse
//           until one workgroup holds the level (<= SC_BASE entries) and scans it in place
//   replay  every thread runs the recurrence again from its chunk's carry and writes the outputs:
//           1 product per element
// ~2.2 products and 32 + 32 + 32 B (the second read mostly from the Infinity Cache) per element.
constexpr int SC_K = 8;             // elements per thread
constexpr int SC_BASE = 2048;       // largest level one workgroup (256 threads x SC_K) scans
// Level 0 moves whole tiles of 256 x SC_K canonical elements between HBM and the threads' consecutive
// chunks through LDS: coalesced 16-byte accesses on the HBM side (all of a tile's loads in flight at once),
// one padding slot per chunk on the LDS side (lane stride 17 x 16 B: conflict-free ds_read/write_b128).
PM_DEV Fr fr_shfl_up(const Fr& v, int d) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = __shfl_up(v.l[i], d);
  return r;
}
// ABI (canonical) -> device form without a product: x 2^5, then the one-limb reduction
PM_DEV Fr abi_to_dev(const Fr& x) { return fe_reduce_weak<FrP>(fr_shl5(x)); }

constexpr int SC_TILE = 256 * SC_K;
constexpr int SC_H = 2;                               // a tile passes through LDS in SC_H rounds of SC_K / SC_H elements per thread
constexpr int SC_R = SC_K / SC_H;                     // elements per thread and round
constexpr int SC_LDS_SLOTS = 256 * (2 * SC_R + 1);   // 16-byte slots: 36 KB -> four workgroups per CU
// `rev`: element e of the tile lives at in[first - e] instead of in[first + e] (Ruffini walks the coefficients down).
// Round h moves elements {chunk * SC_K + h * SC_R + r}: per wave 128-byte runs (whole cache lines) every 256 bytes.
PM_DEV void tile_load(const u32x4* in, long long first, bool rev, long long n_valid /* elements e < n_valid exist */,
                      u32x4* lds, u32 (&w)[SC_K][8]) {
  const u32 t = threadIdx.x;
#pragma unroll
  for (int h = 0; h < SC_H; ++h) {
    u32x4 v[2 * SC_R];
#pragma unroll
    for (int i = 0; i < 2 * SC_R; ++i) {
      const u32 g = i * 256 + t;                                  // piece index of this round, memory order (or its mirror)
      const u32 gm = rev ? (u32)(256 * 2 * SC_R - 1) - g : g;     // rev: the mirrored piece, so that addresses still ascend with g
      const u32 chunk = gm / (2 * SC_R), within = gm % (2 * SC_R);
      const u32 piece = rev ? (within ^ 1u) : within;             // 2 * element + half, inside the thread's chunk
      const u32 e = chunk * SC_K + h * SC_R + (piece >> 1), half = piece & 1u;
      v[i] = u32x4{0u, 0u, 0u, 0u};
      if ((long long)e < n_valid) v[i] = in[2 * (rev ? first - (long long)e : first + (long long)e) + half];
    }
    if (h > 0) __syncthreads();                                   // the previous round's reads are done
#pragma unroll
    for (int i = 0; i < 2 * SC_R; ++i) {
      const u32 g = i * 256 + t;
      const u32 gm = rev ? (u32)(256 * 2 * SC_R - 1) - g : g;
      const u32 chunk = gm / (2 * SC_R), within = gm % (2 * SC_R);
      const u32 piece = rev ? (within ^ 1u) : within;             // slot order inside a chunk: element-major, low half first
      lds[chunk * (2 * SC_R + 1) + piece] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SC_R; ++r) {
      const u32x4 a = lds[t * (2 * SC_R + 1) + 2 * r], b = lds[t * (2 * SC_R + 1) + 2 * r + 1];
      const int k = h * SC_R + r;
      w[k][0] = a.x; w[k][1] = a.y; w[k][2] = a.z; w[k][3] = a.w;
      w[k][4] = b.x; w[k][5] = b.y; w[k][6] = b.z; w[k][7] = b.w;
    }
  }
  __syncthreads();
}
PM_DEV void tile_store(u32x4* out, long long first, bool rev, long long n_valid, u32x4* lds, const u32 (&w)[SC_K][8]) {
  const u32 t = threadIdx.x;
#pragma unroll
  for (int h = 0; h < SC_H; ++h) {
    if (h > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < SC_R; ++r) {
      const int k = h * SC_R + r;
      lds[t * (2 * SC_R + 1) + 2 * r] = u32x4{w[k][0], w[k][1], w[k][2], w[k][3]};
      lds[t * (2 * SC_R + 1) + 2 * r + 1] = u32x4{w[k][4], w[k][5], w[k][6], w[k][7]};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2 * SC_R; ++i) {
      const u32 g = i * 256 + t;
      const u32 gm = rev ? (u32)(256 * 2 * SC_R - 1) - g : g;
      const u32 chunk = gm / (2 * SC_R), within = gm % (2 * SC_R);
      const u32 piece = rev ? (within ^ 1u) : within;
      const u32 e = chunk * SC_K + h * SC_R + (piece >> 1), half = piece & 1u;
      if ((long long)e < n_valid) out[2 * (rev ? first - (long long)e : first + (long long)e) + half] = lds[chunk * (2 * SC_R + 1) + piece];
    }
  }
}

// ---- prefix product.  Level 0 reads canonical ABI elements tile-wise; inner levels hold device-form 48-byte
// entries (1 / SC_K of the data each) and use plain per-thread loops.
__global__ void __launch_bounds__(256) pp_totals0_kernel(const u32x4* in, size_t n, u32x4* tot) {
  extern __shared__ u32x4 sc_lds[];
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE;
  u32 w[SC_K][8];
  tile_load(in, (long long)tile0, false, (long long)(n - tile0), sc_lds, w);
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x, lo = chunk * SC_K;
  if (lo >= n) return;
  Fr acc = abi_to_dev(fe_unpack<FrP>(w[0]));
#pragma unroll
  for (int k = 1; k < SC_K; ++k)
    if (lo + k < n) acc = fe_mul<FrP>(acc, abi_to_dev(fe_unpack<FrP>(w[k])));
  st_tw(tot, chunk, acc);
}
// carry[chunk] = product of everything before the chunk (device form); out_e = that product in ABI form
__global__ void __launch_bounds__(256) pp_replay0_kernel(const u32x4* in, size_t n, const u32x4* carry, u32x4* out) {
  extern __shared__ u32x4 sc_lds[];
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE;
  u32 w[SC_K][8];
  tile_load(in, (long long)tile0, false, (long long)(n - tile0), sc_lds, w);
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x, lo = chunk * SC_K;
  if (lo < n) {
    Fr run = fe_mul<FrP>(ld_tw(carry, chunk), fe_pow2<FrP, 256>());   // device form -> ABI form; ABI x device stays ABI
#pragma unroll
    for (int k = 0; k < SC_K; ++k) {
      const Fr a = abi_to_dev(fe_unpack<FrP>(w[k]));
      fe_canon_pack<FrP>(w[k], run);
      run = fe_mul<FrP>(run, a);
    }
  }
  tile_store(out, (long long)tile0, false, (long long)(n - tile0), sc_lds, w);
}
__global__ void __launch_bounds__(256) pp_totals_kernel(const u32x4* in, size_t n, u32x4* tot, size_t nchunks) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks) return;
  const size_t lo = t * SC_K;
  Fr a[SC_K];
#pragma unroll
  for (int k = 0; k < SC_K; ++k) a[k] = lo + k < n ? ld_tw(in, lo + k) : fe_one<FrP>();   // all loads in flight at once
  Fr acc = a[0];
#pragma unroll
  for (int k = 1; k < SC_K; ++k) acc = fe_mul<FrP>(acc, a[k]);
  st_tw(tot, t, acc);
}
__global__ void __launch_bounds__(256) pp_replay_kernel(const u32x4* in, size_t n, const u32x4* carry, size_t nchunks,
                                                         u32x4* out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks) return;
  const size_t lo = t * SC_K;
  Fr a[SC_K];
#pragma unroll
  for (int k = 0; k < SC_K; ++k) a[k] = lo + k < n ? ld_tw(in, lo + k) : fe_one<FrP>();   // read before the slots are overwritten
  Fr run = ld_tw(carry, t);
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (lo + k < n) st_tw(out, lo + k, run);
    run = fe_mul<FrP>(run, a[k]);
  }
}
// one workgroup, m <= SC_BASE device-form entries, exclusive product scan in place
__global__ void __launch_bounds__(256) pp_base_kernel(u32x4* v, u32 m) {
  __shared__ u32 sh[4 * 9];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const Fr one = fe_one<FrP>();
  const u32 lo = t * SC_K, hi = lo + SC_K < m ? lo + SC_K : m;
  Fr tot = one;
  for (u32 e = lo; e < hi; ++e) tot = fe_mul<FrP>(tot, ld_tw(v, e));
  Fr incl = tot;
  for (int d = 1; d < 64; d <<= 1) {
    const Fr o = fr_shfl_up(incl, d);
    if (lane >= (u32)d) incl = fe_mul<FrP>(incl, o);
  }
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = incl.l[i];
  }
  Fr run = fr_shfl_up(incl, 1);
  if (lane == 0) run = one;
  __syncthreads();
  for (u32 w = 0; w < wave; ++w) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[w * 9 + i];
    run = fe_mul<FrP>(run, o);
  }
  for (u32 e = lo; e < hi; ++e) {
    const Fr a = ld_tw(v, e);
    st_tw(v, e, run);
    run = fe_mul<FrP>(run, a);
  }
}

// ---- prefix product in ONE pass (r03): decoupled look-back.  One read and one write of the vector instead of two reads
// and a write plus the launches of the inner levels.  Tiles of SC_TILE elements are handed out by an
// atomic ticket (a workgroup only ever waits for tiles whose workgroups have started, so they are resident and finish);
// a tile publishes the product of its elements (status 1) as soon as it has it, then wave 0 walks back over its
// predecessors' published values, 64 tiles per round, multiplying aggregates until it meets a tile whose INCLUSIVE
// product is out (status 2), publishes its own inclusive product and replays its elements from the carry.
// Published values travel through agent-scope atomic stores / loads (the XCDs' L2s are not coherent with each other)
// and carry their own validity (pp_publish / pp_fetch below) -- no status word, no cache write-back (a fence per tile
// costs as much as the tile).  Layout of the control block: [ticket | pad to 64 B | 48 B x tiles records]; zeroed
// before every call.
// A tile's record: ten words (five 64-bit stores / loads), each a 29-bit limb (the tenth: zero) with a 2-bit tag above it (1: the zero-start value, 2: the inclusive
// value; the record is overwritten once, 1 -> 2).  Every word is written and read atomically and validates itself: a
// reader takes the record when all ten tags agree, whichever of the two values that is, and reads again otherwise --
// one round trip per look-back round, no status word, no ordering between the stores.
PM_DEV void pp_publish(u32* slot, const Fr& v, u32 tag) {   // five 64-bit stores: two tagged words each
  unsigned long long* s64 = reinterpret_cast<unsigned long long*>(slot);
  const u32 tg = tag << 30;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const u32 lo = v.l[2 * i] | tg, hi = (2 * i + 1 < 9 ? v.l[2 * i + 1] : 0u) | tg;
    __hip_atomic_store(s64 + i, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
PM_DEV u32 pp_fetch(const u32* slot, Fr& v) {   // -> the tag, 0 = not there yet (or caught between the two values)
  const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(slot);
  u32 w[10];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const unsigned long long x = __hip_atomic_load(s64 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w[2 * i] = (u32)x;
    w[2 * i + 1] = (u32)(x >> 32);
  }
  u32 lo = w[0] >> 30, hi = lo;
#pragma unroll
  for (int i = 1; i < 10; ++i) {
    lo = lo < (w[i] >> 30) ? lo : (w[i] >> 30);
    hi = hi > (w[i] >> 30) ? hi : (w[i] >> 30);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) v.l[i] = w[i] & 0x3fffffffu;
  return lo == hi ? lo : 0u;
}
constexpr u32 PP_AHEAD = 4;   // look-back rounds whose records are requested together
// TW: the vector holds device-form 48-byte entries (an inner level of the chunked scan: the chunk totals) instead of canonical
// elements, read and written per thread (no LDS staging), and the exclusive products go back in device form -- in place.
template <bool TW>
__global__ void __launch_bounds__(256) pp_lookback_kernel(const u32x4* in, size_t n, u32x4* out, u32* ctl, u32 tiles) {
  extern __shared__ u32x4 sc_lds[];
  __shared__ u32 sh[4 * 9 + 9];
  __shared__ u32 s_tile;
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  u32* rec = ctl + 16;                    // 12 words per tile
  if (t == 0) s_tile = atomicAdd(ctl, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  if (tile >= tiles) return;
  const size_t tile0 = (size_t)tile * SC_TILE;
  u32 w[SC_K][8];
  if (!TW) tile_load(in, (long long)tile0, false, (long long)(n - tile0), sc_lds, w);
  const size_t lo = tile0 + (size_t)t * SC_K;
  const Fr one = fe_one<FrP>();
  // 1 the thread's product, then the inclusive scan over the workgroup's threads (as pp_base_kernel)
  Fr a[SC_K];
  Fr tot = one;
  if (TW) {
#pragma unroll
    for (int k = 0; k < SC_K; ++k) a[k] = lo + k < n ? ld_tw(in, lo + k) : one;   // all loads in flight at once
  }
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (!TW) a[k] = lo + k < n ? abi_to_dev(fe_unpack<FrP>(w[k])) : one;
    tot = k == 0 ? a[0] : fe_mul<FrP>(tot, a[k]);
  }
  Fr inc = tot;
  for (int d = 1; d < 64; d <<= 1) {
    const Fr o = fr_shfl_up(inc, d);
    if (lane >= (u32)d) inc = fe_mul<FrP>(inc, o);
  }
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = inc.l[i];
  }
  Fr excl = fr_shfl_up(inc, 1);          // product of the earlier threads of this wave
  if (lane == 0) excl = one;
  __syncthreads();
  Fr tile_total = one;
#pragma unroll
  for (u32 wv = 0; wv < 4; ++wv) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[wv * 9 + i];
    if (wv < wave) excl = fe_mul<FrP>(excl, o);
    tile_total = wv == 0 ? o : fe_mul<FrP>(tile_total, o);
  }
  // 2 publish, look back (wave 0), publish the inclusive product
  if (wave == 0) {
    Fr carry = one;
    if (tile > 0) {
      if (lane == 0) pp_publish(rec + 12 * (size_t)tile, tile_total, 1u);
      // every lane multiplies what it fetches into its OWN running product (one product per round); the product over
      // the lanes is taken once, after the last round (the tiles in flight are all in the same phase, so the walk goes
      // back over most of them: ~16 rounds with 1024 resident tiles)
      Fr mine = one;
      bool finished = false;
      for (u32 back = 1; !finished; back += 64 * PP_AHEAD) {
        // the records of PP_AHEAD rounds are requested together (one round trip instead of PP_AHEAD), then taken in order
        bool valid[PP_AHEAD];
        u32 st[PP_AHEAD];
        Fr got[PP_AHEAD];
#pragma unroll
        for (u32 u = 0; u < PP_AHEAD; ++u) {
          valid[u] = tile >= back + 64 * u + lane;    // predecessor tile - back - 64 u - lane exists
          st[u] = valid[u] ? pp_fetch(rec + 12 * (size_t)(tile - back - 64 * u - lane), got[u]) : 2u;   // beyond tile 0: "inclusive product = one"
          if (!valid[u]) got[u] = one;
        }
#pragma unroll
        for (u32 u = 0; u < PP_AHEAD; ++u) {
          if (finished) break;
          while (valid[u] && st[u] == 0u) {
            __builtin_amdgcn_s_sleep(1);
            st[u] = pp_fetch(rec + 12 * (size_t)(tile - back - 64 * u - lane), got[u]);
          }
          const u64 done = __ballot(st[u] == 2u);     // lanes that hold an inclusive product (or lie beyond the start)
          const u32 first = (u32)__ffsll((long long)done) - 1u;   // (ffs of 0 is 0: wraps to ~0 = none)
          if (valid[u] && (done == 0 || lane <= first)) mine = fe_mul<FrP>(mine, got[u]);
          finished = done != 0;
        }
      }
      // product over the lanes (order is irrelevant in a commutative group)
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        Fr o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.l[i] = __shfl_xor(mine.l[i], d);
        mine = fe_mul<FrP>(mine, o);
      }
      carry = mine;
    }
    if (lane == 0) {
      pp_publish(rec + 12 * (size_t)tile, fe_mul<FrP>(carry, tile_total), 2u);
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[36 + i] = carry.l[i];
    }
  }
  __syncthreads();
  // 3 replay from the carry: out_k = (everything before element k), ABI form
  Fr run;
#pragma unroll
  for (int i = 0; i < 9; ++i) run.l[i] = sh[36 + i];
  if (TW) {
    run = fe_mul<FrP>(run, excl);
#pragma unroll
    for (int k = 0; k < SC_K; ++k) {
      if (lo + k < n) st_tw(out, lo + k, run);
      run = fe_mul<FrP>(run, a[k]);
    }
    return;
  }
  run = fe_mul<FrP>(fe_mul<FrP>(run, excl), fe_pow2<FrP, 256>());   // device form -> ABI form; ABI x device stays ABI
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    fe_canon_pack<FrP>(w[k], run);
    run = fe_mul<FrP>(run, a[k]);
  }
  tile_store(out, (long long)tile0, false, (long long)(n - tile0), sc_lds, w);
}

// ---- Ruffini: d_k = c_{n-1-k}, k < m = n - 1 (the coefficients read top down), y_k = d_k + z y_{k-1}, out[m-1-k] = y_k.
// y = d + z y: ABI + device x ABI stays ABI form.
// ---- Ruffini as a SCALED PREFIX SUM (r06).  y_k = d_k + z y_{k-1} has a constant multiplier, so the scan over the threads
// needs no products: with c_g the value thread g's chunk of SC_K elements reaches from a zero start, the true value at the
// end of chunk g is  Y_g = sum_{h <= g} c_h z^(K (g - h)) = z^(K g) S_g,  S_g = sum_{h <= g} v_h,  v_h = c_h z^(-K h):
// a plain prefix SUM of the scaled chunk values.  Per thread: K products for c_g, two for v_g (the power from a two-level
// table: z^(-TILE tile) x z^(-K t)), ADDITIONS for the wave scan, the tile total and the decoupled look-back over the
// tiles, two products for the carry Y_{g-1} = z^(K (g - 1)) S_{g-1}, K products to replay: 2 K + 4 = 20 products per 8
// elements (2.5 per element), none of them on a chain longer than the thread's own -- against ~4.5 per element and 6 x 3
// dependent products in the wave scan plus a product per look-back round in ruf_lookback_kernel (r03 - r05), whose
// latency made 2^20 (the prover's size) cost 79 us.  Three stages (chunk values and in-tile sums | the tile sums scanned by
// one workgroup | carries and replay): a decoupled look-back over the tiles -- tried first, additions only -- still spends
// its time waiting on the in-flight predecessors' records (2^22: 245 us against 177 for the r05 hybrid).
// Forms as above: values ABI, multipliers device.  Tables at `tab` (48-byte device-form entries):
//   neg_lo[256]: z^(-K t) | pos_lo[256]: z^(K (t - 1)) | neg_hi[tiles]: z^(-TILE T) | pos_hi[tiles]: z^(TILE T)
struct RufSum {
  u32 z[9];        // z
  u32 one[9];
};
// base^(2^j), j < RUF_POW_BITS, of the four table bases (host side: 4 x 13 squarings), so that a table entry is the
// product of the powers its index has bits for -- at most 13 dependent products, ~6 on average, instead of the 26 of a
// square-and-multiply from the base (the table kernel is a latency chain in front of stage 1: 16 us -> 5 at 2^20)
constexpr int RUF_POW_BITS = 14;     // 2^14 tiles x 2048 = 2^25 elements per table; larger vectors fall back to fr_pow
struct RufPowers {
  u32 p[4][RUF_POW_BITS][9];   // [neg_lo | pos_lo | neg_hi | pos_hi]: z^-K, z^K, z^-TILE, z^TILE
  u32 pos_lo_scale[9];         // z^-K: pos_lo[t] = z^(K (t - 1))
};
__global__ void __launch_bounds__(256) ruf_sum_tables_kernel(u32x4* tab, const RufPowers c, const RufSum k, u32 tiles) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 512 + 2 * tiles) return;
  const u32 which = i < 256 ? 0u : i < 512 ? 1u : i < 512 + tiles ? 2u : 3u;
  u32 e = which == 0 ? i : which == 1 ? i - 256 : which == 2 ? i - 512 : i - 512 - tiles;
  Fr acc = which == 1 ? fr_limbs(c.pos_lo_scale) : fr_limbs(k.one);
  if (e >> RUF_POW_BITS) {   // beyond the precomputed squarings: finish from the top power
    acc = fr_pow(fr_limbs(c.p[which][RUF_POW_BITS - 1]), (unsigned long long)(e >> RUF_POW_BITS) << 1, acc);
    e &= (1u << RUF_POW_BITS) - 1u;
  }
#pragma unroll 1
  for (int j = 0; j < RUF_POW_BITS; ++j)
    if ((e >> j) & 1u) acc = fe_mul<FrP>(acc, fr_limbs(c.p[which][j]));
  st_tw(tab, i, fr_canon(acc));
}
PM_DEV Fr fr_wadd(const Fr& a, const Fr& b) { return fe_reduce_weak<FrP>(fe_add<FrP>(a, b)); }
// stage 1: per thread the scaled chunk value v_g; its exclusive sum over the earlier threads of the tile -> pre[g], the
// tile's sum -> tot[tile] (48-byte ABI-form entries)
__global__ void __launch_bounds__(256) ruf_sum_totals_kernel(const u32x4* coeffs, size_t n_coeffs, size_t m, u32x4* pre, u32x4* tot,
                                                              const RufSum lk, const u32x4* tab) {
  extern __shared__ u32x4 sc_lds[];
  __shared__ u32 sh[4 * 9];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6, tile = blockIdx.x;
  const u32x4 *neg_lo = tab, *neg_hi = tab + 3 * 512;
  const size_t tile0 = (size_t)tile * SC_TILE;
  u32 w[SC_K][8];
  tile_load(coeffs, (long long)(n_coeffs - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
  const Fr nlo = ld_tw(neg_lo, t), nhi = ld_tw(neg_hi, tile);   // requested before the recurrence, used after it
  const size_t lo = tile0 + (size_t)t * SC_K;
  const Fr z = fr_limbs(lk.z), zero = fe_zero<FrP>();
  // the thread's chunk from a zero start (a short last chunk is padded at the end with zeros: y -> z y; nothing follows it)
  Fr y = zero;
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    const Fr d = lo + k < m ? fe_unpack<FrP>(w[k]) : zero;
    y = k == 0 ? d : fr_wadd(fe_mul<FrP>(y, z), d);
  }
  Fr inc = fe_mul<FrP>(y, fe_mul<FrP>(nhi, nlo));                // v_g = c_g z^(-K g)
  for (int dd = 1; dd < 64; dd <<= 1) {
    const Fr o = fr_shfl_up(inc, dd);
    if (lane >= (u32)dd) inc = fr_wadd(inc, o);
  }
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = inc.l[i];
  }
  Fr excl = fr_shfl_up(inc, 1);          // sum over the earlier threads of this wave
  if (lane == 0) excl = zero;
  __syncthreads();
  Fr tile_total = zero;
#pragma unroll
  for (u32 wv = 0; wv < 4; ++wv) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[wv * 9 + i];
    if (wv < wave) excl = fr_wadd(excl, o);
    tile_total = wv == 0 ? o : fr_wadd(tile_total, o);
  }
  st_tw(pre, (size_t)tile * 256 + t, excl);
  if (t == 0) st_tw(tot, tile, tile_total);
}
// stage 2: tot[T] <- sum of tot[0 .. T) (one workgroup; additions only; 2048 entries per sweep, a thread's eight loads in
// flight together)
__global__ void __launch_bounds__(256) ruf_sum_carry_kernel(u32x4* tot, u32 tiles) {
  __shared__ u32 sh[5 * 9];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const Fr zero = fe_zero<FrP>();
  Fr carry = zero;                                     // sum of the sweeps before this one
  for (u32 base = 0; base < tiles; base += 2048) {
    const u32 lo = base + t * 8;
    Fr a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = lo + k < tiles ? ld_tw(tot, lo + k) : zero;
    Fr sum = a[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) sum = fr_wadd(sum, a[k]);
    Fr inc = sum;
    for (int dd = 1; dd < 64; dd <<= 1) {
      const Fr o = fr_shfl_up(inc, dd);
      if (lane >= (u32)dd) inc = fr_wadd(inc, o);
    }
    if (base) __syncthreads();                         // the previous sweep's reads of sh are done
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = inc.l[i];
    }
    Fr run = fr_shfl_up(inc, 1);
    if (lane == 0) run = zero;
    run = fr_wadd(run, carry);
    __syncthreads();
    Fr all = zero;
#pragma unroll
    for (u32 wv = 0; wv < 4; ++wv) {
      Fr o;
#pragma unroll
      for (int i = 0; i < 9; ++i) o.l[i] = sh[wv * 9 + i];
      if (wv < wave) run = fr_wadd(run, o);
      all = fr_wadd(all, o);
    }
    carry = fr_wadd(carry, all);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (lo + k < tiles) st_tw(tot, lo + k, run);
      run = fr_wadd(run, a[k]);
    }
  }
}
// stage 3: the carry into the thread's chunk, Y_{g-1} = z^(K (g - 1)) S_{g-1}, S_{g-1} = (sum of the tiles before) + pre[g],
// and the replay.  SUM_TILES: stage 2 did not run (at most 512 tiles -- up to 2^20 elements: a launch costs more than the
// sums) -- the workgroup adds up tot[0 .. tile) itself, loads in flight together, one reduction.
template <bool SUM_TILES>
__global__ void __launch_bounds__(256) ruf_sum_replay_kernel(const u32x4* coeffs, size_t n_coeffs, size_t m, const u32x4* pre,
                                                              const u32x4* tot, u32x4* out, const RufSum lk, const u32x4* tab, u32 tiles) {
  extern __shared__ u32x4 sc_lds[];
  __shared__ u32 sh[4 * 9];
  const u32 t = threadIdx.x, tile = blockIdx.x;
  const u32x4 *pos_lo = tab + 3 * 256, *pos_hi = tab + 3 * (512 + (size_t)tiles);
  const size_t tile0 = (size_t)tile * SC_TILE;
  const Fr zero = fe_zero<FrP>();
  Fr before;
  if (SUM_TILES) {
    Fr a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (u32)k * 256 + t < tile ? ld_tw(tot, (size_t)k * 256 + t) : zero;
    Fr sum = a[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) sum = fr_wadd(sum, a[k]);
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
      Fr o;
#pragma unroll
      for (int i = 0; i < 9; ++i) o.l[i] = __shfl_xor(sum.l[i], dd);
      sum = fr_wadd(sum, o);
    }
    if ((t & 63u) == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[(t >> 6) * 9 + i] = sum.l[i];
    }
  } else {
    before = ld_tw(tot, tile);
  }
  const Fr mine = ld_tw(pre, (size_t)tile * 256 + t);
  const Fr pw = fe_mul<FrP>(ld_tw(pos_hi, tile), ld_tw(pos_lo, t));
  u32 w[SC_K][8];
  tile_load(coeffs, (long long)(n_coeffs - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);   // (its barriers publish sh)
  if (SUM_TILES) {
    before = zero;
#pragma unroll
    for (u32 wv = 0; wv < 4; ++wv) {
      Fr o;
#pragma unroll
      for (int i = 0; i < 9; ++i) o.l[i] = sh[wv * 9 + i];
      before = wv == 0 ? o : fr_wadd(before, o);
    }
  }
  const size_t lo = tile0 + (size_t)t * SC_K;
  const Fr z = fr_limbs(lk.z);
  Fr y = fe_mul<FrP>(fr_wadd(before, mine), pw);
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (lo + k < m) {
      y = fr_wadd(fe_mul<FrP>(y, z), fe_unpack<FrP>(w[k]));
      fe_canon_pack<FrP>(w[k], y);
    }
  }
  tile_store(out, (long long)(m - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
}

// z == 0: q_{i-1} = c_i
__global__ void ruffini_shift_kernel(const u32x4* coeffs, u32x4* out, size_t m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[2 * i] = coeffs[2 * (i + 1)];
  out[2 * i + 1] = coeffs[2 * (i + 1) + 1];
}

// ------------------------------------------------------------------ batch inversion
// Montgomery's trick per thread, blocked by QUADS (r06): thread t owns the elements t + k T (k < 4 Q; consecutive lanes,
// consecutive elements: coalesced) as Q quads {4 j .. 4 j + 3}.  Forward, per quad: the product E_j of its (non-zero)
// elements -- 3 products, a two-level tree -- and ONE 48-byte scratch entry, the running product before the quad.  One
// inversion per thread.  Backward, per quad: the elements are read again, 1 / E_j = inv x scratch_j, the four inverses
// come out of the quad's own two-level tree (7 products), inv moves on by E_j (2 products): 15 products per quad
// = 3.75 per element and 32 + 12 + 32 + 12 + 32 = 120 bytes per element.  (r01 - r05: one scratch entry per ELEMENT,
// 3 products and 192 bytes per element -- a streaming kernel at 3.1 TB/s that could not go faster.)
// Forms: u = a 2^5 (ABI -> device form, a shift and a weak reduction, no product); products in device form; the inverse
// of the thread's product is moved to ABI form once, after which inverse(ABI) x anything(device) stays in ABI form.
// The inversion is the binary GCD of field_inv.hip.h: 1291 instructions per round (ISA count, r06), at most 18 rounds,
// ~13 on random inputs (the wave leaves when every lane's a is zero) -- ~75 products' worth per thread, every lane of
// the wave inverting its own product; Q amortises it (host side: Q grows with n, as long as every SIMD keeps two waves).
struct BinvQuad {
  Fr e[4];       // device form, 1 where the element is zero or beyond the end
  bool live[4];  // a non-zero element inside the vector
};
PM_DEV BinvQuad binv_load(const u32x4* v, size_t n, size_t first, size_t T) {
  BinvQuad q;
  u32x4 lo[4], hi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {                       // the four loads in flight together
    const size_t i = first + (size_t)k * T;
    lo[k] = hi[k] = u32x4{0u, 0u, 0u, 0u};
    if (i < n) {
      lo[k] = v[2 * i];
      hi[k] = v[2 * i + 1];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 w[8] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, hi[k].z, hi[k].w};
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) nz |= w[i];
    q.live[k] = nz != 0;                             // (beyond the end: loaded as zero)
    q.e[k] = q.live[k] ? abi_to_dev(fe_unpack<FrP>(w)) : fe_one<FrP>();
  }
  return q;
}
// MUL: out_i = mul_i / v_i instead of 1 / v_i (the prover's num / den of the permutation argument: one more product
// per element here instead of a pass of its own over both vectors)
template <bool MUL>
__global__ void __launch_bounds__(256) batch_inverse_kernel(u32x4* v, const u32x4* mul, size_t n, u32 Q, u32x4* scratch) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fr acc = fe_one<FrP>();
  for (u32 j = 0; j < Q; ++j) {
    const size_t first = t + (size_t)(4 * j) * T;
    if (first >= n) break;
    const BinvQuad q = binv_load(v, n, first, T);
    st_tw(scratch, (size_t)j * T + t, acc);     // product of everything before the quad
    acc = fe_mul<FrP>(acc, fe_mul<FrP>(fe_mul<FrP>(q.e[0], q.e[1]), fe_mul<FrP>(q.e[2], q.e[3])));
  }
  // acc = x R' (device form): the integer inverse is x^-1 / R'; times R R'^2 (and the product's 1 / R') gives x^-1 R,
  // the ABI form the back-substitution wants
  Fr inv = fe_mul<FrP>(fe_inv_int<FrP>(fe_canon_limbs<FrP>(acc)), fe_pow2<FrP, 256 + 2 * 261>());
  for (u32 j = Q; j-- > 0;) {
    const size_t first = t + (size_t)(4 * j) * T;
    if (first >= n) continue;
    const BinvQuad q = binv_load(v, n, first, T);
    const Fr p01 = fe_mul<FrP>(q.e[0], q.e[1]), p23 = fe_mul<FrP>(q.e[2], q.e[3]);
    const Fr iq = fe_mul<FrP>(inv, ld_tw(scratch, (size_t)j * T + t));   // 1 / E_j, ABI form
    const Fr i01 = fe_mul<FrP>(iq, p23), i23 = fe_mul<FrP>(iq, p01);     // 1 / (e0 e1), 1 / (e2 e3)
    Fr o[4] = {fe_mul<FrP>(i01, q.e[1]), fe_mul<FrP>(i01, q.e[0]), fe_mul<FrP>(i23, q.e[3]), fe_mul<FrP>(i23, q.e[2])};
    if (MUL) {
      Fr f[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = q.live[k] ? abi_to_dev(ld_canon(mul, first + (size_t)k * T)) : fe_one<FrP>();
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fe_mul<FrP>(o[k], f[k]);        // ABI x device stays ABI
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (q.live[k]) st_canon(v, first + (size_t)k * T, o[k]);           // zero stays zero; nothing beyond the end is live
    inv = fe_mul<FrP>(inv, fe_mul<FrP>(p01, p23));
  }
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

// One evaluation batch, enqueued on `st` without a host synchronisation: the tables, partial sums and results of batch
// number `slot` live in their own region of the context's polynomial workspace (`slot_bytes` apart), the results are copied
// to `out` by the stream.  The caller holds ctx->mu and the polynomial resource group and synchronises the stream.
// Coefficients per thread.  A workgroup ends in a 256-way sum through LDS (eight levels, two barriers each: ~480 instructions per
// thread) -- as much as TWO Horner steps, so the segment must be long enough to amortise it, and a batch of k polynomials
// brings k times the workgroups: L grows with k (r06: the prover's 15-polynomial opening at 2^20 ran L = 4, where the sum was
// half of the kernel).  At least ~1024 workgroups for one polynomial, ~4096 in all for a batch.
static u32 eval_len_per_thread(uint32_t k, size_t n) {
  const size_t one = n / ((size_t)256 * 1024), many = n * (size_t)k / ((size_t)256 * 4096);
  return (u32)std::max<size_t>(1, std::min<size_t>(64, std::max(one, many)));
}
static size_t eval_slot_bytes(uint32_t k, size_t n) {
  const u32 L = eval_len_per_thread(k, n);
  const size_t seg = (size_t)256 * L;
  const size_t nblocks = (n + seg - 1) / seg;
  return ((256 + (1 + (size_t)k) * nblocks) * 48 + 32 * (size_t)k + 64 + 255) / 256 * 256;
}
static int eval_enqueue(pm_ctx* ctx, hipStream_t st, uint32_t k, const void* const* d_polys, size_t n, const uint64_t point[4],
                        uint64_t* out, char* ws) {
  EvalPolys polys;
  memset(&polys, 0, sizeof polys);
  for (uint32_t j = 0; j < k; ++j) {
    if (!d_polys[j]) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
    polys.p[j] = (const u32x4*)d_polys[j];
  }
  const u32 L = eval_len_per_thread(k, n);
  const size_t seg = (size_t)256 * L;
  const u32 nblocks = (u32)((n + seg - 1) / seg);
  HFr x;
  memcpy(x.l, point, 32);
  EvalConsts kc;
  to_limbs29(kc.x, x);
  to_limbs29(kc.xrow, hfr_pow_u64(x, 256));
  to_limbs29(kc.xseg, hfr_pow_u64(x, seg));
  to_limbs29(kc.one, host::one(host::FR()));
  u32x4* xpow = (u32x4*)ws;
  u32x4* xblk = xpow + 3 * 256;
  u32x4* partial = xblk + 3 * (size_t)nblocks;
  u32x4* d_out = partial + 3 * (size_t)nblocks * k;
  {
    NttConsts c;
    memset(&c, 0, sizeof c);
    memcpy(c.w8[0], kc.x, sizeof kc.x);
    memcpy(c.scale, kc.one, sizeof kc.one);
    hipLaunchKernelGGL(eval_tables_kernel, dim3((256 + nblocks + 255) / 256), dim3(256), 0, st, xpow, xblk, c, nblocks, (u32)seg);
  }
  {
    ProfScope prof(ctx, st, "fr_poly_evaluate");
    hipLaunchKernelGGL(poly_eval_kernel, dim3(nblocks, k), dim3(256), 0, st, polys, n, L, kc, (const u32x4*)xpow,
                       (const u32x4*)xblk, partial);
    hipLaunchKernelGGL(poly_eval_final_kernel, dim3(k), dim3(256), 0, st, (const u32x4*)partial, nblocks, d_out);
  }
  PM_HIP(ctx, hipGetLastError());
  PM_HIP(ctx, hipMemcpyAsync(out, d_out, 32 * (size_t)k, hipMemcpyDeviceToHost, st));
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
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (rc) return rc;
  rc = ensure_buffer(ctx, ctx->poly_ws, eval_slot_bytes(k, n));
  if (rc) return rc;
  rc = eval_enqueue(ctx, st, k, d_polys, n, point, out, (char*)ctx->poly_ws.ptr);
  if (rc) return rc;
  PM_HIP(ctx, hipStreamSynchronize(st));
  return PM_OK;
}
// Up to three batches at their own points (the prover's openings: two groups at z, one at z w) with ONE host
// synchronisation, on the context's stream.
int pm::poly_evaluate_groups(pm_ctx* ctx, uint32_t groups, const uint32_t* k, const void* const* const* polys,
                             const uint64_t* const* points, uint64_t* const* outs, size_t n) {
  if (!ctx || !k || !polys || !points || !outs || n == 0 || groups == 0 || groups > 3) return PM_ERR_BAD_ARG;
  for (uint32_t g = 0; g < groups; ++g)
    if (k[g] == 0 || k[g] > PM_LINCOMB_MAX || !polys[g] || !points[g] || !outs[g]) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (rc) return rc;
  size_t off[4] = {0, 0, 0, 0};
  for (uint32_t g = 0; g < groups; ++g) off[g + 1] = off[g] + eval_slot_bytes(k[g], n);
  rc = ensure_buffer(ctx, ctx->poly_ws, off[groups]);
  if (rc) return rc;
  // results through pinned memory: a copy to the caller's pageable arrays would block the host until the batch is done
  if (!ctx->poly_host_pinned) PM_HIP(ctx, hipHostMalloc(&ctx->poly_host_pinned, 3 * PM_LINCOMB_MAX * 32, hipHostMallocDefault));
  uint64_t* h = (uint64_t*)ctx->poly_host_pinned;
  for (uint32_t g = 0; g < groups && !rc; ++g)
    rc = eval_enqueue(ctx, st, k[g], polys[g], n, points[g], h + 4 * PM_LINCOMB_MAX * g, (char*)ctx->poly_ws.ptr + off[g]);
  if (rc) return rc;
  PM_HIP(ctx, hipStreamSynchronize(st));
  for (uint32_t g = 0; g < groups; ++g) memcpy(outs[g], h + 4 * PM_LINCOMB_MAX * g, 32 * (size_t)k[g]);
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
  HFr zz;
  memcpy(zz.l, z, 32);
  if (host::is_zero(zz)) {
    hipLaunchKernelGGL(ruffini_shift_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
                       (const u32x4*)d_coeffs, (u32x4*)d_out, m);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
  }
  // the scaled prefix sum (ruf_sum_*), every size: tables | chunk values + in-tile sums | tile carries | replay
  const u32 tiles = (u32)((m + SC_TILE - 1) / SC_TILE);
  const size_t tab_entries = 512 + 2 * (size_t)tiles, pre_entries = (size_t)tiles * 256;
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, (tab_entries + pre_entries + tiles) * 48);
  if (rc) return rc;
  const host::Field<4>& F = host::FR();
  const HFr z_inv = host::inv(zz, F);
  RufSum rs;
  to_limbs29(rs.z, zz);
  to_limbs29(rs.one, host::one(F));
  RufPowers rp;
  {
    const HFr zk_inv = hfr_pow_u64(z_inv, SC_K);
    HFr b[4] = {zk_inv, hfr_pow_u64(zz, SC_K), hfr_pow_u64(z_inv, SC_TILE), hfr_pow_u64(zz, SC_TILE)};
    for (int w_ = 0; w_ < 4; ++w_)
      for (int j = 0; j < RUF_POW_BITS; ++j) {
        to_limbs29(rp.p[w_][j], b[w_]);
        b[w_] = host::mul(b[w_], b[w_], F);
      }
    to_limbs29(rp.pos_lo_scale, zk_inv);
  }
  const size_t lds = (size_t)SC_LDS_SLOTS * 16;
  for (const void* fn : {(const void*)ruf_sum_totals_kernel, (const void*)ruf_sum_replay_kernel<false>,
                         (const void*)ruf_sum_replay_kernel<true>})
    if (int lrc = raise_lds_limit(ctx, fn, lds)) return lrc;
  u32x4* tab = (u32x4*)ctx->poly_ws.ptr;
  u32x4* pre = tab + 3 * tab_entries;
  u32x4* tot = pre + 3 * pre_entries;
  ProfScope prof(ctx, st, "fr_poly_ruffini");
  hipLaunchKernelGGL(ruf_sum_tables_kernel, dim3((unsigned)((tab_entries + 255) / 256)), dim3(256), 0, st, tab, rp, rs, tiles);
  hipLaunchKernelGGL(ruf_sum_totals_kernel, dim3(tiles), dim3(256), lds, st, (const u32x4*)d_coeffs, n, m, pre, tot, rs,
                     (const u32x4*)tab);
  if (tiles > 512) {   // (the in-kernel sum of the tiles before costs ~1200 instructions per thread: worth a launch only while the tiles are few)
    hipLaunchKernelGGL(ruf_sum_carry_kernel, dim3(1), dim3(256), 0, st, tot, tiles);
    hipLaunchKernelGGL(ruf_sum_replay_kernel<false>, dim3(tiles), dim3(256), lds, st, (const u32x4*)d_coeffs, n, m, (const u32x4*)pre,
                       (const u32x4*)tot, (u32x4*)d_out, rs, (const u32x4*)tab, tiles);
  } else {
    hipLaunchKernelGGL(ruf_sum_replay_kernel<true>, dim3(tiles), dim3(256), lds, st, (const u32x4*)d_coeffs, n, m, (const u32x4*)pre,
                       (const u32x4*)tot, (u32x4*)d_out, rs, (const u32x4*)tab, tiles);
  }
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
  // One pass (pp_lookback_kernel) while every tile is resident at once (r05: up to two tiles per CU; r03 - r04: four): 2^16 51.8 -> 31.9 us,
  // 2^18 71.4 -> 35.6, 2^20 93.5 -> 61.7, 2^21 135 -> 109; with a second generation of tiles the walk back over the
  // ~1000 tiles in flight (all in the same phase: none has its inclusive product yet) is paid per generation and the
  // three-stage scan below is as fast (2^22: 195 vs 197 us) or faster (2^24: 708 vs 584 us) -- profiles/r03_poly_rows.txt.
  // opt_poly_lookback: 0 = never, 1 = by this rule, 2 = whenever there is more than one tile (tests).
  const size_t lookback_max = ctx->opt_poly_lookback == 2 ? ((size_t)1 << 31) : (size_t)SC_TILE * 2 * (size_t)ctx->num_cus;
  if (ctx->opt_poly_lookback && n > (size_t)SC_TILE && n <= lookback_max) {
    // ticket + status + two 48-byte values per tile, zeroed per call
    const u32 tiles = (u32)((n + SC_TILE - 1) / SC_TILE);
    const size_t ctl_bytes = 64 + (size_t)tiles * 48;
    OrderScope order_scope(ctx, ctx->ord_poly, st);
    int rc = order_scope.rc;
    if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, ctl_bytes);
    if (rc) return rc;
    const size_t lds = (size_t)SC_LDS_SLOTS * 16;
    const void* fn = (const void*)pp_lookback_kernel<false>;
    if (int lrc = raise_lds_limit(ctx, fn, lds)) return lrc;
    ProfScope prof(ctx, st, "fr_prefix_product");
    PM_HIP(ctx, hipMemsetAsync(ctx->poly_ws.ptr, 0, ctl_bytes, st));
    hipLaunchKernelGGL(pp_lookback_kernel<false>, dim3(tiles), dim3(256), lds, st, (const u32x4*)d_in, n, (u32x4*)d_out,
                       (u32*)ctx->poly_ws.ptr, tiles);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
  }
  // r05: the chunked scan above 2^21 keeps its two level-0 passes (1 product per element each) but stops descending at the first
  // level whose tiles are all resident at once and scans THAT level with one look-back launch (pp_lookback_kernel<true>: the
  // regime where the one-pass form wins) instead of three more levels of totals / base / replay launches, each the latency of
  // 8 dependent products: 2^21 109 (one pass) -> 99 us, 2^22 196 -> 160 us (0.171 -> 0.210 of HBM), 2^23 337 -> 315, 2^24 593 -> 569
  // (profiles/r05_poly_rows.txt); the one-pass form keeps the sizes up to 2 tiles per CU (2^20: 62 us)
  const size_t tw_resident = (size_t)SC_TILE * 2 * (size_t)ctx->num_cus;   // two workgroups per CU at this kernel's register count
  std::vector<size_t> sz;
  sz.push_back(n);
  bool top_lookback = false;
  while (sz.back() > (size_t)SC_BASE) {
    sz.push_back((sz.back() + SC_K - 1) / SC_K);
    if (ctx->opt_poly_lookback && sz.back() > (size_t)SC_BASE && sz.back() <= tw_resident) {
      top_lookback = true;
      break;
    }
  }
  if (sz.size() == 1) sz.push_back((n + SC_K - 1) / SC_K);
  size_t tot_entries = 0;
  for (size_t i = 1; i < sz.size(); ++i) tot_entries += sz[i];
  const u32 top_tiles = top_lookback ? (u32)((sz.back() + SC_TILE - 1) / SC_TILE) : 0u;
  const size_t lvl_bytes = (tot_entries * 48 + 64 + 255) / 256 * 256, ctl_bytes = top_lookback ? 64 + (size_t)top_tiles * 48 : 0;
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, lvl_bytes + ctl_bytes);
  if (rc) return rc;
  std::vector<u32x4*> lvl(sz.size());
  {
    u32x4* p = (u32x4*)ctx->poly_ws.ptr;
    for (size_t i = 1; i < sz.size(); ++i) {
      lvl[i] = p;
      p += 3 * sz[i];
    }
  }
  const size_t sc_lds_bytes = (size_t)SC_LDS_SLOTS * 16;
  for (const void* fn : {(const void*)pp_totals0_kernel, (const void*)pp_replay0_kernel})
    if (int lrc = raise_lds_limit(ctx, fn, sc_lds_bytes)) return lrc;
  ProfScope prof(ctx, st, "fr_prefix_product");
  const size_t last = sz.size() - 1;
  for (size_t i = 0; i < last; ++i) {
    const unsigned blocks = (unsigned)((sz[i + 1] + 255) / 256);
    if (i == 0)
      hipLaunchKernelGGL(pp_totals0_kernel, dim3((unsigned)((n + SC_TILE - 1) / SC_TILE)), dim3(256), sc_lds_bytes, st,
                         (const u32x4*)d_in, n, lvl[1]);
    else
      hipLaunchKernelGGL(pp_totals_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4*)lvl[i], sz[i], lvl[i + 1],
                         sz[i + 1]);
  }
  if (top_lookback) {
    u32* ctl = (u32*)((char*)ctx->poly_ws.ptr + lvl_bytes);
    PM_HIP(ctx, hipMemsetAsync(ctl, 0, ctl_bytes, st));
    hipLaunchKernelGGL(pp_lookback_kernel<true>, dim3(top_tiles), dim3(256), 0, st, (const u32x4*)lvl[last], sz[last], lvl[last], ctl,
                       top_tiles);
  } else {
    hipLaunchKernelGGL(pp_base_kernel, dim3(1), dim3(256), 0, st, lvl[last], (u32)sz[last]);
  }
  for (size_t i = last; i-- > 0;) {
    const unsigned blocks = (unsigned)((sz[i + 1] + 255) / 256);
    if (i == 0)
      hipLaunchKernelGGL(pp_replay0_kernel, dim3((unsigned)((n + SC_TILE - 1) / SC_TILE)), dim3(256), sc_lds_bytes, st,
                         (const u32x4*)d_in, n, (const u32x4*)lvl[1], (u32x4*)d_out);
    else
      hipLaunchKernelGGL(pp_replay_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4*)lvl[i], sz[i],
                         (const u32x4*)lvl[i + 1], sz[i + 1], lvl[i]);
  }
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

extern "C" int pm_fr_batch_inverse_dev(pm_ctx* ctx, void* d_inout, size_t n, void* hip_stream) {
  return pm::fr_batch_inverse_mul(ctx, d_inout, nullptr, n, hip_stream);
}
// d_inout[i] <- d_mul[i] / d_inout[i] (d_mul == nullptr: 1 / d_inout[i]); zeros of d_inout stay zero
int pm::fr_batch_inverse_mul(pm_ctx* ctx, void* d_inout, const void* d_mul, size_t n, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  if (!d_inout) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  // Quads per thread: the inversion (~75 products' worth of instructions, every lane inverting its own product) is amortised
  // over 4 Q elements, against threads to fill the chip.  Measured (profiles/r06_binv_quads.txt, us per call):
  //   2^20: Q = 2: 105, 4: 86, 8: 120      2^22: Q = 4: 267, 8: 212, 16: 204, 32: 332      2^24: Q = 16: 695, 32: 672, 64: 715
  // i.e. one wave per SIMD (num_cus x 256 threads) up to Q = 16, Q = 32 beyond: amortising the inversion is worth more than
  // the second wave's issue slots.  Option binv_quads overrides (tools/binv_bench.py).
  size_t quads = (n + 3) / 4;
  size_t Qw = ctx->opt_binv_quads > 0 ? (size_t)ctx->opt_binv_quads
                                      : std::min<size_t>(32, std::max<size_t>(1, quads / ((size_t)ctx->num_cus * 256)));
  const size_t want_threads = (quads + Qw - 1) / Qw;
  const unsigned blocks = (unsigned)((want_threads + 255) / 256);
  const size_t T = (size_t)blocks * 256;
  const u32 Q = (u32)((n + 4 * T - 1) / (4 * T));
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, (size_t)Q * T * 48);     // one entry per quad
  if (rc) return rc;
  ProfScope prof(ctx, st, "fr_batch_inverse");
  if (d_mul)
    hipLaunchKernelGGL(batch_inverse_kernel<true>, dim3(blocks), dim3(256), 0, st, (u32x4*)d_inout, (const u32x4*)d_mul, n, Q,
                       (u32x4*)ctx->poly_ws.ptr);
  else
    hipLaunchKernelGGL(batch_inverse_kernel<false>, dim3(blocks), dim3(256), 0, st, (u32x4*)d_inout, (const u32x4*)nullptr, n, Q,
                       (u32x4*)ctx->poly_ws.ptr);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}
