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

// ---- Ruffini.  Level 0 reads the coefficients top down: d_k = c_{n-1-k}, k < m = n - 1, and writes
// out[m-1-k] = y_k; inner levels hold ABI-form 48-byte entries (lazily reduced) and are scanned in place.
// y = d + z y: ABI + device x ABI stays ABI form, value < 3 r, limbs < 2^30.
struct RufLevel {
  u32 z[9];       // this level's multiplier z^(SC_K^level), device form
};
// level 0: tile over k in [tile0, tile0 + SC_TILE), element k = coefficient n_coeffs - 1 - k (walked downwards)
__global__ void __launch_bounds__(256) ruf_totals0_kernel(const u32x4* coeffs, size_t n_coeffs, size_t m, u32x4* tot,
                                                           const RufLevel lv) {
  extern __shared__ u32x4 sc_lds[];
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE;
  u32 w[SC_K][8];
  tile_load(coeffs, (long long)(n_coeffs - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x, lo = chunk * SC_K;
  if (lo >= m) return;
  const Fr z = fr_limbs(lv.z);
  Fr y = fe_zero<FrP>();
  // a short last chunk is padded at the END with zeros: y -> z y, so that every chunk spans exactly SC_K steps
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    y = fe_mul<FrP>(y, z);
    if (lo + k < m) y = fe_add<FrP>(y, fe_unpack<FrP>(w[k]));
  }
  st_tw(tot, chunk, fe_reduce_weak<FrP>(y));
}
// carry[chunk] = y just before the chunk; out[m - 1 - k] = y_k, canonical
__global__ void __launch_bounds__(256) ruf_replay0_kernel(const u32x4* coeffs, size_t n_coeffs, size_t m, const u32x4* carry,
                                                           u32x4* out, const RufLevel lv) {
  extern __shared__ u32x4 sc_lds[];
  const size_t tile0 = (size_t)blockIdx.x * SC_TILE;
  u32 w[SC_K][8];
  tile_load(coeffs, (long long)(n_coeffs - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x, lo = chunk * SC_K;
  if (lo < m) {
    const Fr z = fr_limbs(lv.z);
    Fr y = ld_tw(carry, chunk);
#pragma unroll
    for (int k = 0; k < SC_K; ++k) {
      if (lo + k < m) {
        y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), fe_unpack<FrP>(w[k])));
        fe_canon_pack<FrP>(w[k], y);
      }
    }
  }
  tile_store(out, (long long)(m - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
}
// inner levels: ABI-form 48-byte entries, scanned in place
__global__ void __launch_bounds__(256) ruf_totals_kernel(const u32x4* in, size_t m, u32x4* tot, size_t nchunks,
                                                          const RufLevel lv) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks) return;
  const Fr z = fr_limbs(lv.z);
  const size_t lo = t * SC_K;
  Fr a[SC_K];
#pragma unroll
  for (int k = 0; k < SC_K; ++k) a[k] = lo + k < m ? ld_tw(in, lo + k) : fe_zero<FrP>();
  Fr y = fe_zero<FrP>();
#pragma unroll
  for (int k = 0; k < SC_K; ++k) y = fe_add<FrP>(fe_mul<FrP>(y, z), a[k]);
  st_tw(tot, t, fe_reduce_weak<FrP>(y));
}
// carry[t] = y just before chunk t of this level; every entry k is replaced by the carry INTO it (y_{k-1}): that is
// what the level below replays from
__global__ void __launch_bounds__(256) ruf_replay_kernel(u32x4* v, size_t m, const u32x4* carry, size_t nchunks,
                                                          const RufLevel lv) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks) return;
  const Fr z = fr_limbs(lv.z);
  const size_t lo = t * SC_K;
  Fr a[SC_K];
#pragma unroll
  for (int k = 0; k < SC_K; ++k) a[k] = lo + k < m ? ld_tw(v, lo + k) : fe_zero<FrP>();
  Fr y = ld_tw(carry, t);
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (lo + k < m) st_tw(v, lo + k, y);
    y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), a[k]));
  }
}
// one workgroup, m <= SC_BASE entries T_k (ABI form): in place T_k <- y_{k-1} (the carry INTO chunk k), where
// y_k = T_k + z y_{k-1}, y_{-1} = 0.
__global__ void __launch_bounds__(256) ruf_base_kernel(u32x4* v, u32 m, const RufLevel lv) {
  __shared__ u32 sh[4 * 9];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const Fr z = fr_limbs(lv.z);
  const u32 lo = t * SC_K;
  Fr y = fe_zero<FrP>();
  for (u32 k = lo; k < lo + SC_K; ++k) {     // padded like the totals: every thread spans SC_K steps
    y = fe_mul<FrP>(y, z);
    if (k < m) y = fe_add<FrP>(y, ld_tw(v, k));
  }
  y = fe_reduce_weak<FrP>(y);
  // inclusive scan over the threads: Y_t = y_t + z^(SC_K) Y_{t-1}; the multiplier of a step of d threads is z^(SC_K d)
  Fr zp = z;
  for (int i = 1; i < SC_K; i <<= 1) zp = fe_mul<FrP>(zp, zp);   // z^SC_K (SC_K is a power of two)
  Fr incl = y;
  for (int d = 1; d < 64; d <<= 1) {
    const Fr o = fr_shfl_up(incl, d);
    if (lane >= (u32)d) incl = fe_reduce_weak<FrP>(fe_add<FrP>(incl, fe_mul<FrP>(o, zp)));
    zp = fe_mul<FrP>(zp, zp);
  }
  // zp = z^(SC_K 64): one wave's span
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = incl.l[i];
  }
  Fr carry = fr_shfl_up(incl, 1);            // y before this thread, inside the wave
  if (lane == 0) carry = fe_zero<FrP>();
  __syncthreads();
  // what enters this wave: W = sum over earlier waves; then it reaches lane l multiplied by z^(SC_K l)
  Fr wcar = fe_zero<FrP>();
  for (u32 w = 0; w < wave; ++w) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[w * 9 + i];
    wcar = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(wcar, zp), o));
  }
  if (wave > 0) {
    Fr zl = fe_one<FrP>(), b = z;              // z^(SC_K lane) by square-and-multiply on (SC_K lane)
    for (u32 e = SC_K * lane; e; e >>= 1) {
      if (e & 1) zl = fe_mul<FrP>(zl, b);
      b = fe_mul<FrP>(b, b);
    }
    carry = fe_reduce_weak<FrP>(fe_add<FrP>(carry, fe_mul<FrP>(wcar, zl)));
  }
  y = carry;
  for (u32 k = lo; k < lo + SC_K && k < m; ++k) {
    const Fr a = ld_tw(v, k);
    st_tw(v, k, y);
    y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), a));
  }
}
// Ruffini in ONE pass (r03): the same decoupled look-back as pp_lookback_kernel, for y_k = d_k + z y_{k-1}.  A tile's
// published value is the y at its end computed from a zero start (status 1) or from the true carry (status 2); a value
// `b` tiles back enters this tile multiplied by (z^SC_TILE)^(b-1), so every lane of the walking wave keeps its own
// power of z^SC_TILE (one product per round) next to its running sum.  Forms as in the kernels above: values ABI,
// multipliers device.
struct RufLook {
  u32 z[9];        // z
  u32 zt[9];       // z^SC_TILE
  u32 zt64[9];     // z^(64 SC_TILE)
};
// TW (r05): the vector is an INNER level of the chunked scan -- ABI-form 48-byte entries T_k in forward order, multiplier z = this
// level's z^(SC_K^level) -- scanned in place: T_k <- y_{k-1}, the carry INTO chunk k (what ruf_base_kernel leaves).
template <bool TW>
__global__ void __launch_bounds__(256) ruf_lookback_kernel(const u32x4* coeffs, size_t n_coeffs, size_t m, u32x4* out,
                                                            u32* ctl, u32 tiles, const RufLook lk) {
  extern __shared__ u32x4 sc_lds[];
  __shared__ u32 sh[4 * 9 + 9 + 9];
  __shared__ u32 s_tile;
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  u32* rec = ctl + 16;                    // 12 words per tile
  if (t == 0) s_tile = atomicAdd(ctl, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  if (tile >= tiles) return;
  const size_t tile0 = (size_t)tile * SC_TILE;
  u32 w[SC_K][8];
  if (!TW) tile_load(coeffs, (long long)(n_coeffs - 1 - tile0), true, (long long)(m - tile0), sc_lds, w);
  const size_t lo = tile0 + (size_t)t * SC_K;
  const Fr z = fr_limbs(lk.z), one = fe_one<FrP>(), zero = fe_zero<FrP>();
  // 1 the thread's chunk from a zero start (a short last chunk is padded at the end with zeros: y -> z y)
  Fr d[SC_K];
  Fr y = zero;
  if (TW) {
#pragma unroll
    for (int k = 0; k < SC_K; ++k) d[k] = lo + k < m ? ld_tw(coeffs, lo + k) : zero;   // all loads in flight at once
  }
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (!TW) d[k] = lo + k < m ? fe_unpack<FrP>(w[k]) : zero;
    y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), d[k]));
  }
  // 2 inclusive scan over the threads of the wave: Y_t = y_t + z^SC_K Y_{t-1}; beside it the powers z^(SC_K (lane + 1))
  Fr zp = z;
  for (int i = 1; i < SC_K; i <<= 1) zp = fe_mul<FrP>(zp, zp);   // z^SC_K
  Fr inc = y, pw = zp;
  for (int dd = 1; dd < 64; dd <<= 1) {
    const Fr o = fr_shfl_up(inc, dd), po = fr_shfl_up(pw, dd);
    if (lane >= (u32)dd) {
      inc = fe_reduce_weak<FrP>(fe_add<FrP>(inc, fe_mul<FrP>(o, zp)));
      pw = fe_mul<FrP>(pw, po);
    }
    zp = fe_mul<FrP>(zp, zp);
  }
  // zp = z^(SC_K 64): one wave's span; pw = z^(SC_K (lane + 1))
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[wave * 9 + i] = inc.l[i];
  }
  Fr carry = fr_shfl_up(inc, 1), mult = fr_shfl_up(pw, 1);   // y before this thread inside the wave; z^(SC_K lane)
  if (lane == 0) {
    carry = zero;
    mult = one;
  }
  __syncthreads();
  // what enters this wave from the earlier waves of the tile (zero start), and the tile's total; the thread's
  // multiplier for whatever enters the TILE: z^(SC_K (64 wave + lane))
  Fr wcar = zero, tile_total = zero;
#pragma unroll
  for (u32 wv = 0; wv < 4; ++wv) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = sh[wv * 9 + i];
    if (wv < wave) {
      wcar = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(wcar, zp), o));
      mult = fe_mul<FrP>(mult, zp);
    }
    tile_total = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(tile_total, zp), o));
  }
  if (wave > 0) carry = fe_reduce_weak<FrP>(fe_add<FrP>(carry, fe_mul<FrP>(wcar, fr_shfl_up(pw, 1))));
  // (lane 0 of a later wave: fr_shfl_up leaves its own pw, but its in-wave multiplier is one: fix below)
  if (wave > 0 && lane == 0) carry = wcar;
  // 3 publish, look back (wave 0), publish the inclusive value
  if (wave == 0) {
    Fr C = zero;
    if (tile > 0) {
      if (lane == 0) pp_publish(rec + 12 * (size_t)tile, tile_total, 1u);
      // this lane's power: (z^SC_TILE)^(back - 1 + lane)
      const Fr zt = fr_limbs(lk.zt), zt64 = fr_limbs(lk.zt64);
      Fr lp = zt;
      for (int dd = 1; dd < 64; dd <<= 1) {
        const Fr po = fr_shfl_up(lp, dd);
        if (lane >= (u32)dd) lp = fe_mul<FrP>(lp, po);
      }
      lp = fr_shfl_up(lp, 1);                      // zt^lane
      if (lane == 0) lp = one;
      Fr mine = zero;
      for (u32 back = 1;; back += 64) {
        const bool valid = tile >= back + lane;
        const u32 pred = valid ? tile - back - lane : 0u;
        u32 stt = valid ? 0u : 2u;                  // beyond tile 0: "inclusive value = zero"
        Fr got = zero;
        while (valid && stt == 0u) {
          stt = pp_fetch(rec + 12 * (size_t)pred, got);
          if (stt == 0u) __builtin_amdgcn_s_sleep(1);
        }
        const u64 done = __ballot(stt == 2u);
        const u32 first = (u32)__ffsll((long long)done) - 1u;
        if (valid && (done == 0 || lane <= first)) mine = fe_reduce_weak<FrP>(fe_add<FrP>(mine, fe_mul<FrP>(got, lp)));
        if (done != 0) break;
        lp = fe_mul<FrP>(lp, zt64);
      }
#pragma unroll
      for (int dd = 32; dd > 0; dd >>= 1) {
        Fr o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.l[i] = __shfl_xor(mine.l[i], dd);
        mine = fe_reduce_weak<FrP>(fe_add<FrP>(mine, o));
      }
      C = mine;
    }
    if (lane == 0) {
      const Fr zt = fr_limbs(lk.zt);
      pp_publish(rec + 12 * (size_t)tile, fe_reduce_weak<FrP>(fe_add<FrP>(tile_total, fe_mul<FrP>(C, zt))), 2u);
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[36 + i] = C.l[i];
    }
  }
  __syncthreads();
  // 4 replay from y before the thread's chunk = (zero-start carry) + z^(SC_K thread) C
  Fr C;
#pragma unroll
  for (int i = 0; i < 9; ++i) C.l[i] = sh[36 + i];
  y = fe_reduce_weak<FrP>(fe_add<FrP>(carry, fe_mul<FrP>(C, mult)));
  if (TW) {
#pragma unroll
    for (int k = 0; k < SC_K; ++k) {
      if (lo + k < m) st_tw(out, lo + k, y);
      y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), d[k]));
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < SC_K; ++k) {
    if (lo + k < m) {
      y = fe_reduce_weak<FrP>(fe_add<FrP>(fe_mul<FrP>(y, z), d[k]));
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
// Montgomery's trick per thread over `L` elements taken with stride T (coalesced), one inversion per thread.  Forms: u = a * 2^5 (ABI -> device form, a shift and a weak reduction, no
// product); prefix products in device form; the inverse of the thread's product is moved to ABI form
// once, after which inverse(ABI) x prefix(device) and inverse(ABI) x u(device) both stay in ABI form:
// 3 products per element (r01: 6) + one binary-GCD inversion per thread (r03; r02: x^(r-2), ~325 dependent products,
// a 0.15 ms latency floor however the elements were distributed).
__global__ void __launch_bounds__(256) batch_inverse_kernel(u32x4* v, size_t n, u32 L, u32x4* scratch) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Fr one = fe_one<FrP>();
  Fr acc = one;
  for (u32 j = 0; j < L; ++j) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    const Fr raw = ld_canon(v, i);
    u32 nz = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) nz |= raw.l[q];
    st_tw(scratch, i, acc);                    // product of the earlier non-zero elements
    if (nz) acc = fe_mul<FrP>(acc, abi_to_dev(raw));
  }
  // 1 / acc by the binary GCD of field_inv.hip.h (~16 k instructions; r01 / r02: acc^(r-2), ~76 k and 57 % of a thread's
  // work).  acc = x R' (device form): the integer inverse is x^-1 / R'; times R R'^2 (and the product's 1 / R') gives
  // x^-1 R, the ABI form the back-substitution wants
  Fr inv = fe_mul<FrP>(fe_inv_int<FrP>(fe_canon_limbs<FrP>(acc)), fe_pow2<FrP, 256 + 2 * 261>());
  for (u32 j = L; j-- > 0;) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) continue;
    const Fr raw = ld_canon(v, i);
    u32 nz = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) nz |= raw.l[q];
    if (!nz) continue;                         // zero stays zero
    st_canon(v, i, fe_mul<FrP>(inv, ld_tw(scratch, i)));   // a_i^-1, ABI form
    inv = fe_mul<FrP>(inv, abi_to_dev(raw));
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
static size_t eval_slot_bytes(uint32_t k, size_t n) {
  const u32 L = (u32)std::max<size_t>(1, std::min<size_t>(64, n / ((size_t)256 * 1024)));
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
  // one pass (see pm_fr_prefix_product_dev) while the tiles are few: this recurrence costs ~4.5 products per element in one
  // pass (the powers of z every thread and every look-back lane needs) against 2.2 in three stages -- 2^16 69.7 -> 44.5 us,
  // 2^18 87.2 -> 48.6, 2^19 67 -> 51, but 2^20 79 (the r05 hybrid below) against 94, 2^21 109 against 191: up to 1.5 tiles per CU
  // (profiles/r03_poly_rows.txt, profiles/r05_poly_rows.txt)
  const size_t lookback_max = ctx->opt_poly_lookback == 2 ? ((size_t)1 << 31) : (size_t)SC_TILE * 3 * (size_t)ctx->num_cus / 2;
  if (ctx->opt_poly_lookback && m > (size_t)SC_TILE && m <= lookback_max) {
    const u32 tiles = (u32)((m + SC_TILE - 1) / SC_TILE);
    const size_t head = 64 + (size_t)tiles * 48;
    OrderScope order_scope(ctx, ctx->ord_poly, st);
    int rc = order_scope.rc;
    if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, head);
    if (rc) return rc;
    RufLook lk2;
    const HFr zt = hfr_pow_u64(zz, SC_TILE);
    to_limbs29(lk2.z, zz);
    to_limbs29(lk2.zt, zt);
    to_limbs29(lk2.zt64, hfr_pow_u64(zt, 64));
    const size_t lds = (size_t)SC_LDS_SLOTS * 16;
    const void* fn = (const void*)ruf_lookback_kernel<false>;
    if (int lrc = raise_lds_limit(ctx, fn, lds)) return lrc;
    ProfScope prof(ctx, st, "fr_poly_ruffini");
    PM_HIP(ctx, hipMemsetAsync(ctx->poly_ws.ptr, 0, head, st));
    hipLaunchKernelGGL(ruf_lookback_kernel<false>, dim3(tiles), dim3(256), lds, st, (const u32x4*)d_coeffs, n, m, (u32x4*)d_out,
                       (u32*)ctx->poly_ws.ptr, tiles, lk2);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
  }
  // level sizes: m, ceil(m / K), ... until one workgroup holds the level
  // r05 (as for the prefix product): the descent stops at the first inner level whose tiles are all resident at once (this kernel:
  // one workgroup per CU) and that level is scanned by ONE look-back launch (ruf_lookback_kernel<true>)
  const size_t tw_resident = (size_t)SC_TILE * (size_t)ctx->num_cus;
  std::vector<size_t> sz;
  sz.push_back(m);
  bool top_lookback = false;
  while (sz.back() > (size_t)SC_BASE) {
    sz.push_back((sz.back() + SC_K - 1) / SC_K);
    if (ctx->opt_poly_lookback && sz.back() > (size_t)SC_BASE && sz.back() <= tw_resident) {
      top_lookback = true;
      break;
    }
  }
  size_t tot_entries = 0;
  for (size_t i = 1; i < sz.size(); ++i) tot_entries += sz[i];
  if (sz.size() == 1) tot_entries = (m + SC_K - 1) / SC_K;       // a small input still goes totals -> base -> replay
  const u32 top_tiles = top_lookback ? (u32)((sz.back() + SC_TILE - 1) / SC_TILE) : 0u;
  const size_t lvl_bytes = (tot_entries * 48 + 64 + 255) / 256 * 256, ctl_bytes = top_lookback ? 64 + (size_t)top_tiles * 48 : 0;
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, lvl_bytes + ctl_bytes);
  if (rc) return rc;
  if (sz.size() == 1) sz.push_back((m + SC_K - 1) / SC_K);
  std::vector<u32x4*> lvl(sz.size());
  {
    u32x4* p = (u32x4*)ctx->poly_ws.ptr;
    for (size_t i = 1; i < sz.size(); ++i) {
      lvl[i] = p;
      p += 3 * sz[i];
    }
  }
  std::vector<RufLevel> zl(sz.size());
  {
    HFr zz_l = zz;
    for (size_t i = 0; i < sz.size(); ++i) {
      to_limbs29(zl[i].z, zz_l);
      zz_l = hfr_pow_u64(zz_l, SC_K);
    }
  }
  const size_t sc_lds_bytes = (size_t)SC_LDS_SLOTS * 16;
  for (const void* fn : {(const void*)ruf_totals0_kernel, (const void*)ruf_replay0_kernel})
    if (int lrc = raise_lds_limit(ctx, fn, sc_lds_bytes)) return lrc;
  ProfScope prof(ctx, st, "fr_poly_ruffini");
  const size_t last = sz.size() - 1;
  for (size_t i = 0; i < last; ++i) {   // totals of level i -> level i + 1
    const unsigned blocks = (unsigned)((sz[i + 1] + 255) / 256);
    if (i == 0)
      hipLaunchKernelGGL(ruf_totals0_kernel, dim3((unsigned)((m + SC_TILE - 1) / SC_TILE)), dim3(256), sc_lds_bytes, st,
                         (const u32x4*)d_coeffs, n, m, lvl[1], zl[0]);
    else
      hipLaunchKernelGGL(ruf_totals_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4*)lvl[i], sz[i], lvl[i + 1],
                         sz[i + 1], zl[i]);
  }
  if (top_lookback) {
    u32* ctl = (u32*)((char*)ctx->poly_ws.ptr + lvl_bytes);
    HFr zlast = zz;
    for (size_t i = 0; i < last; ++i) zlast = hfr_pow_u64(zlast, SC_K);        // this level's multiplier z^(SC_K^last)
    RufLook lk2;
    const HFr zt = hfr_pow_u64(zlast, SC_TILE);
    to_limbs29(lk2.z, zlast);
    to_limbs29(lk2.zt, zt);
    to_limbs29(lk2.zt64, hfr_pow_u64(zt, 64));
    PM_HIP(ctx, hipMemsetAsync(ctl, 0, ctl_bytes, st));
    hipLaunchKernelGGL(ruf_lookback_kernel<true>, dim3(top_tiles), dim3(256), 0, st, (const u32x4*)lvl[last], sz[last], sz[last],
                       lvl[last], ctl, top_tiles, lk2);
  } else {
    hipLaunchKernelGGL(ruf_base_kernel, dim3(1), dim3(256), 0, st, lvl[last], (u32)sz[last], zl[last]);
  }
  for (size_t i = last; i-- > 0;) {     // carries of level i + 1 -> outputs of level i
    const unsigned blocks = (unsigned)((sz[i + 1] + 255) / 256);
    if (i == 0)
      hipLaunchKernelGGL(ruf_replay0_kernel, dim3((unsigned)((m + SC_TILE - 1) / SC_TILE)), dim3(256), sc_lds_bytes, st,
                         (const u32x4*)d_coeffs, n, m, (const u32x4*)lvl[1], (u32x4*)d_out, zl[0]);
    else
      hipLaunchKernelGGL(ruf_replay_kernel, dim3(blocks), dim3(256), 0, st, lvl[i], sz[i], (const u32x4*)lvl[i + 1],
                         sz[i + 1], zl[i]);
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
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  if (!d_inout) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  // enough threads to fill the chip, at most 64 elements per thread
  // one Fermat inversion (~380 products) per thread: 64 elements per thread amortise it to ~6
  // products per element; below 2^22 elements keep at least one wave per SIMD instead
  const size_t per = 32;   // elements per thread and inversion (2^22: 271 us with 32, 287 with 64, 318 with 16)
  const size_t want_threads = std::max<size_t>((n + per - 1) / per, std::min<size_t>(n, (size_t)ctx->num_cus * 256));
  const unsigned blocks = (unsigned)((want_threads + 255) / 256);
  const size_t T = (size_t)blocks * 256;
  const u32 L = (u32)((n + T - 1) / T);
  OrderScope order_scope(ctx, ctx->ord_poly, st);
  int rc = order_scope.rc;
  if (!rc) rc = ensure_buffer(ctx, ctx->poly_ws, n * 48);
  if (rc) return rc;
  ProfScope prof(ctx, st, "fr_batch_inverse");
  hipLaunchKernelGGL(batch_inverse_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)d_inout, n, L,
                     (u32x4*)ctx->poly_ws.ptr);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}
