// Bucket fill of the G1 MSM: scalars -> signed window digits -> (bucket key, point) pairs grouped by bucket.
//
// This is the per-window bucket fill of dusk_bls12_381::multiscalar_mul::msm_variable_base (dusk-bls12_381 0.8,
// ref:Cargo.toml:20): "for each scalar, take its c-bit digit and add the point to buckets[digit - 1]".
// On the GPU the additions of one bucket must sit next to each other for the segmented accumulate of msm.hip, so the
// fill is a sort of m = n x windows pairs by bucket.  Three hand-written kernels, most-significant bits first, no
// intermediate (key, value) arrays between digit extraction and the first partition pass (r01/r02 wrote the pairs
// with one kernel and sorted them with rocPRIM's onesweep: write 8 m, then two or three passes of read + write 8 m):
//
//   msm_digits_hist_kernel     reads the scalars, extracts the digits in registers and counts the pairs per
//                              PARTITION (bucket set x top P bits of the bucket index) in an LDS histogram, per tile
//                              of 1024 scalars; leaves the integer form of the scalars (32 B each: cheaper than
//                              parking 8 B x windows pairs in HBM) and the tiles' count rows for the next kernel; the
//                              workgroup that finishes last turns the totals into partition offsets (exclusive scan)
//   msm_digits_scatter_kernel  extracts the digits again from the integer scalars, orders one tile's pairs by
//                              partition in LDS, reserves a range per (tile, partition) with ONE returning atomic
//                              on the partition's cursor and writes whole runs
//   msm_sort_local_kernel      one workgroup per partition: counting sort on the remaining R bits in LDS
//                              (ranks from LDS atomics, staged output, coalesced stores); partitions that do not
//                              fit one tile of 16 K pairs (2^24-point MSMs, skewed scalars) are streamed tile by
//                              tile against running per-bucket cursors kept in LDS
//
// The order inside a bucket is arbitrary (atomics); the bucket sum is the same group element in any order and the
// MSM result leaves the library in affine-normalised form, so results stay bit-exact.  Zero digits are dropped here
// (r02 sorted them to the end under a trash key): the number of pairs that exist, and the chunk length the
// accumulate kernel derives from it, live in a small control block in device memory.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/plonk_mi355x.h"
#include "fields.hip.h"

namespace pm {

static constexpr u32 KEY_INVALID = 0xffffffffu;

static constexpr u32 SORT_THREADS = 1024;        // local sort: 16 waves
#ifndef SORT_T1
#define SORT_T1 1024
#endif
static constexpr u32 SORT_THREADS1 = SORT_T1;    // scatter kernel: threads = scalars per tile at most
#ifndef SORT_T0
#define SORT_T0 256
#endif
static constexpr u32 SORT_THREADS0 = SORT_T0;    // histogram kernel (a tile is walked in steps of this many scalars)
static constexpr u32 SORT_TILE1_PAIRS = 14 * SORT_THREADS1;   // scatter kernel: pairs staged per tile (8 B each in LDS)
static constexpr u32 SORT_IPT = 16;              // local sort: pairs per thread and tile
static constexpr u32 SORT_TILE2_PAIRS = SORT_THREADS * SORT_IPT;
static constexpr u32 SORT_MAX_BINS = 4096;       // partitions one MSM of a batch may have (LDS histogram of a tile)
static constexpr u32 SORT_MAX_RBITS = 12;        // local bins: 2^R counters + 2^R cursors in LDS

struct MsmGeom {
  u32 c;         // window bits
  u32 nwin;      // digit windows
  u32 nsets;     // bucket sets per MSM: nwin, or 1 when the bases carry a table of 2^(c w) P
  u32 batch;     // MSMs sharing the bases in this launch sequence (their sets are laid side by side)
  u32 bbits;     // c - 1: bits of the bucket field
  u32 nbuckets;  // 1 << bbits per set
  u32 trash;     // first key that is not a bucket (nsets * batch << bbits)
  u32 row_stride;  // table mode: points per table row (value = window * row_stride + index)
  // the sort plan
  u32 pbits;     // P: bits of the bucket index that select the partition
  u32 rbits;     // R = bbits - P: bits left for the local sort
  u32 sa;        // table mode: the buckets below a_size (the short TOP window's digits land there, on top of every
  u32 a_size;    //   other window's share) are cut into partitions 2^sa times narrower; 0 = one width everywhere
  u32 na;        // partitions below a_size
  u32 na_off;    // na - (a_size >> R): what the narrow region adds to the partition index of the rest
  u32 pps;       // partitions per bucket set: (1 << P) + na_off
  u32 bins;      // partitions per MSM of the batch: nsets * pps
  u32 np;        // partitions in all: batch * bins
  u32 ts;        // scalars per tile of the histogram / scatter kernels
  u32 tiles;     // tiles per MSM of the batch
  u32 ctl_cap;   // partitions the control block is laid out for (>= np; fixed per allocation, set by the caller)
};

// control block (u32 words) in device memory, zero when idle: the histogram kernel leaves it zeroed again
static constexpr u32 CTL_TICKET = 0;     // workgroups of the histogram kernel that have flushed
static constexpr u32 CTL_M_EFF = 1;      // pairs that exist (non-zero digits)
static constexpr u32 CTL_CHUNK = 2;      // entries per thread for the level-1 accumulate
static constexpr u32 CTL_HEADER = 8;
// then: count[cap], start[cap + 1], cursor[cap] for a capacity of `cap` partitions that is fixed when the block is
// allocated -- NOT the np of the current MSM: only count[] returns to zero, and with offsets that moved with np the
// counters of one MSM would land on the offsets the previous one left behind
static inline size_t sort_ctl_words(u32 cap) { return CTL_HEADER + 3 * (size_t)cap + 1; }

static inline MsmGeom make_geom(size_t n, long opt_c, u32 table_c, size_t table_stride, u32 batch) {
  MsmGeom g;
  u32 lg = 0;
  while (((size_t)1 << (lg + 1)) <= (n > 1 ? n : 1)) ++lg;
  long c = opt_c ? opt_c : (lg < 9 ? 5 : (lg > 20 ? 16 : (long)lg - 4));
  if (table_c) c = table_c;  // fixed when the table was built
  g.c = (u32)c;
  g.nwin = (256 + g.c - 1) / g.c;
  g.nsets = table_c ? 1u : g.nwin;
  g.batch = batch;
  g.bbits = g.c - 1;
  g.nbuckets = 1u << g.bbits;
  g.trash = (g.nsets * batch) << g.bbits;
  g.row_stride = (u32)table_stride;
  // partitions of ~13 K pairs (one tile of the local sort with room for the spread of uniform digits), as long as a
  // tile of the scatter kernel still writes runs of several pairs per partition
  const size_t per_set = table_c ? n * g.nwin : n;
  u32 p = 0;
  while (p < g.bbits && (per_set >> p) > 13500) ++p;
  const u32 p_min = g.bbits > SORT_MAX_RBITS ? g.bbits - SORT_MAX_RBITS : 0u;
  u32 p_max = 10;
  while (p_max > p_min && (g.nsets << p_max) > SORT_MAX_BINS) --p_max;
  if (p > p_max) p = p_max;
  if (p < p_min) p = p_min;
  g.pbits = p;
  g.rbits = g.bbits - p;
  // With the window table every window feeds ONE bucket set, and the top window is short (c = 20: 16 bits, digits
  // below 2^15 of 2^19 buckets): the low buckets carry (windows - 1) / 2^bbits + 1 / 2^tb of the pairs each instead
  // of (windows - 1) / 2^bbits -- 2.3 times the others for c = 20, 24 times for c = 22.  Partitions of equal width
  // would make 64 (c = 22: 8) workgroups of the local sort run two (ten) tiles while the rest run one; the low
  // region is cut 2^sa times finer instead, so that every partition holds about the same number of pairs.
  g.sa = g.a_size = g.na = g.na_off = 0;
  if (table_c && g.nwin > 1 && p >= 1) {
    const u32 tbits = 256 - g.c * (g.nwin - 1);   // bits of the top window; the scalar is below 2^255
    if (tbits < g.c && tbits >= 2 && tbits - 1 >= g.rbits) {
      const u32 tb = tbits - 1;
      const double ratio = 1.0 + (double)(1u << (g.bbits - tb)) / (double)(g.nwin - 1);
      u32 sa = 0;
      while (ratio / (double)(1u << sa) > 1.18 && sa < g.rbits) ++sa;
      if (sa) {
        g.sa = sa;
        g.a_size = 1u << tb;
        g.na = g.a_size >> (g.rbits - sa);
        g.na_off = g.na - (g.a_size >> g.rbits);
      }
    }
  }
  g.pps = (1u << p) + g.na_off;
  g.bins = g.nsets * g.pps;
  g.np = batch * g.bins;
  g.ts = SORT_TILE1_PAIRS / g.nwin < SORT_THREADS1 ? SORT_TILE1_PAIRS / g.nwin : SORT_THREADS1;
  g.tiles = (u32)((n + g.ts - 1) / g.ts);
  g.ctl_cap = g.np;
  return g;
}

#ifdef SORT_TIMING   // tools/sort_bench.hip: phase timestamps (100 MHz wall clock) of one workgroup per kernel
__device__ unsigned long long sort_dbg[3][16];
__device__ unsigned long long sort_span[3][4096][2];   // [kernel][workgroup]: first and last stamp
#define SORT_T(kern, i)                                                                                    \
  do {                                                                                                     \
    if (threadIdx.x == 0) {                                                                                \
      const unsigned long long now_ = wall_clock64();                                                      \
      if (blockIdx.x == gridDim.x / 2) sort_dbg[kern][i] = now_;                                           \
      if (blockIdx.x < 4096 && (i) == 0) sort_span[kern][blockIdx.x][0] = now_;                            \
      if (blockIdx.x < 4096 && (i) == 5) sort_span[kern][blockIdx.x][1] = now_;                            \
    }                                                                                                      \
  } while (0)
#else
#define SORT_T(kern, i)
#endif

// partition of a bucket inside its set, and back: first bucket and bucket-index bits of a partition
__host__ __device__ inline u32 part_of(const MsmGeom& g, u32 bucket) {
  return bucket < g.a_size ? bucket >> (g.rbits - g.sa) : g.na_off + (bucket >> g.rbits);
}
__host__ __device__ inline void part_range(const MsmGeom& g, u32 pl, u32& first_bucket, u32& bits) {
  if (pl < g.na) {
    bits = g.rbits - g.sa;
    first_bucket = pl << bits;
  } else {
    bits = g.rbits;
    first_bucket = (pl - g.na_off) << bits;
  }
}

// ------------------------------------------------------------------ shared pieces
// A wave-uniform word that an earlier kernel wrote (partition offsets, the pair count), read with an agent-scope
// load: a vector load past the scalar and vector L1 caches instead of a scalar load through the constant cache.
PM_DEV u32 ld_uniform(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive scan, in place, of a[0, nb) (LDS, nb <= 8 * NT) by the whole workgroup of NT threads; returns the total.
// The caller has synchronised the writes to a[]; wsum is 16 words of LDS.
template <u32 NT = SORT_THREADS>
PM_DEV u32 block_exclusive_scan(u32* a, u32 nb, u32* wsum) {
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 per = (nb + NT - 1) / NT;
  const u32 b0 = tid * per;
  u32 v[8], s = 0;
#pragma unroll
  for (u32 i = 0; i < 8; ++i) {
    v[i] = s;
    if (i < per && b0 + i < nb) s += a[b0 + i];
  }
  u32 inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(inc, d);
    if ((int)lane >= d) inc += t;
  }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  u32 wp = 0, total = 0;
#pragma unroll
  for (u32 w = 0; w < NT / 64; ++w) {
    const u32 t = wsum[w];
    if (w < wave) wp += t;
    total += t;
  }
  const u32 base = wp + inc - s;
#pragma unroll
  for (u32 i = 0; i < 8; ++i)
    if (i < per && b0 + i < nb) a[b0 + i] = base + v[i];
  __syncthreads();
  return total;
}

// signed c-bit digits of a 256-bit integer, least significant window first: emit(window, |d|, negative).
// The words are consumed through a 64-bit shift register so that w[] is only ever indexed statically.
template <class F>
PM_DEV void for_each_digit(const u32 (&w)[8], u32 c, u32 nwin, F&& emit) {
  const u32 mask = (1u << c) - 1u, half = 1u << (c - 1);
  u64 buf = 0;
  u32 have = 0, k = 0, carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    buf |= (u64)w[i] << have;
    have += 32;
    while (have >= c && k < nwin) {
      u32 d = ((u32)buf & mask) + carry;
      buf >>= c;
      have -= c;
      u32 neg = 0;
      if (d > half) {
        d = (1u << c) - d;
        neg = 1;
        carry = 1;
      } else {
        carry = 0;
      }
      emit(k, d, neg);
      ++k;
    }
  }
  if (k < nwin) {   // the top window holds fewer than c bits (the scalar is below 2^255: no carry leaves it)
    u32 d = ((u32)buf & mask) + carry;
    u32 neg = 0;
    if (d > half) {
      d = (1u << c) - d;
      neg = 1;
    }
    emit(k, d, neg);
  }
}

// The same for a window width known at compile time: every shift and word index is a constant, no loop control (the
// generic walk above costs ~40 instruction slots per window, three times this one; the digit walk is the largest
// part of both digit kernels).  The widths the library picks by itself (13, 16, 20 with a table; 22 as an option) get
// this form, anything else the generic one.
template <u32 C, class F>
PM_DEV void for_each_digit_c(const u32 (&w)[8], F&& emit) {
  constexpr u32 NW = (256 + C - 1) / C, MASK = (1u << C) - 1u, HALF = 1u << (C - 1);
  u32 carry = 0;
#pragma unroll
  for (u32 k = 0; k < NW; ++k) {
    const u32 lo = k * C, j = lo >> 5, sh = lo & 31u;
    u32 d = w[j] >> sh;
    if (sh + C > 32 && j + 1 < 8) d |= w[j + 1] << (32 - sh);
    d = (d & MASK) + carry;
    u32 neg = 0;
    if (d > HALF) {
      d = (1u << C) - d;
      neg = 1;
    }
    carry = neg;
    emit(k, d, neg);
  }
}
template <class F>
PM_DEV void digits(const u32 (&w)[8], u32 c, u32 nwin, F&& emit) {
  switch (c) {
    case 13: for_each_digit_c<13>(w, emit); break;
    case 16: for_each_digit_c<16>(w, emit); break;
    case 20: for_each_digit_c<20>(w, emit); break;
    case 22: for_each_digit_c<22>(w, emit); break;
    default: for_each_digit(w, c, nwin, emit);
  }
}

// ------------------------------------------------------------------ 1: histogram + partition offsets
// Per tile of g.ts scalars: the integer form of every scalar (kept for the scatter kernel: the Montgomery -> integer
// product and the canonical packing are ~350 of the ~600 instructions a scalar costs here, and both kernels are
// instruction-bound), the tile's pair count per partition as a row of u16 (the scatter kernel starts from it instead
// of counting again), and the partition totals by atomics.  The workgroup that takes the last ticket scans the totals.
template <u32 NT>
__global__ void __launch_bounds__(NT) msm_digits_hist_kernel(const u32x4* scalars, size_t n, size_t sc_stride,
                                                                       u32 scalar_form, const MsmGeom g, u32 tiles_per_wg,
                                                                       u32 wgs_per_msm, u32 l1_threads, u32 min_chunk,
                                                                       u32* ctl, u32x4* canon, unsigned short* rows) {
  extern __shared__ u32 lds_hist[];   // g.bins counters
  __shared__ u32 wsum[NT / 64];
  __shared__ u32 s_last;
  const u32 tid = threadIdx.x;
  const u32 j = blockIdx.x / wgs_per_msm, wg = blockIdx.x % wgs_per_msm;
  u32* count = ctl + CTL_HEADER;
  const u32 t0 = wg * tiles_per_wg, t1 = t0 + tiles_per_wg < g.tiles ? t0 + tiles_per_wg : g.tiles;
  for (u32 t = t0; t < t1; ++t) {
    SORT_T(0, 0);
    for (u32 b = tid; b < g.bins; b += NT) lds_hist[b] = 0;
    __syncthreads();
    // NT threads walk the tile of g.ts scalars in steps of NT; the next step's scalar is requested before the current
    // one is processed (and several workgroups share a CU: the loads of one hide behind the arithmetic of the others)
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
    u32x4 ra = zero4, rb = zero4;
    {
      const size_t i = (size_t)t * g.ts + tid;
      if (tid < g.ts && i < n) {
        const u32x4* sp = scalars + 2 * ((size_t)j * sc_stride + i);
        ra = sp[0];
        rb = sp[1];
      }
    }
    for (u32 q0 = 0; q0 < g.ts; q0 += NT) {
      const u32 si = q0 + tid;
      const size_t i = (size_t)t * g.ts + si;
      const u32 sw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
      {
        const u32 sn = si + NT;
        const size_t in = i + NT;
        if (sn < g.ts && in < n) {
          const u32x4* sp = scalars + 2 * ((size_t)j * sc_stride + in);
          ra = sp[0];
          rb = sp[1];
        }
      }
      if (si < g.ts && i < n) {
        Fr s = fe_unpack<FrP>(sw);
        // Montgomery (x * 2^256): multiply by 2^5 / 2^261 -> x.   Canonical: multiply by one -> x mod r.
        if (scalar_form == PM_SCALAR_MONTGOMERY)
          s = fe_mul_limb<FrP>(s, 32u);
        else
          s = fe_mul<FrP>(s, fe_one<FrP>());
        u32 w[8];
        fe_canon_pack<FrP>(w, s);
        u32x4* co = canon + 2 * ((size_t)j * n + i);
        co[0] = u32x4{w[0], w[1], w[2], w[3]};
        co[1] = u32x4{w[4], w[5], w[6], w[7]};
        digits(w, g.c, g.nwin, [&](u32 k, u32 d, u32) {
          if (d) atomicAdd(&lds_hist[(g.nsets == 1 ? 0u : k) * g.pps + part_of(g, d - 1)], 1u);
        });
      }
    }
    __syncthreads();
    SORT_T(0, 1);
    unsigned short* row = rows + ((size_t)j * g.tiles + t) * g.bins;
    for (u32 b = tid; b < g.bins; b += NT) {
      const u32 v = lds_hist[b];
      row[b] = (unsigned short)v;
      if (v) atomicAdd(&count[(size_t)j * g.bins + b], v);
    }
    __syncthreads();
  }
  // every wave waits for its own adds to be acknowledged, then one lane takes the ticket (a __threadfence() per
  // thread here writes back the XCD's L2 16 K times per launch: 290 us instead of 30 at 2^20)
  SORT_T(0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SORT_T(0, 5);
  if (tid == 0) s_last = (atomicAdd(&ctl[CTL_TICKET], 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: counts -> exclusive offsets; the counts and the ticket go back to zero for the next MSM
  __threadfence();
  u32* start = count + g.ctl_cap;
  u32* cursor = start + g.ctl_cap + 1;
  const u32 per = (g.np + NT - 1) / NT;
  const u32 b0 = tid * per, b1 = b0 + per < g.np ? b0 + per : g.np;
  u32 s = 0;
  for (u32 b = b0; b < b1; ++b) {   // read-and-reset at the place the adds were performed; parked in start[] for pass two
    const u32 v = atomicExch(&count[b], 0u);
    start[b] = v;
    s += v;
  }
  const u32 lane = tid & 63u, wave = tid >> 6;
  u32 inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(inc, d);
    if ((int)lane >= d) inc += t;
  }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  u32 wp = 0, total = 0;
#pragma unroll
  for (u32 w = 0; w < NT / 64; ++w) {
    const u32 t = wsum[w];
    if (w < wave) wp += t;
    total += t;
  }
  u32 run = wp + inc - s;
  for (u32 b = b0; b < b1; ++b) {
    const u32 v = start[b];
    start[b] = run;
    cursor[b] = run;
    run += v;
  }
  if (tid == 0) {
    start[g.np] = total;
    ctl[CTL_M_EFF] = total;
    // entries per thread of the level-1 accumulate: the grid was sized for every digit being non-zero
    u32 chunk = l1_threads ? (total + l1_threads - 1) / l1_threads : total;
    if (chunk < min_chunk) chunk = min_chunk;
    ctl[CTL_CHUNK] = chunk;
    ctl[CTL_TICKET] = 0;
  }
}

// ------------------------------------------------------------------ 2: scatter into the partitions
// One workgroup per CU (the staged tile fills most of the LDS) walks tiles blockIdx.x, + gridDim.x, ..; the integer
// scalars and the count row of the NEXT tile are requested before the current one is processed.
template <u32 NT>
__global__ void __launch_bounds__(NT) msm_digits_scatter_kernel(const u32x4* canon, const unsigned short* rows, size_t n,
                                                                const MsmGeom g, u32 offset, u32* ctl, u64* pairs,
                                                                u32 tiles_total) {
  extern __shared__ u32 lds_sc[];   // [bins, rounded up to even] counts -> cursors -> deltas, then the staged pairs
  __shared__ u32 wsum[NT / 64];
  constexpr u32 NQ = SORT_MAX_BINS / NT;
  const u32 tid = threadIdx.x;
  u32* hist = lds_sc;
  u64* stage = reinterpret_cast<u64*>(lds_sc + ((g.bins + 1) & ~1u));
  u32* cursor = ctl + CTL_HEADER + 2 * (size_t)g.ctl_cap + 1;
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
  u32x4 wa = zero4, wb = zero4;
  u32 rv[NQ];
  auto fetch = [&](u32 tile, u32x4& a_, u32x4& b_, u32 (&r_)[NQ]) {
    const u32 j_ = tile / g.tiles, t_ = tile % g.tiles;
    const size_t i_ = (size_t)t_ * g.ts + tid;
    a_ = zero4;
    b_ = zero4;
    if (tid < g.ts && i_ < n) {
      const u32x4* ci = canon + 2 * ((size_t)j_ * n + i_);
      a_ = ci[0];
      b_ = ci[1];
    }
    const unsigned short* row = rows + (size_t)tile * g.bins;
#pragma unroll
    for (u32 q = 0; q < NQ; ++q) {
      const u32 b = tid + q * NT;
      r_[q] = b < g.bins ? row[b] : 0u;
    }
  };
  u32 tile = blockIdx.x;
  if (tile < tiles_total) fetch(tile, wa, wb, rv);
  for (; tile < tiles_total; tile += gridDim.x) {
    const u32 j = tile / g.tiles, t = tile % g.tiles;
    const size_t i = (size_t)t * g.ts + tid;
    const bool live = tid < g.ts && i < n;
    SORT_T(1, 0);
    // one returning atomic per non-empty (tile, partition): the range this tile's run goes to (its latency hides
    // behind the scan and the LDS pass below: the result is first needed for the deltas)
    u32 gb[NQ];
#pragma unroll
    for (u32 q = 0; q < NQ; ++q) {
      const u32 b = tid + q * NT;
      gb[q] = 0;
      if (b < g.bins) {
        hist[b] = rv[q];
        if (rv[q]) gb[q] = atomicAdd(&cursor[(size_t)j * g.bins + b], rv[q]);
      }
    }
    const u32 w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    if (tile + gridDim.x < tiles_total) fetch(tile + gridDim.x, wa, wb, rv);   // the next tile: in flight from here on
    __syncthreads();
    SORT_T(1, 1);
    const u32 total = block_exclusive_scan<NT>(hist, g.bins, wsum);
    SORT_T(1, 2);
    if (live) {
      digits(w, g.c, g.nwin, [&](u32 k, u32 d, u32 neg) {
        if (d) {
          const u32 set_local = g.nsets == 1 ? 0u : k;
          const u32 pos = atomicAdd(&hist[set_local * g.pps + part_of(g, d - 1)], 1u);
          const u32 key = ((j * g.nsets + set_local) << g.bbits) | (d - 1);
          // table of 2^(c k) P (nsets == 1): every window feeds the one bucket set of its MSM
          const u32 val = ((g.nsets == 1 ? k * g.row_stride : 0u) + offset + (u32)i) | (neg << 31);
          stage[pos] = ((u64)key << 32) | val;
        }
      });
    }
    __syncthreads();
    SORT_T(1, 3);
    // hist[b] is now the END of bin b in the staged tile; delta[b] = reserved range - start of the bin
    u32 st[NQ];
#pragma unroll
    for (u32 q = 0; q < NQ; ++q) {
      const u32 b = tid + q * NT;
      st[q] = (b < g.bins && b > 0) ? hist[b - 1] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (u32 q = 0; q < NQ; ++q) {
      const u32 b = tid + q * NT;
      if (b < g.bins) hist[b] = gb[q] - st[q];
    }
    __syncthreads();
    SORT_T(1, 4);
    for (u32 e = tid; e < total; e += NT) {
      const u64 pr = stage[e];
      const u32 key = (u32)(pr >> 32);
      const u32 set_local = (key >> g.bbits) - j * g.nsets;
      const u32 bin = set_local * g.pps + part_of(g, key & (g.nbuckets - 1u));
      pairs[hist[bin] + e] = pr;
    }
    __syncthreads();   // the next tile overwrites hist[] and the stage
    SORT_T(1, 5);
  }
}

// ------------------------------------------------------------------ 3: counting sort inside a partition
__global__ void __launch_bounds__(SORT_THREADS) msm_sort_local_kernel(const MsmGeom g, const u32* ctl, const u64* pairs,
                                                                      u32* keys_out, u32* vals_out) {
  extern __shared__ u32 lds_ls[];   // cur[2^R] | thist[2^R] | stage_v[TILE2] | stage_k[TILE2] (u16)
  __shared__ u32 wsum[SORT_THREADS / 64];
  const u32 tid = threadIdx.x;
  const u32 p = blockIdx.x;
  const u32* startp = ctl + CTL_HEADER + g.ctl_cap;
  const u32 lo = ld_uniform(startp + p), size = ld_uniform(startp + p + 1) - lo;
  if (size == 0) return;
  u32 first_bucket, pr_bits;
  part_range(g, p % g.pps, first_bucket, pr_bits);
  const u32 nb = 1u << pr_bits, lmask = nb - 1u;
  u32* cur = lds_ls;
  u32* thist = cur + nb;
  u32* stage_v = thist + nb;
  unsigned short* stage_k = reinterpret_cast<unsigned short*>(stage_v + SORT_TILE2_PAIRS);
  const u32 key_hi = ((p / g.pps) << g.bbits) | first_bucket;
  const u32 ntile = (size + SORT_TILE2_PAIRS - 1) / SORT_TILE2_PAIRS;
  const u64* src = pairs + lo;
  if (ntile > 1) {   // streamed partition: bucket counts over the whole partition first
    SORT_T(2, 6);
    for (u32 b = tid; b < nb; b += SORT_THREADS) cur[b] = 0;
    __syncthreads();
    for (u32 e0 = 0; e0 < size; e0 += SORT_THREADS) {
      const u32 e = e0 + tid;
      const bool on = e < size;
      const u32 b = on ? (u32)(src[e] >> 32) & lmask : 0u;
      // a wave whose lanes all hit ONE bucket (all-equal scalars, a dominant value) adds once: 64 atomics on one
      // LDS word are served one after the other
      const u64 act = __ballot(on);
      const u32 b0 = __shfl(b, act ? __ffsll((long long)act) - 1 : 0);
      if (__ballot(on && b == b0) == act) {
        if (on && (tid & 63u) == (u32)(__ffsll((long long)act) - 1)) atomicAdd(&cur[b0], (u32)__popcll(act));
      } else if (on) {
        atomicAdd(&cur[b], 1u);
      }
    }
    __syncthreads();
    (void)block_exclusive_scan(cur, nb, wsum);
    SORT_T(2, 7);
  }
  for (u32 t = 0; t < ntile; ++t) {
    const u32 base = t * SORT_TILE2_PAIRS;
    const u32 cnt = size - base < SORT_TILE2_PAIRS ? size - base : SORT_TILE2_PAIRS;
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 0);   // tools/sort_bench: a full tile
    for (u32 b = tid; b < nb; b += SORT_THREADS) thist[b] = 0;
    u32 lk[SORT_IPT], lv[SORT_IPT];
#pragma unroll
    for (u32 q = 0; q < SORT_IPT; ++q) {
      const u32 e = tid + q * SORT_THREADS;
      const u64 pr = e < cnt ? src[base + e] : 0ull;
      lk[q] = (u32)(pr >> 32) & lmask;
      lv[q] = (u32)pr;
    }
    __syncthreads();
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 1);   // tools/sort_bench: a full tile
    u32 rk[SORT_IPT];
    {
      // ranks inside the tile from LDS atomics.  A wave whose pairs ALL carry one bucket (all-equal scalars, a
      // dominant value) takes one atomic for the lot -- 64 adds to one LDS word are served one after the other --
      // tested once per tile, not per pair
      u32 mine = 0, x = lk[0];
      bool same = true;
#pragma unroll
      for (u32 q = 0; q < SORT_IPT; ++q)
        if (tid + q * SORT_THREADS < cnt) {
          same = same && lk[q] == x;
          ++mine;
        }
      const u32 lane = tid & 63u;
      const u64 act = __ballot(mine != 0);
      const u32 first = act ? (u32)__ffsll((long long)act) - 1u : 0u;
      const u32 x0 = __shfl(x, first);
      if (act && __ballot(same && (mine == 0 || x == x0)) == ~0ull) {
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const u32 t_ = __shfl_up(inc, d);
          if ((int)lane >= d) inc += t_;
        }
        const u32 tot = __shfl(inc, 63);
        u32 base = 0;
        if (lane == first) base = atomicAdd(&thist[x0], tot);
        u32 run = __shfl(base, first) + inc - mine;
#pragma unroll
        for (u32 q = 0; q < SORT_IPT; ++q)
          if (tid + q * SORT_THREADS < cnt) rk[q] = run++;
      } else {
#pragma unroll
        for (u32 q = 0; q < SORT_IPT; ++q)
          if (tid + q * SORT_THREADS < cnt) rk[q] = atomicAdd(&thist[lk[q]], 1u);
      }
    }
    __syncthreads();
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 2);   // tools/sort_bench: a full tile
    (void)block_exclusive_scan(thist, nb, wsum);
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 3);   // tools/sort_bench: a full tile
#pragma unroll
    for (u32 q = 0; q < SORT_IPT; ++q)
      if (tid + q * SORT_THREADS < cnt) {
        const u32 pos = thist[lk[q]] + rk[q];
        stage_v[pos] = lv[q];
        stage_k[pos] = (unsigned short)lk[q];
      }
    __syncthreads();
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 4);   // tools/sort_bench: a full tile
    if (ntile == 1) {
      for (u32 e = tid; e < cnt; e += SORT_THREADS) {
        keys_out[lo + e] = key_hi | stage_k[e];
        vals_out[lo + e] = stage_v[e];
      }
    } else {
      for (u32 e = tid; e < cnt; e += SORT_THREADS) {
        const u32 k = stage_k[e];
        const u32 dst = lo + cur[k] + (e - thist[k]);
        keys_out[dst] = key_hi | k;
        vals_out[dst] = stage_v[e];
      }
      __syncthreads();
      // advance the cursors by this tile's counts: thist[] holds the exclusive starts, the last bin ends at cnt
      for (u32 b = tid; b < nb; b += SORT_THREADS) cur[b] += (b + 1 < nb ? thist[b + 1] : cnt) - thist[b];
    }
#ifdef SORT_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (t == (ntile > 1 ? 1u : 0u)) SORT_T(2, 5);   // tools/sort_bench: a full tile
  }
}

static inline size_t sort_scatter_lds(const MsmGeom& g) { return (size_t)((g.bins + 1) & ~1u) * 4 + (size_t)g.ts * g.nwin * 8; }
static inline size_t sort_local_lds(const MsmGeom& g) { return ((size_t)2 << g.rbits) * 4 + (size_t)SORT_TILE2_PAIRS * 6; }

}  // namespace pm
