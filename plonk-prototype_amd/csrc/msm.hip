// G1 variable-base MSM (KZG commit) for gfx950.
//
// Replaces dusk_bls12_381::multiscalar_mul::msm_variable_base / pippenger
// (dusk-bls12_381 0.8, ref:Cargo.toml:20; ark-ec VariableBaseMSM::multi_scalar_mul) as called
// by dusk_plonk's CommitKey::commit (ref:Cargo.toml:19).  SURVEY.md CS-4, section 8a a8-a10.
//
// Pipeline (all on one stream, no host round trip until the window sums come back):
//   1 + 2 bucket fill  (msm_sort.hip.h) scalar -> canonical integer -> signed c-bit digits; one (key, value) pair per
//                     non-zero digit: key = window << (c-1) | (|d| - 1), value = point index | sign << 31;
//                     the pairs grouped by key with a hand-written partition + local counting sort -> every
//                     bucket's points are contiguous; the sum is order independent (exact group law)
//   3 msm_accumulate  segmented reduction with a FIXED chunk of sorted entries per thread, so
//                     load balance does not depend on the scalar distribution (all-equal
//                     scalars, the 0/1-heavy witness vectors of a real prover): runs inside a
//                     chunk go straight to their bucket, the head/tail runs of each chunk go
//                     to a partial list that the same kernel shape reduces again (XYZZ inputs)
//                     until one thread is left
//   4 msm_bucket_reduce  per bucket set: sum_b (b+1) B_b by running sums over a few buckets per lane
//                     pair, a suffix scan and a tree across the wave, and two or three small levels
//                     of the same shape (every point on two lanes: struct Half in ec.hip.h)
//   5 host            2-4 points per bucket set -> the powers of two of the reduction levels, the
//                     Horner fold with c doublings per window (none with the window table) and
//                     the final inversion on one CPU core (a serial chain of 20-300 group
//                     operations: well under 0.2 ms on the host, several ms on a single GPU lane)
//
// Differences from the reference algorithm are confined to scheduling: signed digits (half
// the buckets), a window width chosen for the GPU, sort + segmented sum instead of a serial
// bucket loop.  The result is the same group element, returned in affine-normalised form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "context.h"
#include "ec.hip.h"
#include "field_inv.hip.h"
#include "host_field.h"
#include "msm_sort.hip.h"

namespace pm {

// ------------------------------------------------------------------ bases
// ABI affine (R = 2^384 Montgomery, canonical) -> device form (R' = 2^392, canonical saturated)
__global__ void bases_convert_kernel(const u32x4* in, u32x4* out, size_t n_coords) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_coords) return;
  Fp v = fe_load<FpP>(in + 3 * i);
  fe_store<FpP>(out + 3 * i, fe_abi_to_dev<FpP>(v));
}

// Table of window multiples for a resident SRS: row w = 2^(c w) * bases.  One step doubles a row
// c times in XYZZ (pass A, one point per thread, result to scratch) and normalises back to affine
// with Montgomery's trick (pass B: `per` points per thread, taken with stride so that loads
// coalesce, one Fermat inversion of the ZZZ product per thread).
__global__ void __launch_bounds__(128) precompute_double_kernel(const u32x4* src, size_t n, u32 c, u32x4* scratch) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp x = fe_load<FpP>(src + 6 * i), y = fe_load<FpP>(src + 6 * i + 3);
  u32 nz = 0;
#pragma unroll
  for (int k = 0; k < 14; ++k) nz |= x.l[k] | y.l[k];
  Xyzz p = xyzz_identity();
  if (nz) {
    p = xyzz_double_affine(x, y);
    for (u32 k = 1; k < c; ++k) p = xyzz_double(p);
  }
  st_xyzz(scratch, i, p);
}
__global__ void __launch_bounds__(128) precompute_affine_kernel(const u32x4* scratch, size_t n, u32 per, u32x4* prefix,
                                                                u32x4* dst, u32 to_abi = 0) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Fp one = fe_one<FpP>();
  Fp acc = one;
  for (u32 j = 0; j < per; ++j) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    st_fp_limbs(prefix + 4 * i, acc);
    Fp zzz = ld_fp_limbs(scratch + 16 * i + 12);
    u32 nz = 0;
#pragma unroll
    for (int k = 0; k < 14; ++k) nz |= ld_fp_limbs(scratch + 16 * i + 8).l[k];
    if (nz) acc = fe_mul<FpP>(acc, zzz);
  }
  // 1 / acc (binary GCD, field_inv.hip.h; r01 / r02: acc^(p-2), 381 dependent squarings per thread)
  Fp inv = fe_inv_dev<FpP>(acc);
  for (u32 j = per; j-- > 0;) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) continue;
    Xyzz p = ld_xyzz(scratch, i);
    u32x4* o = dst + 6 * i;
    if (p.inf) {
      const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 6; ++k) o[k] = z;
      continue;
    }
    Fp zi = fe_mul<FpP>(inv, ld_fp_limbs(prefix + 4 * i));   // 1 / ZZZ_i
    inv = fe_mul<FpP>(inv, p.zzz);
    Fp t2 = fe_mul<FpP>(zi, p.zz);                            // ZZ / ZZZ
    Fp zzi = fe_sqr<FpP>(t2);                                 // ZZ^2 / ZZZ^2 = 1 / ZZ
    Fp ax = fe_mul<FpP>(p.x, zzi), ay = fe_mul<FpP>(p.y, zi);
    if (to_abi) {   // device form (x 2^392) -> ABI Montgomery (x 2^384) for results handed back to the caller
      ax = fe_mul<FpP>(ax, fe_pow2<FpP, 384>());
      ay = fe_mul<FpP>(ay, fe_pow2<FpP, 384>());
    }
    fe_store<FpP>(o, ax);
    fe_store<FpP>(o + 3, ay);
  }
}

// buckets[b].ZZ <- 0 (four 16-byte stores per bucket, one per thread): the identity, see ld_xyzz
__global__ void msm_clear_buckets_kernel(u32x4* buckets, size_t nb) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * 4) return;
  buckets[16 * (t >> 2) + 8 + (t & 3)] = u32x4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------ 3: segmented accumulate
struct AccArgs {
  const u32* keys;
  const u32* vals;          // level 1: point index | sign << 31
  const u32x4* pts_in;      // level >= 2: XYZZ list
  const u32x4* bases;       // level 1: affine device-form bases (96 B each)
  const u32* ctl;           // level 1: the bucket fill's control block (pairs that exist, entries per thread)
  size_t grid_threads;      // level 1: threads the host sized the partial lists for
  size_t len;               // level >= 2: entries at this level
  u32 chunk;                // level >= 2: entries per thread
  u32 offset;               // level >= 2: thread t covers [t*chunk - offset, (t+1)*chunk - offset)
  u32 trash;                // level 1: first key that is not a bucket
  u32 final_level;          // 1: every run goes to its bucket
  u32x4* buckets;           // XYZZ, indexed by key
  u32* part_keys;           // partial list written by this level: level 1 two per WAVE ([2w] the run that
  u32x4* part_pts;          // enters the wave from the left, [2w+1] the run that leaves it to the right),
                            // deeper levels two per thread; keys carry PART_OL / PART_OR
  u32x4* head_pts;          // level 1 scratch, one slot per thread: a first run that continues from the left
};
// A partial-list key = bucket key | PART_OL (the run continues from an earlier entry) | PART_OR (it
// continues in a later entry).  Entries of one run are adjacent in the list (holes aside), so a
// run is complete exactly when its first entry has no PART_OL and its last no PART_OR.
static constexpr u32 PART_OL = 0x80000000u, PART_OR = 0x40000000u, PART_KEY = 0x3fffffffu;

PM_DEV Xyzz xyzz_shfl_up1(const Xyzz& v) {
  Xyzz r;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    r.x.l[i] = __shfl_up(v.x.l[i], 1);
    r.y.l[i] = __shfl_up(v.y.l[i], 1);
    r.zz.l[i] = __shfl_up(v.zz.l[i], 1);
    r.zzz.l[i] = __shfl_up(v.zzz.l[i], 1);
  }
  r.inf = __shfl_up((int)v.inf, 1) != 0;
  return r;
}
PM_DEV Xyzz xyzz_shfl_up(const Xyzz& v, int d) {
  Xyzz r;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    r.x.l[i] = __shfl_up(v.x.l[i], d);
    r.y.l[i] = __shfl_up(v.y.l[i], d);
    r.zz.l[i] = __shfl_up(v.zz.l[i], d);
    r.zzz.l[i] = __shfl_up(v.zzz.l[i], d);
  }
  r.inf = __shfl_up((int)v.inf, d) != 0;
  return r;
}
PM_DEV Xyzz xyzz_shfl_down(const Xyzz& v, int d) {
  Xyzz r;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    r.x.l[i] = __shfl_down(v.x.l[i], d);
    r.y.l[i] = __shfl_down(v.y.l[i], d);
    r.zz.l[i] = __shfl_down(v.zz.l[i], d);
    r.zzz.l[i] = __shfl_down(v.zzz.l[i], d);
  }
  r.inf = __shfl_down((int)v.inf, d) != 0;
  return r;
}

// Joins the open pieces the lanes of one wave hold (all 64 lanes call it).  Per lane:
//   open      the lane passes something to the right: its own piece `acc` with key `cur` (a "tail",
//             which starts a segment), or -- `through` -- whatever arrives from the left plus `acc`
//   have_head the lane holds the END of a run that arrives from the left: key head_key, value in
//             memory at head_src[head_idx] (kept out of the register file until it is needed)
// A segmented scan over the lanes sums every segment; the lane with the head adds it and stores
// the bucket.  What enters the wave from the left edge goes to partial slot 2w (PART_OL), what
// leaves it to the right to slot 2w+1 (PART_OR, plus PART_OL when it also entered from the left).
// At the final level (one wave holds the whole list) nothing can cross an edge; if a malformed
// list claims otherwise the piece is stored to its bucket.
PM_DEV void wave_join(const AccArgs& a, size_t wave, bool wave_live, u32 lane, bool open, bool through, u32 cur,
                      Xyzz& acc, bool have_head, u32 head_key, const u32x4* head_src, size_t head_idx) {
  if (!open) acc = xyzz_identity();
  int flag = (open && through) ? 0 : 1;  // 1: a segment starts here (a tail, or nothing to pass on)
  for (int d = 1; d < 64; d <<= 1) {
    if (__ballot(flag == 0) == 0) break;  // wave-uniform: every segment has found its start
    const Xyzz up = xyzz_shfl_up(acc, d);
    const int fup = __shfl_up(flag, d);
    const u32 kup = __shfl_up(cur, d);
    if ((int)lane >= d && !flag) {
      acc = xyzz_add(up, acc);
      if (cur == KEY_INVALID) cur = kup;   // a hole takes the key of the chain it passes on
      flag = fup;
    }
  }
  // acc (open lanes): sum of the run from its start inside the wave -- or from the wave's left
  // edge when flag is still 0 -- up to this lane
  const Xyzz from_up = xyzz_shfl_up(acc, 1);
  const bool up_open = (__shfl_up((int)open, 1) != 0) && lane != 0;
  const bool up_started = (__shfl_up(flag, 1) != 0) && lane != 0;
  const bool wave_head = have_head && !(up_open && up_started);  // the run entered the wave from the left
  const bool wave_tail = open && (lane == 63u) && cur != KEY_INVALID;
  const bool any_head = __ballot(wave_head) != 0;
  if (wave_tail) {  // leaves the wave to the right
    if (a.final_level) {
      st_xyzz(a.buckets, cur, acc);
    } else {
      a.part_keys[2 * wave + 1] = cur | PART_OR | (flag ? 0u : PART_OL);
      st_xyzz(a.part_pts, 2 * wave + 1, acc);
    }
  } else if (lane == 63u && wave_live && !a.final_level) {
    a.part_keys[2 * wave + 1] = KEY_INVALID;
  }
  if (lane == 0u && !any_head && wave_live && !a.final_level) a.part_keys[2 * wave] = KEY_INVALID;
  // (acc is dead from here on: the merge below needs the registers)
  if (have_head) {
    Xyzz h = ld_xyzz(head_src, head_idx);
    if (up_open) h = xyzz_add(from_up, h);
    if (wave_head && !a.final_level) {
      a.part_keys[2 * wave] = head_key | PART_OL;
      st_xyzz(a.part_pts, 2 * wave, h);
    } else {
      st_xyzz(a.buckets, head_key, h);
    }
  }
}

// Level 1: sorted (key, point index) pairs, affine bases, one thread per chunk of `chunk`
// entries.  A run is classified exactly by looking one key past either end of the chunk:
//   neither continues    -> complete, stored to its bucket
//   first run that continues from the left and ends inside the chunk -> "head", parked in head_pts[t]
//   last run that continues to the right -> stays in registers: "tail" if it started inside the
//                           chunk, "through" if the whole chunk is one run open on both sides
// The open pieces of a wave are then joined in the register file by a segmented scan over the
// lanes: a tail starts a segment, through lanes extend it, and the lane holding the head of the
// same run adds its parked head and stores the bucket.  For uniform digits with chunks of a few
// run lengths a run is split over exactly two neighbouring lanes and the scan degenerates to one
// shuffle and one addition (no through lanes: the loop below does not iterate); long runs (small
// n, skewed or all-equal scalars) cost up to six additions per wave instead of a pass through a
// per-thread partial list.  Only what crosses a WAVE boundary is left for the next level: two
// partial slots per wave.
__global__ void __launch_bounds__(256, 2) msm_accumulate_l1_kernel(const AccArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the number of pairs and the chunk length come from the bucket fill (zero digits are dropped there): the host
  // sized the grid for every digit being non-zero, sparser inputs use shorter chunks on the same grid
  const size_t len = ld_uniform(a.ctl + CTL_M_EFF), chunk = ld_uniform(a.ctl + CTL_CHUNK);
  const size_t nthreads = (len + chunk - 1) / chunk;
  const bool active = t < nthreads;
  const u32 lane = threadIdx.x & 63u;
  const size_t wave = t >> 6;
  const size_t lo = t * chunk, hi = lo + chunk < len ? lo + chunk : len;
  u32 prev_key = KEY_INVALID, next_key = KEY_INVALID;
  if (active) {
    if (lo > 0) prev_key = a.keys[lo - 1];
    if (hi < len) next_key = a.keys[hi];
    if (next_key >= a.trash) next_key = KEY_INVALID;
  }
  u32 cur = KEY_INVALID, head_key = KEY_INVALID;
  bool first_run = true, have_head = false, open = false, through = false;
  Xyzz acc = xyzz_identity();
  if (active) {
    // software pipeline: the (key, value, point) of entry e+1 is requested before the ~5000
    // instructions of entry e's addition, so the random 96-byte gather is never waited for
    u32 nk = a.keys[lo], nv = a.vals[lo];
    u32x4 nraw[6];
    {
      const u32x4* bp = a.bases + 6 * (size_t)(nv & 0x7fffffffu);
#pragma unroll
      for (int i = 0; i < 6; ++i) nraw[i] = bp[i];
    }
    for (size_t e = lo; e < hi; ++e) {
      const u32 k = nk, v = nv;
      u32x4 raw[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) raw[i] = nraw[i];
      if (e + 1 < hi) {
        nk = a.keys[e + 1];
        nv = a.vals[e + 1];
        const u32x4* bp = a.bases + 6 * (size_t)(nv & 0x7fffffffu);
#pragma unroll
        for (int i = 0; i < 6; ++i) nraw[i] = bp[i];
      }
      if (k >= a.trash) break;  // sorted: only zero digits from here on
      if (k != cur) {
        if (cur != KEY_INVALID) {  // a run that ends inside the chunk
          const bool cfb = first_run && cur == prev_key;
          if (a.final_level || !cfb) {
            st_xyzz(a.buckets, cur, acc);
          } else {  // head continuing from the previous chunk: park it
            st_xyzz(a.head_pts, t, acc);
            have_head = true;
            head_key = cur;
          }
          first_run = false;
        }
        cur = k;
        acc = xyzz_identity();
      }
      u32 sx[12], sy[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        sx[4 * i] = raw[i].x; sx[4 * i + 1] = raw[i].y; sx[4 * i + 2] = raw[i].z; sx[4 * i + 3] = raw[i].w;
        sy[4 * i] = raw[3 + i].x; sy[4 * i + 1] = raw[3 + i].y; sy[4 * i + 2] = raw[3 + i].z; sy[4 * i + 3] = raw[3 + i].w;
      }
      Fp x = fe_unpack<FpP>(sx), y = fe_unpack<FpP>(sy);
      u32 nz = 0;
#pragma unroll
      for (int i = 0; i < 14; ++i) nz |= x.l[i] | y.l[i];
      if (nz == 0) continue;  // the point at infinity among the bases
      if (v >> 31) y = fe_sub<FpP, 2, 1>(fe_zero<FpP>(), y);  // -y as 2p - y   (3, <2)
      acc = xyzz_madd(acc, x, y);
    }
    if (cur != KEY_INVALID) {  // the last run: it ends at the chunk end
      const bool cfb = first_run && cur == prev_key;
      const bool ca = cur == next_key;
      if (a.final_level || (!cfb && !ca)) {
        st_xyzz(a.buckets, cur, acc);
      } else if (cfb && !ca) {
        st_xyzz(a.head_pts, t, acc);
        have_head = true;
        head_key = cur;
      } else {
        open = true;      // stays in registers
        through = cfb;    // ... and also continues from the left
      }
    }
  }
  if (a.final_level) return;  // a single thread: nothing can be open
  wave_join(a, wave, (wave << 6) < a.grid_threads, lane, open, through, cur, acc, have_head, head_key, a.head_pts, t);
}

// Level >= 2: a list of (key | PART_OL | PART_OR, XYZZ) partials in key order, with holes
// (KEY_INVALID): ONE entry per lane (shifted by one slot, so that the slot a run leaves wave w
// through, 2w+1, and the slot it enters wave w+1 through, 2w+2, sit in neighbouring lanes of one
// wave), joined by the same segmented scan.  An entry with PART_OR only starts a segment, PART_OL
// only ends one, both continues one; a hole passes the chain on.  Every level shrinks the list by
// 32, and a run of any length costs at most six additions per level.
__global__ void __launch_bounds__(128) msm_accumulate_ln_kernel(const AccArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = a.len + 1;
  const u32 lane = threadIdx.x & 63u;
  const size_t wave = t >> 6;
  const bool active = t >= 1 && t < nthreads;
  const size_t e = t - 1;
  const u32 k = active ? a.keys[e] : KEY_INVALID;
  const bool hole = (k == KEY_INVALID);
  const u32 key = hole ? KEY_INVALID : (k & PART_KEY);
  const bool ol = !hole && (k & PART_OL), orr = !hole && (k & PART_OR);
  // holes inside the list pass a chain on (through, nothing of their own); lanes outside the list end it
  const bool open = active && (hole || orr);
  const bool through = active && (hole || (ol && orr));
  const bool have_head = ol && !orr;
  Xyzz acc = xyzz_identity();
  if (!hole && orr) acc = ld_xyzz(a.pts_in, e);
  if (!hole && !ol && !orr) st_xyzz(a.buckets, key, ld_xyzz(a.pts_in, e));  // (not produced by a well-formed level)
  wave_join(a, wave, (wave << 6) < nthreads, lane, open, through, key, acc, have_head, key, a.pts_in, e);
}

// ------------------------------------------------------------------ 4: bucket reduce
// sum_b (b + 1) B_b for every bucket set (SURVEY CS-4: "running = 0; for b in buckets.rev() { running += b; acc += running }"),
// as a hierarchy of wave-level reductions whose index weights are never multiplied out on the device.  Every point is
// held by a PAIR of lanes (struct Half in ec.hip.h: 7 field-product rounds per addition instead of 14).
//
// What shaped it (r04, profiles/r04_bucket_reduce_stamps.txt -- in-kernel clock stamps of the r03 kernel): two waves that
// share a SIMD do NOT share its issue slots -- the older wave runs at the speed of a lone wave (~5 cycles per VALU
// instruction for this code, whatever the number of dependency chains), the younger one gets ~10 % and only starts
// properly when the older has left; and single-wave workgroups are not spread evenly over the SIMDs.  So the reduction
// runs ONE wave per SIMD, placed exactly (four-wave workgroups, one per CU: a workgroup's waves go to the CU's four
// SIMDs, and its LDS request keeps a second workgroup out), and what a wave does after its buckets is five dependent
// additions instead of fifteen:
//   level 1 (msm_bucket_reduce_kernel): a lane pair takes `lb` consecutive buckets with the running sums
//       S = sum B,  T = sum (i + 1) B_i,  so its chunk t = 32 w + l contributes T + lb t S.
//       The 32 pairs of the wave are then combined by a BUTTERFLY through LDS (wave_butterfly): after step j every
//       block of 2^j lanes holds  A = sum S,  Tt = sum T  and the BIT PLANES  p_k = sum_{l : bit k of l} S_l  (k < j) --
//       merging two blocks is one addition per value, all of them on different lane pairs of the block (j + 1 <= 2^j),
//       and the new plane p_(j-1) is the right block's A: a copy.  Five steps leave  W_w = sum_l S_l,  Tt_w  and
//       p_0 .. p_4 with  sum_l l S_l = sum_k 2^k p_k  -- no scan, no doubling, no second tree.
//       The set's total is  sum_w Tt_w + lb sum_k 2^k sum_w p_{k,w} + 32 lb sum_w w W_w.
//   level k >= 2 (msm_reduce_level_kernel): 32 consecutive items per wave.  Sequences that only need adding up (Tt and
//       every plane of the earlier levels) get a 5-step tree each; the sequence W goes through the same butterfly and
//       yields five more planes (the next five bits of the chunk index) and W'.  Independent waves take the jobs.
//   The host receives  Tt,  the planes Q_0 .. Q_(5 L - 1)  (bit b of the chunk index) and at most four items of the last
//   W, and evaluates  Tt + lb (sum_b 2^b Q_b + 32^L sum_l l W_l)  by Horner's rule: a doubling costs well under 1 us there.
static constexpr u32 GROUP = 32;                 // items per wave in the reduction: one per lane PAIR (ec.hip.h, struct Half)
static constexpr u32 PLANES = 5;                 // log2 GROUP: bit planes one butterfly leaves
static constexpr u32 RED_SLOTS = 2 * GROUP + 48; // 256-byte LDS records per wave: the butterfly's two buffers (64 + 16 x 3)
static constexpr u32 RED_WAVES = 4;              // waves per workgroup in level 1: one per SIMD of the CU

// LDS traffic of ONE wave: its DS operations execute in order, so a read sees the wave's earlier writes; the fences
// only keep the compiler from moving them across the step boundary.
PM_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// On entry lane pair `pi` holds S (and T when HAS_T).  Returns the buffer whose records 0 .. 6 hold
//   W = sum S,  Tt = sum T,  p_0 .. p_4  (p_k = sum of S over the lane pairs whose bit k is set).
// State after step j, in LDS: block b (2^j lane pairs) owns records b (j + 2) .. : A, Tt, p_0 .. p_(j-1).
template <bool HAS_T>
PM_DEV const u32x4* wave_butterfly(u32x4* lds, u32 pi, bool isB, const Half& S, const Half& T) {
  st_half(lds, 2 * pi, S, isB);
  st_half(lds, 2 * pi + 1, T, isB);
  wave_lds_sync();
  // NOT unrolled: one copy of the addition serves the five steps (unrolled, the steps are five 50 KB stretches of code that
  // every wave runs once -- 400 KB against a 64 KB instruction cache: 12.4 us per step instead of the 8.8 us of an addition)
#pragma unroll 1
  for (u32 j = 1; j <= PLANES; ++j) {
    const u32x4* src = (j - 1) & 1 ? lds + 2 * GROUP * 16 : lds;
    u32x4* dst = j & 1 ? lds + 2 * GROUP * 16 : lds;
    const u32 b = pi >> j, o = pi & ((1u << j) - 1u);
    const u32 nv = j + 1;                                   // values per child block
    const u32 L = 2 * b * nv, R = L + nv, D = b * (nv + 1);
    if (o < nv) {
      if (HAS_T || o != 1) {
        const Half x = ld_half(src, L + o, isB), y = ld_half(src, R + o, isB);
        st_half(dst, D + o, half_add(x, y, isB), isB);
      } else {
        st_half(dst, D + o, half_identity(), isB);
      }
    }
    if (o == (j == 1 ? 0u : nv)) st_half(dst, D + nv, ld_half(src, R, isB), isB);   // p_(j-1) = A of the right block
    wave_lds_sync();
  }
  return PLANES & 1 ? lds + 2 * GROUP * 16 : lds;
}
// butterfly record -> output sequence: sequences are [Tt | planes ... | W]; `first_plane` = index of this level's p_0
PM_DEV u32 red_seq_of(u32 rec, u32 first_plane, u32 w_seq) { return rec == 0 ? w_seq : (rec == 1 ? 0u : first_plane + rec - 2); }

// Diagnostic build only (make EXTRA=-DPM_DEV_STAMPS, tools/ab_reduce.sh): every wave leaves shader-clock and
// wall-clock stamps at its phase boundaries, from which the host prints the waves' start / end skew, the phase
// lengths and the in-kernel clock.  The product build has none of it.
#ifdef PM_DEV_STAMPS
__device__ unsigned long long* g_bucket_stamps = nullptr;
#define BUCKET_STAMP(k)                                                                       \
  if (g_bucket_stamps && (threadIdx.x & 63u) == 0) {                                          \
    g_bucket_stamps[(size_t)widx * 16 + 2 * (k)] = __builtin_amdgcn_s_memtime();              \
    g_bucket_stamps[(size_t)widx * 16 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();      \
  }
#else
#define BUCKET_STAMP(k)
#endif
// out: 7 sequences of n_waves records ([Tt | p_0 .. p_4 | W], seq_stride u32x4 apart), record index = the wave's index
__global__ void __launch_bounds__(64 * RED_WAVES, 1)
    msm_bucket_reduce_kernel(const u32x4* buckets, u32 nbuckets, u32 lb, u32 n1, u32 n_waves, u32 finalize, u32x4* out,
                             size_t seq_stride) {
  extern __shared__ u32x4 red_lds[];
  const u32 wv = threadIdx.x >> 6;
  const u32 widx = blockIdx.x * RED_WAVES + wv;   // this wave's item: (set, w)
  if (widx >= n_waves) return;                    // whole waves only, and no workgroup barrier below
  BUCKET_STAMP(0);
  const u32 set = widx / n1, w = widx % n1;
  const u32 pi = (threadIdx.x & 63u) >> 1;        // the pair's index in the wave: 32 chunks per wave
  const bool isB = threadIdx.x & 1;
  const u32 nt = nbuckets / lb;                   // chunks per set
  const u32 t = GROUP * w + pi;
  Half running = half_identity(), sum = half_identity();
  if (t < nt) {
    const u32x4* B = buckets + (size_t)16 * ((size_t)set * nbuckets + (size_t)t * lb);
    for (u32 i = lb; i-- > 0;) {
      Half b = ld_half(B, i, isB);
      running = half_add(running, b, isB);
      sum = half_add(sum, running, isB);
    }
  }
  BUCKET_STAMP(1);
  const u32x4* res = wave_butterfly<true>(red_lds + (size_t)wv * RED_SLOTS * 16, pi, isB, running, sum);
  BUCKET_STAMP(2);
  if (pi < PLANES + 2) {
    Half v = ld_half(res, pi, isB);
    if (finalize && !v.inf) v.c0 = fe_mul<FpP>(v.c0, fe_one<FpP>());   // the host needs x, y that fit 384 bits
    st_half(out + red_seq_of(pi, 1, PLANES + 1) * seq_stride, widx, v, isB);
  }
  BUCKET_STAMP(3);
}

// in: n_seq_in sequences of n_items records per set ([Tt | planes | W]); out: n_seq_in + 5 sequences of n_groups records
struct RedLevelArgs {
  const u32x4* in;
  u32x4* out;
  size_t in_stride, out_stride;   // u32x4 per sequence
  u32 n_seq_in;
  u32 n_items;
  u32 n_groups;                   // ceil(n_items / GROUP)
  u32 finalize;
};
__global__ void __launch_bounds__(64, 2) msm_reduce_level_kernel(const RedLevelArgs a) {
  __shared__ u32x4 lds[RED_SLOTS * 16];
  const u32 set = blockIdx.x / a.n_groups, g = blockIdx.x % a.n_groups, job = blockIdx.y;
  const u32 pi = threadIdx.x >> 1;
  const bool isB = threadIdx.x & 1;
  const u32 idx = GROUP * g + pi;
  const size_t src = (size_t)set * a.n_items + idx;
  Half acc = half_identity();
  if (idx < a.n_items) acc = ld_half(a.in + job * a.in_stride, src, isB);
  if (job + 1 < a.n_seq_in) {   // a sequence that only needs adding up
#pragma unroll 1   // one copy of the addition in the instruction cache (see wave_butterfly)
    for (int d = GROUP / 2; d > 0; d >>= 1) {   // pairs >= d would add their own value to itself (the slow P + P path)
      Half o = half_shfl_down(acc, d);
      if (pi < (u32)d) acc = half_add(acc, o, isB);
    }
    if (pi == 0) {
      if (a.finalize && !acc.inf) acc.c0 = fe_mul<FpP>(acc.c0, fe_one<FpP>());
      st_half(a.out + job * a.out_stride, blockIdx.x, acc, isB);
    }
    return;
  }
  // the sequence W: five more planes of the chunk index, and W'
  const u32x4* res = wave_butterfly<false>(lds, pi, isB, acc, half_identity());
  if (pi < PLANES + 2 && pi != 1) {
    Half v = ld_half(res, pi, isB);
    if (a.finalize && !v.inf) v.c0 = fe_mul<FpP>(v.c0, fe_one<FpP>());
    st_half(a.out + red_seq_of(pi, a.n_seq_in - 1, a.n_seq_in + PLANES - 1) * a.out_stride, blockIdx.x, v, isB);
  }
}

// What the host fold would do per set -- Tt + lb (sum_b 2^b Q_b + 32^L sum_l l W_l) -- for launches with MANY sets (a
// table-free MSM has one set per window, a batch one per vector and window): one wave per set, one lane pair per term.
// Pair q sums its sequence over the host_items items (the W sequence appears twice, once per bit of the item index: its
// two bit planes), doubles the sum e(q) times -- every pair in lock step, the pairs with small exponents idle -- and the
// terms are added up by a 5-step tree.  A chain of ~3 + e_max + 5 operations (~0.15-0.2 ms) whatever the number of
// sets, against ~40 us of host arithmetic per set (host additions ~1 us, doublings ~0.65 us: tools/host_field_bench.cpp).
__global__ void __launch_bounds__(64, 2) msm_reduce_finish_kernel(const u32x4* in, size_t in_stride, u32 n_sets, u32 host_items,
                                                                  u32 nplanes, u32 log_lb, u32x4* out) {
  const u32 set = blockIdx.x;
  const u32 pi = threadIdx.x >> 1;
  const bool isB = threadIdx.x & 1;
  // term q: 0 = Tt (weight 1), 1 .. nplanes = plane q - 1 (weight lb 2^(q-1)), then bit planes 0 and 1 of the W items
  u32 seq = 0, mask = 0, e = 0;
  const u32 all = (1u << host_items) - 1u;
  if (pi == 0) {
    mask = all;
  } else if (pi <= nplanes) {
    seq = pi, mask = all, e = log_lb + pi - 1;
  } else if (pi <= nplanes + 2 && host_items > 1) {
    const u32 bit = pi - nplanes - 1;
    seq = nplanes + 1, mask = all & (bit == 0 ? 0xAu : 0xCu), e = log_lb + nplanes + bit;
  }
  Half acc = half_identity();
#pragma unroll 1
  for (u32 l = 0; l < host_items; ++l)
    if ((mask >> l) & 1u) acc = half_add(acc, ld_half(in + seq * in_stride, (size_t)set * host_items + l, isB), isB);
  const u32 e_max = log_lb + nplanes + 1;
#pragma unroll 1
  for (u32 k = 0; k < e_max; ++k)
    if (k < e) acc = half_double(acc, isB);
#pragma unroll 1
  for (int d = GROUP / 2; d > 0; d >>= 1) {
    Half o = half_shfl_down(acc, d);
    if (pi < (u32)d) acc = half_add(acc, o, isB);
  }
  if (pi == 0) {
    if (!acc.inf) acc.c0 = fe_mul<FpP>(acc.c0, fe_one<FpP>());   // the host needs x, y that fit 384 bits
    st_half(out, set, acc, isB);
  }
}

// ------------------------------------------------------------------ host-side group law
using host::HFp;
using host::XYZZ;

// device limbs (14 x 28 bit, value < 2^384, Montgomery R' = 2^392) -> host Montgomery (R = 2^384)
static HFp limbs_to_host(const u32* l) {
  HFp v = host::zero<6>();
  u64 nl[14], carry = 0;
  for (int i = 0; i < 14; ++i) {  // full carry so the 28-bit fields cannot overlap
    u64 t = (u64)l[i] + carry;
    nl[i] = (i == 13) ? t : (t & 0xfffffffull);
    carry = t >> 28;
  }
  for (int i = 0; i < 14; ++i) {
    const int lo = 28 * i, j = lo / 64, sh = lo % 64;
    v.l[j] |= nl[i] << sh;
    if (sh + 28 > 64 && j + 1 < 6) v.l[j + 1] |= nl[i] >> (64 - sh);
  }
  HFp c = host::zero<6>();
  c.l[376 / 64] = (u64)1 << (376 % 64);  // x * 2^392 * 2^376 / 2^384 = x * 2^384
  return host::mul(v, c, host::FP());
}
static XYZZ xyzz_to_host(const u32* p) {  // p: 64 words
  XYZZ r;
  bool z = true;
  for (int i = 0; i < 14; ++i) z = z && p[32 + i] == 0;
  if (z) return host::xyzz_identity();
  r.x = limbs_to_host(p);
  r.y = limbs_to_host(p + 16);
  r.zz = limbs_to_host(p + 32);
  r.zzz = limbs_to_host(p + 48);
  return r;
}
static void write_projective(uint64_t out[18], const XYZZ& p) {
  HFp x, y;
  memset(out, 0, 18 * 8);
  if (!host::xyzz_to_affine(p, x, y)) {
    memcpy(out + 6, host::FP().one, 48);  // (0, 1, 0)
    return;
  }
  memcpy(out, x.l, 48);
  memcpy(out + 6, y.l, 48);
  memcpy(out + 12, host::FP().one, 48);
}
// the same for k points with ONE field inversion (Montgomery's trick over the ZZZ coordinates): a host inversion
// is ~50 us, a prover round's batch has four results
static void write_projective_batch(uint64_t* out /* k x 18 */, const XYZZ* p, size_t k) {
  const host::Field<6>& F = host::FP();
  std::vector<HFp> pre(k);
  HFp acc = host::one(F);
  for (size_t i = 0; i < k; ++i) {
    pre[i] = acc;
    if (!host::is_zero(p[i].zz)) acc = host::mul(acc, p[i].zzz, F);
  }
  HFp inv = host::inv(acc, F);
  memset(out, 0, k * 18 * 8);
  for (size_t i = k; i-- > 0;) {
    uint64_t* o = out + 18 * i;
    if (host::is_zero(p[i].zz)) {
      memcpy(o + 6, F.one, 48);  // (0, 1, 0)
      continue;
    }
    const HFp zi = host::mul(inv, pre[i], F);                       // 1 / ZZZ_i
    inv = host::mul(inv, p[i].zzz, F);
    const HFp t = host::mul(zi, p[i].zz, F);                        // ZZ / ZZZ
    const HFp x = host::mul(p[i].x, host::mul(t, t, F), F);         // X / ZZ
    const HFp y = host::mul(p[i].y, zi, F);
    memcpy(o, x.l, 48);
    memcpy(o + 6, y.l, 48);
    memcpy(o + 12, F.one, 48);
  }
}
static XYZZ projective_to_xyzz(const uint64_t* xyz) {  // homogeneous (X/Z, Y/Z)
  const host::Field<6>& F = host::FP();
  HFp X, Y, Z;
  memcpy(X.l, xyz, 48);
  memcpy(Y.l, xyz + 6, 48);
  memcpy(Z.l, xyz + 12, 48);
  if (host::is_zero(Z)) return host::xyzz_identity();
  XYZZ r;  // x = X/Z = (X Z)/Z^2, y = Y/Z = (Y Z^2)/Z^3
  HFp z2 = host::mul(Z, Z, F);
  r.x = host::mul(X, Z, F);
  r.y = host::mul(Y, z2, F);
  r.zz = z2;
  r.zzz = host::mul(z2, Z, F);
  return r;
}

// ------------------------------------------------------------------ driver
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// PM_MSM_DEBUG=1 in the environment: synchronise after every stage and name it on stderr (localises a device fault)
static bool msm_debug() {
  static const bool on = getenv("PM_MSM_DEBUG") != nullptr;
  return on;
}
#define MSM_STAGE(ctx, st, name)                                                \
  do {                                                                          \
    if (msm_debug()) {                                                          \
      fprintf(stderr, "[msm] %s ...", name);                                    \
      PM_HIP(ctx, hipStreamSynchronize(st));                                    \
      fprintf(stderr, " done\n");                                               \
    }                                                                           \
  } while (0)

// What the host needs to finish one piece of an MSM call (a sub-batch that went through the kernels on its own).
struct MsmPiece {
  u32 batch, nsets, c, n_dev, host_items, log_lb, nsets_all;
  bool finished;        // the device applied the weights (msm_reduce_finish_kernel): ONE record per set
  const u32* hw;        // pinned host memory: (7 + 5 n_dev) sequences x nsets_all x host_items XYZZ records
};
// One piece: every kernel of the pipeline plus the copy of its (7 + 5 n_dev) x nsets_all x host_items result points, enqueued on
// `st` -- no host synchronisation.  ws == nullptr: nothing is launched, only *need_ws / *need_pinned are set (bytes of
// device workspace and of pinned host memory a piece of this shape takes).  front_done (optional) is recorded after the
// last accumulate level: from there on the piece only reads its own buckets (not the control block of the bucket fill).
static int msm_piece(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n, const void* d_scalars, size_t sc_stride,
                     u32 batch, u32 scalar_form, hipStream_t st, hipEvent_t front_done, char* ws, size_t* need_ws,
                     void* pinned, size_t* need_pinned, MsmPiece* piece) {
  MsmGeom g = make_geom(n, ctx->opt_msm_window_bits, bases->table_c, bases->n, batch);
  if (g.bins > SORT_MAX_BINS || g.rbits > SORT_MAX_RBITS || g.ts == 0)
    return set_err(ctx, PM_ERR_BAD_ARG, "window width outside what the bucket fill is laid out for");
  const size_t m = n * batch * g.nwin;  // (key, value) pairs at most: one per non-zero digit
  const u32 nsets_all = g.nsets * batch;
  // Entries per thread in the big kernel (measured, profiles/r01_msm_sweep.txt, r02_msm_sweep.txt): about
  // four times the mean run length m / #buckets when the grid allows it (then a run is split over at
  // most two neighbouring lanes and the in-wave join costs one addition), at least 128, and never so
  // long that the grid drops below 2^17 threads; small inputs end up with chunks shorter than a run
  // and pay up to six additions per wave in the segmented scan instead.
  // The kernel holds two waves per SIMD, i.e. `slots` threads at a time, and every thread does the same work: a grid
  // of 3.25 x slots threads (a batch of four MSMs with 128-entry chunks) runs as four rounds, the last one a quarter
  // full.  So the grid is a whole number of rounds and the chunk follows from it (batch of four: 104 entries, as for
  // a single MSM; measured: 2.55 -> 2.3 ms of accumulate per MSM in a batch).
  const size_t avg_run = std::max<size_t>(1, m / ((size_t)g.nbuckets * nsets_all));
  const size_t slots = (size_t)ctx->num_cus * 4 * 2 * 64;
  const size_t want = std::max<size_t>(128, 4 * avg_run);
  const size_t rounds = std::max<size_t>(1, (m + slots * want - 1) / (slots * want));
  // Small inputs (the commit rounds of a 2^10 .. 2^14-gate circuit) fill a fraction of one round and the kernel's time is
  // the length of one thread's chain: the chunk goes down to 12 entries, and to 4 where the runs are that short -- below
  // ~0.6 of a run the in-wave join pays for what the chain saves (profiles/r03_small_msm.txt: batch of four at 2^14,
  // accumulate 434 -> 365 us; at 2^10, 185 -> 108 us)
  u32 chunk_lo = (u32)std::min<size_t>(12, std::max<size_t>(4, (6 * avg_run + 9) / 10));
  u32 L1 = ctx->opt_msm_chunk ? (u32)ctx->opt_msm_chunk
                              : (u32)std::max<size_t>(chunk_lo, (m + rounds * slots - 1) / (rounds * slots));
  // r05: while the whole grid fits ONE wave per SIMD -- placed exactly, below -- a thread's time is its chain: chunk mixed
  // additions at a lone wave's ~11.8 us each, then the in-wave join, one general addition (~16.5 us) per doubling of the lanes
  // a run is spread over.  The chunk that minimises that sum (profiles/r05_small_msm.txt: 2^12 points, one vector: 12 -> 5
  // entries, accumulate 174 -> 131 us)
  const size_t lone_waves = (size_t)ctx->num_cus * 4;
  if (!ctx->opt_msm_chunk) {
    double best = 1e30;
    for (u32 L = 2; L <= 16; ++L) {
      const size_t waves = ((m + L - 1) / L + 63) / 64;
      if (waves > lone_waves) continue;
      const double span = (double)avg_run / L + 1.0;
      const double cost = 11.8 * L + 16.5 * std::ceil(std::log2(span));
      if (cost < best) {
        best = cost;
        L1 = L;
        chunk_lo = std::min<u32>(chunk_lo, L);
      }
    }
  }
  // Partial lists: every level leaves two slots per WAVE; the deeper levels take one slot per lane, so
  // the list shrinks by 32 per level and ends in a single wave (final level).
  struct Level {
    size_t len;   // entries read at this level
    u32 chunk, offset;
  };
  std::vector<Level> lv;
  lv.push_back({m, L1, 0});
  {
    size_t nthr = (m + L1 - 1) / L1;
    while (nthr > 1) {
      const size_t len = 2 * ((nthr + 63) / 64);
      lv.push_back({len, 1, 1});
      nthr = len + 1;            // one lane per slot, shifted by one
      if (nthr <= 64) break;     // a single wave: final
    }
  }
  const size_t l1_threads = (m + L1 - 1) / L1;
  const size_t total_buckets = (size_t)g.nbuckets * nsets_all;
  // Buckets per lane pair in level 1 of the reduction (a power of two).  A pair does 2 LB + 5 group operations and the
  // kernel runs ONE wave per SIMD (section 4 above): waves beyond 4 per CU queue for a second round.  Every further
  // level is a launch of ~5 dependent operations, and the host fold pays per sequence and item it receives.  LB is the
  // candidate with the smallest estimate of the three together (us; the constants are measured: profiles/r04_small_msm.txt
  // -- 2^19 buckets: 16, a batch of four: 64, an 8-way shard's 2^15: 1, four sets of 2^12: 4).
  // The host takes over when at most HOST_ITEMS items per set are left: a launch that folds two or three items is a
  // ~55 us chain on one wave, the same fold is a handful of additions (~1 us each) in the host fold below.
  constexpr u32 HOST_ITEMS = 4;
  struct RedPlan {
    u32 lb, n1, host_items;
    std::vector<u32> groups;
    double est;
  };
  auto red_plan = [&](u32 lb) {
    RedPlan p;
    p.lb = lb;
    p.n1 = (g.nbuckets / lb + GROUP - 1) / GROUP;   // waves per set in level 1
    p.host_items = p.n1;
    while (p.host_items > HOST_ITEMS) {
      p.host_items = (p.host_items + GROUP - 1) / GROUP;
      p.groups.push_back(p.host_items);
    }
    const double waves = (double)p.n1 * nsets_all, slots = (double)ctx->num_cus * RED_WAVES;
    const double rounds = std::ceil(waves / slots), n_seq = (PLANES + 2) + PLANES * (double)p.groups.size();
    p.est = rounds * (2.0 * lb + 6.0) * 9.0 + 65.0 * (double)p.groups.size() +
            (double)nsets_all * n_seq * (0.35 * p.host_items + 0.7 * (p.host_items - 1));
    return p;
  };
  RedPlan plan = red_plan(1);
  if (ctx->opt_msm_lb) {
    plan = red_plan(std::min<u32>((u32)ctx->opt_msm_lb, g.nbuckets));
  } else {
    for (u32 lb = 2; lb <= 256 && lb <= g.nbuckets; lb *= 2) {
      RedPlan p = red_plan(lb);
      if (p.est < plan.est) plan = p;
    }
  }
  const u32 LB = plan.lb;
  u32 log_lb = 0;
  while ((1u << log_lb) < LB) ++log_lb;
  const u32 n1 = plan.n1, host_items = plan.host_items;
  const std::vector<u32>& lvl_groups = plan.groups;   // group counts of the follow-up levels
  const u32 n_dev = (u32)lvl_groups.size();   // follow-up launches
  if (n_dev > 3) return set_err(ctx, PM_ERR_BAD_ARG, "internal: bucket reduction deeper than four levels");
  // what the host receives: 7 + 5 n_dev sequences (Tt, five planes per level, W) of host_items entries per set
  const u32 n_seq_host = (PLANES + 2) + PLANES * n_dev;

  // workspace layout
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  // bucket fill: integer scalars + per-tile count rows (histogram -> scatter), partitioned pairs, sorted keys / values
  const size_t o_canon = take((size_t)n * batch * 32), o_rows = take((size_t)batch * g.tiles * g.bins * 2);
  const size_t o_pairs = take(m * 8), o_keys1 = take(m * 4 + 4), o_vals1 = take(m * 4 + 4);
  const size_t o_buckets = take(total_buckets * 256);
  std::vector<size_t> o_pkeys(lv.size()), o_ppts(lv.size());
  for (size_t i = 1; i < lv.size(); ++i) {
    o_pkeys[i] = take(lv[i].len * 4);
    o_ppts[i] = take(lv[i].len * 256);
  }
  const size_t o_heads = take(lv.size() > 1 ? l1_threads * 256 : 0);
  // level 1 writes 7 sequences of one record per wave, every further level 5 sequences more of one record per group;
  // the last one writes to o_win (sequence-major: nsets_all x host_items records per sequence), which the host reads
  const size_t o_l1 = take((size_t)(PLANES + 2) * nsets_all * n1 * 256);
  std::vector<size_t> o_lvl(n_dev);
  for (u32 k = 0; k < n_dev; ++k) o_lvl[k] = take((size_t)((PLANES + 2) + PLANES * (k + 1)) * nsets_all * lvl_groups[k] * 256);
  const size_t o_win = take((size_t)n_seq_host * nsets_all * host_items * 256);
  // Who applies the weights: the host fold (~1 us per addition, ~0.65 us per doubling, per set) or one more launch
  // (msm_reduce_finish_kernel: a fixed chain whatever the number of sets).  One or a few sets: the host.
  const u32 nplanes = PLANES * (n_dev + 1);
  const double fold_host_us = (double)nsets_all * (n_seq_host * (0.3 * host_items + 1.0 * (host_items - 1)) + 1.65 * nplanes);
  const double fold_dev_us = (host_items + 0.72 * (log_lb + nplanes + 1) + 5.0) * 9.0 + 20.0 + 0.5 * nsets_all;
  const bool use_finish = fold_host_us > fold_dev_us;
  const size_t o_fin = take(use_finish ? (size_t)nsets_all * 256 : 0);
  const size_t pinned_bytes = use_finish ? (size_t)nsets_all * 256 : (size_t)n_seq_host * nsets_all * host_items * 256;
  if (!ws) {
    *need_ws = off;
    *need_pinned = pinned_bytes;
    return PM_OK;
  }
  u32 *keys1 = (u32*)(ws + o_keys1), *vals1 = (u32*)(ws + o_vals1);
  g.ctl_cap = ctx->msm_ctl_cap;   // the block's layout is fixed per allocation, whatever this MSM's partition count
  u32* ctl = (u32*)ctx->msm_ctl.ptr;
  u32x4* buckets = (u32x4*)(ws + o_buckets);

  // 1 + 2 bucket fill (msm_sort.hip.h)
  {
    u32 tiles_per_wg = 1;   // at most ~2 K workgroups: the partition totals cost one atomic per (workgroup-tile, partition)
    while ((size_t)batch * ((g.tiles + tiles_per_wg - 1) / tiles_per_wg) > 2048) ++tiles_per_wg;
    const u32 wgs_per_msm = (g.tiles + tiles_per_wg - 1) / tiles_per_wg;
    const size_t lds1 = sort_scatter_lds(g), lds2 = sort_local_lds(g);
    const void* k1 = (const void*)msm_digits_scatter_kernel<SORT_THREADS1>;
    const void* k2 = (const void*)msm_sort_local_kernel;
    if (int lrc = raise_lds_limit(ctx, k1, lds1)) return lrc;
    if (int lrc = raise_lds_limit(ctx, k2, lds2)) return lrc;
    {
      ProfScope prof(ctx, st, "msm_digits");
      hipLaunchKernelGGL(msm_digits_hist_kernel<SORT_THREADS0>, dim3(batch * wgs_per_msm), dim3(SORT_THREADS0), (size_t)g.bins * 4, st,
                         (const u32x4*)d_scalars, n, sc_stride, scalar_form, g, tiles_per_wg, wgs_per_msm, (u32)l1_threads,
                         ctx->opt_msm_chunk ? L1 : chunk_lo, ctl, (u32x4*)(ws + o_canon), (unsigned short*)(ws + o_rows));
      MSM_STAGE(ctx, st, "digits histogram");
      hipLaunchKernelGGL(msm_digits_scatter_kernel<SORT_THREADS1>, dim3(std::min<u32>(batch * g.tiles, (u32)ctx->num_cus)),
                         dim3(SORT_THREADS1), lds1, st, (const u32x4*)(ws + o_canon), (const unsigned short*)(ws + o_rows), n, g,
                         (u32)offset, ctl, (u64*)(ws + o_pairs), batch * g.tiles);
      MSM_STAGE(ctx, st, "digits scatter");
    }
    PM_HIP(ctx, hipGetLastError());
    {
      ProfScope prof(ctx, st, "msm_sort_pairs");
      hipLaunchKernelGGL(msm_sort_local_kernel, dim3(g.np), dim3(SORT_THREADS), lds2, st, g, (const u32*)ctl,
                         (const u64*)(ws + o_pairs), keys1, vals1);
      MSM_STAGE(ctx, st, "local sort");
    }
    PM_HIP(ctx, hipGetLastError());
  }
  // 3 accumulate
  // empty buckets = the identity = ZZ all zero: only that quarter of every 256-byte record is cleared
  hipLaunchKernelGGL(msm_clear_buckets_kernel, dim3((unsigned)((total_buckets * 4 + 255) / 256)), dim3(256), 0, st, buckets,
                     total_buckets);
  PM_HIP(ctx, hipGetLastError());
  AccArgs a;
  memset(&a, 0, sizeof a);
  a.bases = (const u32x4*)(bases->table_c ? bases->d_table : bases->d_xy);
  a.trash = g.trash;
  a.ctl = ctl;
  a.grid_threads = l1_threads;
  a.buckets = buckets;
  a.head_pts = (u32x4*)(ws + o_heads);
  for (size_t lvl = 0; lvl < lv.size(); ++lvl) {
    const bool last = (lvl + 1 == lv.size());
    a.len = lv[lvl].len;
    a.chunk = lv[lvl].chunk;
    a.offset = lv[lvl].offset;
    a.final_level = last ? 1u : 0u;
    if (lvl == 0) {
      a.keys = keys1;
      a.vals = vals1;
    } else {
      a.keys = (const u32*)(ws + o_pkeys[lvl]);
      a.pts_in = (const u32x4*)(ws + o_ppts[lvl]);
    }
    if (!last) {
      a.part_keys = (u32*)(ws + o_pkeys[lvl + 1]);
      a.part_pts = (u32x4*)(ws + o_ppts[lvl + 1]);
    }
    const size_t nthr = lvl == 0 ? l1_threads : (a.len + a.offset + a.chunk - 1) / a.chunk;
    const unsigned blocks = (unsigned)((nthr + 127) / 128);
    if (lvl > 0 && last && nthr > 64) return set_err(ctx, PM_ERR_BAD_ARG, "internal: final MSM level wider than a wave");
    {
      ProfScope prof(ctx, st, lvl == 0 ? "msm_accumulate_l1" : "msm_accumulate_ln");
      if (lvl == 0) {
        // Two waves that share a SIMD run one after the other (oldest first, section 4 of DESIGN.md), and two-wave workgroups
        // are not spread evenly: a grid of at most one (two) waves per SIMD is launched as four-wave workgroups -- a wave per
        // SIMD of a CU -- with an LDS request that keeps a second (third) workgroup off the CU.  The kernel uses no LDS.
        const size_t waves = (nthr + 63) / 64;
        size_t place_lds = 0;
        // (just over a half / a third of the CU's 160 KB: what is left -- 79 KB / 52 KB -- still takes the workgroups of the
        // prover's side stream, the coset transforms that run beside the commitments of rounds 1 and 2; with 96 / 72 KB those
        // kept accumulate workgroups waiting for a CU: 2^16-gate proofs took 4.13 ms or 4.47 ms, at random)
        if (waves <= lone_waves) place_lds = 81 * 1024;
        else if (waves <= 2 * lone_waves) place_lds = 54 * 1024;
        if (place_lds) {
          if (int lrc = raise_lds_limit(ctx, (const void*)msm_accumulate_l1_kernel, place_lds)) return lrc;
          hipLaunchKernelGGL(msm_accumulate_l1_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), place_lds, st, a);
        } else {
          hipLaunchKernelGGL(msm_accumulate_l1_kernel, dim3(blocks), dim3(128), 0, st, a);
        }
      } else {
        hipLaunchKernelGGL(msm_accumulate_ln_kernel, dim3(blocks), dim3(128), 0, st, a);
      }
      MSM_STAGE(ctx, st, lvl == 0 ? "accumulate level 1" : "accumulate level n");
    }
    PM_HIP(ctx, hipGetLastError());
  }
  if (front_done) PM_HIP(ctx, hipEventRecord(front_done, st));
  // 4 bucket reduce: level 1 over the buckets, then the small levels (see msm_bucket_reduce_kernel)
  u32x4* win = (u32x4*)(ws + o_win);
  const u32 n_waves = nsets_all * n1;
#ifdef PM_DEV_STAMPS
  static unsigned long long* d_stamps = nullptr;
  if (getenv("PM_MSM_STAMPS")) {
    if (d_stamps) (void)hipFree(d_stamps);
    PM_HIP(ctx, hipMalloc(&d_stamps, (size_t)n_waves * 128));
    PM_HIP(ctx, hipMemsetAsync(d_stamps, 0, (size_t)n_waves * 128, st));
    PM_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_bucket_stamps), &d_stamps, sizeof d_stamps, 0, hipMemcpyHostToDevice, st));
  }
#endif
  {
    ProfScope prof(ctx, st, "msm_bucket_chunk");
    // four-wave workgroups, ONE per CU (its LDS request keeps a second one out): a wave per SIMD
    const size_t red_lds = (size_t)RED_WAVES * RED_SLOTS * 256;
    if (int lrc = raise_lds_limit(ctx, (const void*)msm_bucket_reduce_kernel, red_lds)) return lrc;
    hipLaunchKernelGGL(msm_bucket_reduce_kernel, dim3((n_waves + RED_WAVES - 1) / RED_WAVES), dim3(64 * RED_WAVES), red_lds, st,
                       (const u32x4*)buckets, g.nbuckets, LB, n1, n_waves, n_dev == 0 && !use_finish ? 1u : 0u,
                       n_dev == 0 ? win : (u32x4*)(ws + o_l1), (size_t)n_waves * 16);
  }
  PM_HIP(ctx, hipGetLastError());
#ifdef PM_DEV_STAMPS
  if (getenv("PM_MSM_STAMPS") && d_stamps) {
    PM_HIP(ctx, hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)n_waves * 16);
    PM_HIP(ctx, hipMemcpy(h.data(), d_stamps, (size_t)n_waves * 128, hipMemcpyDeviceToHost));
    const int NPH = 3;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (size_t i = 0; i < n_waves; ++i) {
      r0 = std::min(r0, h[16 * i + 1]);
      r1 = std::max(r1, h[16 * i + 2 * NPH + 1]);
    }
    double sum_c[NPH] = {0}, sum_w[NPH] = {0}, start_sum = 0, end_sum = 0, start_max = 0, end_min = 1e30;
    for (size_t i = 0; i < n_waves; ++i) {
      for (int k = 0; k < NPH; ++k) {
        sum_c[k] += (double)(h[16 * i + 2 * (k + 1)] - h[16 * i + 2 * k]);
        sum_w[k] += (double)(h[16 * i + 2 * (k + 1) + 1] - h[16 * i + 2 * k + 1]);
      }
      const double s0 = (double)(h[16 * i + 1] - r0) / 100.0, e0 = (double)(h[16 * i + 2 * NPH + 1] - r0) / 100.0;
      start_sum += s0; end_sum += e0;
      start_max = std::max(start_max, s0); end_min = std::min(end_min, e0);
    }
    const double nw = (double)n_waves;   // wall clock: 100 MHz ticks
    fprintf(stderr, "[stamps] waves %u lb %u  kernel span %.1f us | start mean %.1f max %.1f us | end mean %.1f min %.1f us\n",
            n_waves, LB, (double)(r1 - r0) / 100.0, start_sum / nw, start_max, end_sum / nw, end_min);
    static const char* ph[NPH] = {"per-pair", "butterfly", "store"};
    for (int k = 0; k < NPH; ++k)
      fprintf(stderr, "[stamps]   %-9s %.1f us  %.0f shader cycles  -> %.3f GHz\n", ph[k], sum_w[k] / nw / 100.0, sum_c[k] / nw,
              sum_c[k] / (sum_w[k] * 10.0));
    unsigned hist[11] = {0};
    for (size_t i = 0; i < n_waves; ++i) hist[(size_t)(10.0 * (double)(h[16 * i + 2 * NPH + 1] - r0) / (double)(r1 - r0))]++;
    fprintf(stderr, "[stamps]   end-time histogram (tenths of the span):");
    for (int k = 0; k < 11; ++k) fprintf(stderr, " %u", hist[k]);
    fprintf(stderr, "\n");
  }
#endif
  if (n_dev) {
    ProfScope prof(ctx, st, "msm_window_sum");
    const u32x4* in = (const u32x4*)(ws + o_l1);
    u32 items = n1, n_seq = PLANES + 2;
    for (u32 k = 0; k < n_dev; ++k) {
      const u32 groups = lvl_groups[k];
      const bool last = k + 1 == n_dev;   // writes the host's sequences, fit for conversion
      RedLevelArgs la;
      la.in = in;
      la.out = last ? win : (u32x4*)(ws + o_lvl[k]);
      la.in_stride = (size_t)nsets_all * items * 16;
      la.out_stride = (size_t)nsets_all * groups * 16;
      la.n_seq_in = n_seq;
      la.n_items = items;
      la.n_groups = groups;
      la.finalize = last && !use_finish ? 1u : 0u;
      hipLaunchKernelGGL(msm_reduce_level_kernel, dim3(nsets_all * groups, n_seq), dim3(64), 0, st, la);
      in = la.out;
      items = groups;
      n_seq += PLANES;
    }
  }
  if (use_finish) {
    ProfScope prof(ctx, st, "msm_window_sum");
    hipLaunchKernelGGL(msm_reduce_finish_kernel, dim3(nsets_all), dim3(64), 0, st, (const u32x4*)win,
                       (size_t)nsets_all * host_items * 16, nsets_all, host_items, nplanes, log_lb, (u32x4*)(ws + o_fin));
  }
  PM_HIP(ctx, hipGetLastError());
  PM_HIP(ctx, hipMemcpyAsync(pinned, ws + (use_finish ? o_fin : o_win), pinned_bytes, hipMemcpyDeviceToHost, st));
  piece->batch = batch;
  piece->nsets = g.nsets;
  piece->c = g.c;
  piece->n_dev = n_dev;
  piece->host_items = host_items;
  piece->finished = use_finish;
  piece->log_lb = log_lb;
  piece->nsets_all = nsets_all;
  piece->hw = (const u32*)pinned;
  return PM_OK;
}
// 5 host fold of a piece whose copy has arrived.  Per set the device left 7 + 5 n_dev sequences of host_items entries:
// Tt, the planes Q_0 .. Q_(5 L - 1) (L = n_dev + 1; plane b = the sum of S over the chunks whose index has bit b set) and the
// last W.  The set's total is  Tt + lb (sum_b 2^b Q_b + 32^L sum_l l W_l),  by Horner's rule from the top.
static void msm_fold(const MsmPiece& pc, XYZZ* totals) {
  const u32* hw = pc.hw;
  auto entry = [&](u32 seq, size_t set, u32 item) {
    return xyzz_to_host(hw + 64 * (((size_t)seq * pc.nsets_all + set) * pc.host_items + item));
  };
  auto seq_sum = [&](u32 seq, size_t set) {
    XYZZ f = entry(seq, set, 0);
    for (u32 l = 1; l < pc.host_items; ++l) f = host::xyzz_add(f, entry(seq, set, l));
    return f;
  };
  auto set_total = [&](size_t set) {
    if (pc.finished) return xyzz_to_host(hw + 64 * set);
    const u32 nplanes = PLANES * (pc.n_dev + 1), w_seq = nplanes + 1;
    XYZZ acc = host::xyzz_identity();
    if (pc.host_items > 1) {   // sum_l l W_l as the sum of the suffix sums from l = 1
      XYZZ suffix = host::xyzz_identity();
      for (u32 l = pc.host_items; l-- > 1;) {
        suffix = host::xyzz_add(suffix, entry(w_seq, set, l));
        acc = host::xyzz_add(acc, suffix);
      }
    }
    for (u32 b = nplanes; b-- > 0;) {
      acc = host::xyzz_double(acc);
      acc = host::xyzz_add(acc, seq_sum(1 + b, set));
    }
    for (u32 d = 0; d < pc.log_lb; ++d) acc = host::xyzz_double(acc);
    return host::xyzz_add(acc, seq_sum(0, set));
  };
  for (u32 j = 0; j < pc.batch; ++j) {
    XYZZ total = host::xyzz_identity();
    for (u32 w = pc.nsets; w-- > 0;) {  // one set (table mode): no window doublings at all
      if (w + 1 < pc.nsets)
        for (u32 k = 0; k < pc.c; ++k) total = host::xyzz_double(total);
      total = host::xyzz_add(total, set_total((size_t)j * pc.nsets + w));
    }
    totals[j] = total;
  }
}

// A call with `batch` scalar vectors normally runs as ONE piece (more only when the 31-bit pair indices force it).
// Option "msm_pipeline" = 1 runs it as up to four pieces in a two-stream software pipeline instead -- piece i + 1's
// bucket fill and accumulate start when piece i's accumulate is done, so that piece i's tail (bucket reduction, window
// sums, result copy: ~0.8 ms of kernels that are latency-bound at one or two waves per SIMD) would run under the next
// piece's accumulate.  Built, tested (tests/test_gpu_msm.py runs both settings) and MEASURED TO
// LOSE: 2^20-gate proof 36.5 ms as one piece, 44.1 ms as pieces (profiles/r03_msm_pipeline_ab.txt).  The accumulate
// grid is sized to fill the chip exactly once (two waves per SIMD at 236 VGPRs); a tail kernel that holds even a few
// of those slots when the next accumulate is dispatched pushes that many of its workgroups into a second round, which
// costs a whole chunk time (~2 ms) however few they are.  Inside one proof nothing else can fill the tails either:
// round k + 1's scalars depend on round k's commitments through the transcript.  Two PROOFS in flight do recover the
// idle slots (bench.py, two_contexts_ms_per_proof), because there whole accumulates alternate.
int msm_run(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n, const void* d_scalars, size_t sc_stride,
            u32 batch, u32 scalar_form, uint64_t* out_xyz /* batch x 18 */, hipStream_t st) {
  if (batch == 0) return PM_OK;
  if (n == 0) {
    for (u32 j = 0; j < batch; ++j) write_projective(out_xyz + 18 * j, host::xyzz_identity());
    return PM_OK;
  }
  if (n > 0x7fffffffu) return set_err(ctx, PM_ERR_BAD_ARG, "n >= 2^31");
  if (bases->table_c && (size_t)bases->n * ((256 + bases->table_c - 1) / bases->table_c) > 0x7fffffffu)
    return set_err(ctx, PM_ERR_BAD_ARG, "window table too large for 31-bit point indices");
  OrderScope order_scope(ctx, ctx->ord_msm, st);   // msm_ws and the pinned result buffer are shared by all streams
  if (order_scope.rc) return order_scope.rc;
  // pieces: pair indices are 31-bit, so a piece holds at most 2^31 - 1 (digit, point) pairs (the 15 key polynomials of
  // a 2^24-gate circuit do not fit one)
  const size_t pairs_per_msm = n * make_geom(n, ctx->opt_msm_window_bits, bases->table_c, bases->n, 1).nwin;
  const size_t max_pairs = ctx->opt_msm_max_pairs ? (size_t)ctx->opt_msm_max_pairs : (size_t)0x7fffffffu;
  if (pairs_per_msm > max_pairs) return set_err(ctx, PM_ERR_LENGTH, "n * windows exceeds 2^31 pairs");
  u32 npieces = ctx->opt_msm_pipeline ? std::min<u32>(batch, 4u) : 1u;
  while ((size_t)((batch + npieces - 1) / npieces) * pairs_per_msm > max_pairs) ++npieces;
  const u32 per_piece = (batch + npieces - 1) / npieces;
  npieces = (batch + per_piece - 1) / per_piece;
  // sizes: all pieces but the last have per_piece vectors
  size_t region = 0, pin_each = 0;
  int rc = msm_piece(ctx, bases, offset, n, d_scalars, sc_stride, per_piece, scalar_form, st, nullptr, nullptr, &region, nullptr,
                     &pin_each, nullptr);
  if (rc) return rc;
  region = align_up(region, 4096);
  rc = ensure_buffer(ctx, ctx->msm_ws, npieces > 1 ? 2 * region : region);
  if (rc) return rc;
  if (ctx->msm_host_pinned_bytes < pin_each * npieces) {
    if (ctx->msm_host_pinned) (void)hipHostFree(ctx->msm_host_pinned);
    ctx->msm_host_pinned = nullptr;
    ctx->msm_host_pinned_bytes = std::max<size_t>(64 * 256, pin_each * npieces);
    PM_HIP(ctx, hipHostMalloc(&ctx->msm_host_pinned, ctx->msm_host_pinned_bytes, hipHostMallocDefault));
  }
  {
    // control block of the bucket fill: zero when idle (the histogram kernel restores that), zeroed when (re)allocated
    const MsmGeom g1 = make_geom(n, ctx->opt_msm_window_bits, bases->table_c, bases->n, per_piece);
    if (ctx->msm_ctl_cap < g1.np) {
      const u32 cap = std::max<u32>(g1.np, 1u << 16);
      rc = ensure_buffer(ctx, ctx->msm_ctl, sort_ctl_words(cap) * 4);
      if (rc) return rc;
      PM_HIP(ctx, hipMemsetAsync(ctx->msm_ctl.ptr, 0, ctx->msm_ctl.bytes, st));
      ctx->msm_ctl_cap = cap;
    }
  }
  hipStream_t streams[2] = {st, st};
  if (npieces > 1) {
    if (!ctx->msm_side) PM_HIP(ctx, hipStreamCreateWithFlags(&ctx->msm_side, hipStreamNonBlocking));
    streams[1] = ctx->msm_side;
    while (ctx->msm_events.size() < npieces) {
      hipEvent_t e = nullptr;
      PM_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->msm_events.push_back(e);
    }
  }
  std::vector<MsmPiece> pieces(npieces);
  for (u32 i = 0; i < npieces && rc == PM_OK; ++i) {
    const u32 first = i * per_piece, cnt = std::min<u32>(per_piece, batch - first);
    hipStream_t si = streams[i & 1];
    // piece i starts when piece i - 1 has left the accumulate (and with it everything the caller queued on `st` before
    // this call: piece 0 runs on `st`); the piece that used this region and this stream before, i - 2, is in order
    if (i > 0) {
      hipError_t we = hipStreamWaitEvent(si, ctx->msm_events[i - 1], 0);
      if (we != hipSuccess) {
        rc = set_err(ctx, PM_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(we));
        break;
      }
    }
    size_t dummy_ws = 0, dummy_pin = 0;
    rc = msm_piece(ctx, bases, offset, n, (const char*)d_scalars + (size_t)first * sc_stride * 32, sc_stride, cnt, scalar_form, si,
                   npieces > 1 ? ctx->msm_events[i] : nullptr, (char*)ctx->msm_ws.ptr + (size_t)(i & 1) * region, &dummy_ws,
                   (char*)ctx->msm_host_pinned + (size_t)i * pin_each, &dummy_pin, &pieces[i]);
  }
  // whatever was enqueued is waited for, also on an error path: the side stream must be idle when the call returns
  host_mark(ctx, "msm: enqueued");
  hipError_t e0 = hipStreamSynchronize(st), e1 = npieces > 1 ? hipStreamSynchronize(ctx->msm_side) : hipSuccess;
  host_mark(ctx, "msm: device done");
  if (rc) return rc;
  PM_HIP(ctx, e0);
  PM_HIP(ctx, e1);
  std::vector<XYZZ> totals(batch);
  for (u32 i = 0; i < npieces; ++i) msm_fold(pieces[i], totals.data() + (size_t)i * per_piece);
  host_mark(ctx, "msm: folded");
  write_projective_batch(out_xyz, totals.data(), batch);
  host_mark(ctx, "msm: affine");
  return PM_OK;
}

// ------------------------------------------------------------------ fixed-base multiplication
// out[i] = scalars[i] * B from the table T[w][d-1] = d 2^(8 w) B (32 rows of 255 affine points, the
// layout pm_g1_bases_precompute(window_bits = 8) builds from the row [B, 2B, .. 255B]): at most 32
// mixed additions per scalar, no doublings.  SRS generation: powers_of_g[i] = tau^i G.
__global__ void __launch_bounds__(128) fixed_base_kernel(const u32x4* scalars, size_t n, u32 scalar_form,
                                                         const u32x4* table, u32x4* scratch) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = fe_load<FrP>(scalars + 2 * i);
  Fr f;
  if (scalar_form == PM_SCALAR_MONTGOMERY) {
    f = fe_zero<FrP>();
    f.l[0] = 32u;
  } else {
    f = fe_one<FrP>();
  }
  u32 w[8];
  fe_canon_pack<FrP>(w, fe_mul<FrP>(s, f));
  Xyzz acc = xyzz_identity();
  for (u32 k = 0; k < 32; ++k) {
    const u32 d = (w[k >> 2] >> (8 * (k & 3))) & 255u;
    if (d) {
      const u32x4* p = table + 6 * ((size_t)k * 255 + d - 1);
      acc = xyzz_madd(acc, fe_load<FpP>(p), fe_load<FpP>(p + 3));
    }
  }
  st_xyzz(scratch, i, acc);
}

}  // namespace pm

// ------------------------------------------------------------------ C ABI
using namespace pm;

extern "C" int pm_g1_bases_upload(pm_ctx* ctx, const uint64_t* xy, size_t n, pm_bases** out) {
  if (!ctx || !out || (!xy && n)) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  *out = nullptr;
  PM_HIP(ctx, hipSetDevice(ctx->device));
  pm_bases* b = new pm_bases();
  b->n = n;
  b->device = ctx->device;
  if (n) {
    void* staging = nullptr;
    hipError_t e = hipMalloc(&b->d_xy, n * 96);
    if (e == hipSuccess) e = hipMalloc(&staging, n * 96);
    if (e != hipSuccess) {
      if (b->d_xy) (void)hipFree(b->d_xy);
      delete b;
      return set_err(ctx, PM_ERR_OOM, "hipMalloc for bases failed");
    }
    e = hipMemcpyAsync(staging, xy, n * 96, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(bases_convert_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0,
                         ctx->stream, (const u32x4*)staging, (u32x4*)b->d_xy, 2 * n);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(staging);
    if (e != hipSuccess) {
      (void)hipFree(b->d_xy);
      delete b;
      return set_err(ctx, PM_ERR_HIP, std::string("bases upload: ") + hipGetErrorString(e));
    }
  }
  *out = b;
  return PM_OK;
}

extern "C" void pm_g1_bases_free(pm_ctx* ctx, pm_bases* bases) {
  if (!bases) return;
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (bases->d_table) (void)hipFree(bases->d_table);
    else if (bases->d_xy) (void)hipFree(bases->d_xy);
  }
  delete bases;
}

// Build the table of window multiples 2^(c w) * P_i (w < ceil(256 / c)) for a resident SRS.
// Costs ~8 MSMs of time once and (ceil(256/c) - 1) x 96 n bytes of HBM; afterwards every MSM on
// these bases uses ONE bucket set: no per-window bucket reduction, no doublings in the fold, and a
// wider window (fewer additions per scalar).  window_bits 0 = the library's choice (log2 n, <= 20).
extern "C" int pm_g1_bases_precompute(pm_ctx* ctx, pm_bases* bases, uint32_t window_bits) {
  if (!ctx || !bases) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (bases->table_c) return set_err(ctx, PM_ERR_BAD_ARG, "bases already carry a window table");
  const size_t n = bases->n;
  u32 lg = 0;
  while (((size_t)2 << lg) <= std::max<size_t>(n, 1)) ++lg;
  // default (measured, profiles/r02_msm_sweep.txt): the accumulate kernel does n * ceil(256 / c) additions and
  // the bucket reduction ~60 sequential group operations whatever the bucket count, so the window grows
  // with n -- 2^17 points (an 8-way shard of 2^20): c = 16, 1.4 ms; 2^20: c = 20, 4.1 ms; 2^21: c = 20,
  // 6.8 ms.  Only widths whose top window still spans several bits are used (13 -> 8 bits, 16 -> 15,
  // 20 -> 15): a top window of 1-4 bits (c = 17, 18, 21) puts n / 8 points into each of a few buckets.
  // 2^24 (profiles/r03_msm_sweep.txt): c = 22 -- 12 table rows instead of 13, 2^21 buckets -- 36.6 ms against 38.9 ms
  // for c = 20 and 37.7 ms for c = 24 (whose 2^23 buckets cost 4.3 ms to reduce).
  // measured per size with the r04 bucket reduction (profiles/r05_small_msm.txt): 13 up to 2^13 points, 16 from 2^14 (r02 - r04:
  // from 2^16; a 2^14-gate proof 3.05 -> 2.81 ms), 20 from 2^19, 22 from 2^23
  const u32 c = window_bits ? window_bits : (lg <= 13 ? 13u : (lg <= 18 ? 16u : (lg <= 22 ? 20u : 22u)));
  if (c < 8 || c > 24) return set_err(ctx, PM_ERR_BAD_ARG, "window_bits must be 8..24");
  if (n == 0) return PM_OK;
  const u32 nwin = (256 + c - 1) / c;
  if (n * nwin > 0x7fffffffu) return set_err(ctx, PM_ERR_LENGTH, "n * windows exceeds 2^31 table entries");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  void* table = nullptr;
  PM_HIP(ctx, hipMalloc(&table, n * nwin * 96));
  int rc = ensure_buffer(ctx, ctx->msm_ws, n * (256 + 64));
  if (rc) {
    (void)hipFree(table);
    return rc;
  }
  u32x4* scratch = (u32x4*)ctx->msm_ws.ptr;
  u32x4* prefix = scratch + 16 * n;
  hipError_t e = hipMemcpyAsync(table, bases->d_xy, n * 96, hipMemcpyDeviceToDevice, st);
  const size_t want = std::max<size_t>((n + 63) / 64, std::min<size_t>(n, (size_t)ctx->num_cus * 512));
  const unsigned blocks = (unsigned)((want + 127) / 128);
  const u32 per = (u32)((n + (size_t)blocks * 128 - 1) / ((size_t)blocks * 128));
  for (u32 w = 1; w < nwin && e == hipSuccess; ++w) {
    const u32x4* src = (const u32x4*)table + 6 * n * (w - 1);
    u32x4* dst = (u32x4*)table + 6 * n * w;
    hipLaunchKernelGGL(precompute_double_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, src, n, c, scratch);
    hipLaunchKernelGGL(precompute_affine_kernel, dim3(blocks), dim3(128), 0, st, (const u32x4*)scratch, n, per, prefix, dst, 0u);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)hipFree(table);
    return set_err(ctx, PM_ERR_HIP, std::string("bases precompute: ") + hipGetErrorString(e));
  }
  (void)hipFree(bases->d_xy);
  bases->d_xy = table;  // row 0 == the bases themselves
  bases->d_table = table;
  bases->table_c = c;
  return PM_OK;
}

extern "C" size_t pm_g1_bases_len(const pm_bases* bases) { return bases ? bases->n : 0; }

// test hook (pure host): what one piece of an MSM call of this shape would take -- the library's own sizing pass, on a
// context that never touches a device.  out[4]: return code of the sizing pass, device workspace bytes, pinned host bytes,
// (digit, point) pairs at most.
extern "C" int pm_test_msm_sizing(size_t n, uint32_t batch, long window_bits, uint32_t table_window_bits, uint32_t num_cus,
                                  uint64_t out[4]) {
  if (!out || !batch) return PM_ERR_BAD_ARG;
  pm_ctx ctx;
  ctx.num_cus = num_cus ? (int)num_cus : 256;
  ctx.opt_msm_window_bits = window_bits;
  pm_bases bases;
  bases.n = n;
  bases.table_c = table_window_bits;
  size_t need_ws = 0, need_pin = 0;
  const int rc = n ? msm_piece(&ctx, &bases, 0, n, nullptr, n, batch, PM_SCALAR_MONTGOMERY, nullptr, nullptr, nullptr, &need_ws,
                               nullptr, &need_pin, nullptr)
                   : PM_OK;
  out[0] = (uint64_t)(int64_t)rc;
  out[1] = need_ws;
  out[2] = need_pin;
  out[3] = n ? (uint64_t)n * batch * make_geom(n, window_bits, table_window_bits, table_window_bits ? n : 0, batch).nwin : 0;
  return PM_OK;
}

// test hook (pure host): the bucket-fill layout the library would use for an MSM of this shape, and its bucket ->
// partition map.  out[16]: c, windows, bucket sets, bucket bits, partition bits, local bits, partitions per set, bins,
// partitions, scalars per scatter tile, tiles, LDS bytes of the scatter kernel, of the local sort, finer low partitions
// (count, extra bits), 0.  part_of_bucket / first_bucket / width_bits (each n_buckets_out entries, may be NULL): the
// partition of every bucket of one set, and for every partition its first bucket and log2 of its bucket count.
extern "C" int pm_test_msm_geometry(size_t n, long window_bits, uint32_t table_window_bits, uint32_t batch, uint32_t out[16],
                                    uint32_t* part_of_bucket, uint32_t* first_bucket, uint32_t* width_bits) {
  if (!out || !n || !batch) return PM_ERR_BAD_ARG;
  const MsmGeom g = make_geom(n, window_bits, table_window_bits, table_window_bits ? n : 0, batch);
  const uint32_t v[16] = {g.c, g.nwin, g.nsets, g.bbits, g.pbits, g.rbits, g.pps, g.bins, g.np, g.ts, g.tiles,
                          (uint32_t)sort_scatter_lds(g), (uint32_t)sort_local_lds(g), g.na, g.sa, 0};
  memcpy(out, v, sizeof v);
  if (part_of_bucket)
    for (u32 b = 0; b < g.nbuckets; ++b) part_of_bucket[b] = part_of(g, b);
  for (u32 p = 0; p < g.pps && first_bucket && width_bits; ++p) part_range(g, p, first_bucket[p], width_bits[p]);
  return PM_OK;
}

extern "C" int pm_g1_msm_dev(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n,
                             const void* d_scalars, uint32_t scalar_form, uint64_t out_xyz[18],
                             void* hip_stream) {
  if (!ctx || !bases || !out_xyz || (!d_scalars && n)) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (scalar_form > PM_SCALAR_CANONICAL) return set_err(ctx, PM_ERR_BAD_ARG, "scalar_form");
  if (offset > bases->n || n > bases->n - offset)
    return set_err(ctx, PM_ERR_LENGTH, "more scalars than uploaded bases");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  return msm_run(ctx, bases, offset, n, d_scalars, n, 1, scalar_form, out_xyz, st);
}

extern "C" int pm_g1_msm_batch_dev(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n, const void* d_scalars,
                                   size_t scalar_stride, uint32_t batch, uint32_t scalar_form, uint64_t* out_xyz,
                                   void* hip_stream) {
  if (!ctx || !bases || !out_xyz || (!d_scalars && n && batch)) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (scalar_form > PM_SCALAR_CANONICAL) return set_err(ctx, PM_ERR_BAD_ARG, "scalar_form");
  if (offset > bases->n || n > bases->n - offset)
    return set_err(ctx, PM_ERR_LENGTH, "more scalars than uploaded bases");
  if (batch > 1 && scalar_stride < n) return set_err(ctx, PM_ERR_BAD_ARG, "scalar_stride shorter than n");
  if (batch > 64) return set_err(ctx, PM_ERR_BAD_ARG, "batch > 64");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  return msm_run(ctx, bases, offset, n, d_scalars, scalar_stride, batch, scalar_form, out_xyz, st);
}

extern "C" int pm_g1_msm(pm_ctx* ctx, const pm_bases* bases, size_t n, const uint64_t* scalars,
                         uint32_t scalar_form, uint64_t out_xyz[18]) {
  if (!ctx || !bases || !out_xyz || (!scalars && n)) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (scalar_form > PM_SCALAR_CANONICAL) return set_err(ctx, PM_ERR_BAD_ARG, "scalar_form");
  if (n > bases->n) return set_err(ctx, PM_ERR_LENGTH, "more scalars than uploaded bases");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  if (n) {
    int rc = ensure_buffer(ctx, ctx->msm_scalars, n * 32);
    if (rc) return rc;
    PM_HIP(ctx, hipMemcpyAsync(ctx->msm_scalars.ptr, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  return msm_run(ctx, bases, 0, n, ctx->msm_scalars.ptr, n, 1, scalar_form, out_xyz, ctx->stream);
}

extern "C" int pm_g1_fold(const uint64_t* xyz_parts, size_t k, uint64_t out_xyz[18]) {
  if (!out_xyz || (!xyz_parts && k)) return PM_ERR_BAD_ARG;
  XYZZ total = host::xyzz_identity();
  for (size_t i = 0; i < k; ++i) total = host::xyzz_add(total, projective_to_xyzz(xyz_parts + 18 * i));
  write_projective(out_xyz, total);
  return PM_OK;
}

extern "C" int pm_g1_to_affine(const uint64_t xyz[18], uint64_t xy[12], int* is_identity) {
  if (!xyz || !xy) return PM_ERR_BAD_ARG;
  XYZZ p = projective_to_xyzz(xyz);
  HFp x, y;
  memset(xy, 0, 96);
  const bool ok = host::xyzz_to_affine(p, x, y);
  if (ok) {
    memcpy(xy, x.l, 48);
    memcpy(xy + 6, y.l, 48);
  }
  if (is_identity) *is_identity = ok ? 0 : 1;
  return PM_OK;
}

// k points at once with ONE field inversion (Montgomery's trick over the Z coordinates; identities are skipped and
// come back as (0, 0)): what a prover round's batch of commitments needs -- a host inversion is ~50 us.
extern "C" int pm_g1_to_affine_batch(const uint64_t* xyz, size_t k, uint64_t* xy, int* is_identity) {
  if ((!xyz || !xy) && k) return PM_ERR_BAD_ARG;
  if (k == 0) return PM_OK;   // (memset / memcpy on a null pointer are undefined even for zero bytes)
  const host::Field<6>& F = host::FP();
  std::vector<HFp> Z(k), pre(k);
  HFp acc = host::one(F);
  for (size_t i = 0; i < k; ++i) {
    memcpy(Z[i].l, xyz + 18 * i + 12, 48);
    pre[i] = acc;                                   // product of the earlier non-zero Z
    if (!host::is_zero(Z[i])) acc = host::mul(acc, Z[i], F);
  }
  // results of pm_g1_msm* and pm_g1_fold are already normalised (Z = 1): then there is nothing to invert
  HFp inv = memcmp(acc.l, F.one, 48) == 0 ? acc : host::inv(acc, F);
  memset(xy, 0, 96 * k);
  for (size_t i = k; i-- > 0;) {
    const bool ident = host::is_zero(Z[i]);
    if (is_identity) is_identity[i] = ident ? 1 : 0;
    if (ident) continue;
    const HFp zi = host::mul(inv, pre[i], F);       // 1 / Z_i
    inv = host::mul(inv, Z[i], F);
    HFp X, Y;
    memcpy(X.l, xyz + 18 * i, 48);
    memcpy(Y.l, xyz + 18 * i + 6, 48);
    const HFp x = host::mul(X, zi, F), y = host::mul(Y, zi, F);
    memcpy(xy + 12 * i, x.l, 48);
    memcpy(xy + 12 * i + 6, y.l, 48);
  }
  return PM_OK;
}

// Device-resident affine points (ABI layout) -> pm_bases, no host round trip (an SRS generated or
// received on the GPU).
extern "C" int pm_g1_bases_from_dev(pm_ctx* ctx, const void* d_xy, size_t n, pm_bases** out) {
  if (!ctx || !out || (!d_xy && n)) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  *out = nullptr;
  PM_HIP(ctx, hipSetDevice(ctx->device));
  pm_bases* b = new pm_bases();
  b->n = n;
  b->device = ctx->device;
  if (n) {
    hipError_t e = hipMalloc(&b->d_xy, n * 96);
    if (e != hipSuccess) {
      delete b;
      return set_err(ctx, PM_ERR_OOM, "hipMalloc for bases failed");
    }
    hipLaunchKernelGGL(bases_convert_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const u32x4*)d_xy, (u32x4*)b->d_xy, 2 * n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(b->d_xy);
      delete b;
      return set_err(ctx, PM_ERR_HIP, std::string("bases from device: ") + hipGetErrorString(e));
    }
  }
  *out = b;
  return PM_OK;
}

extern "C" int pm_g1_fixed_base_mul_dev(pm_ctx* ctx, const uint64_t base_xy[12], const void* d_scalars, size_t n,
                                        uint32_t scalar_form, void* d_out_xy, void* hip_stream) {
  if (!ctx || !base_xy) return PM_ERR_BAD_ARG;
  if (scalar_form > PM_SCALAR_CANONICAL) return set_err(ctx, PM_ERR_BAD_ARG, "scalar_form");
  if (n == 0) return PM_OK;
  if (!d_scalars || !d_out_xy) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  if (n > 0x7fffffffu) return set_err(ctx, PM_ERR_LENGTH, "n > 2^31");
  // row 0 of the table on the host: d B for d = 1..255 (ABI Montgomery affine)
  using host::HFp;
  HFp bx, by;
  memcpy(bx.l, base_xy, 48);
  memcpy(by.l, base_xy + 6, 48);
  std::vector<uint64_t> row(255 * 12, 0);
  if (!(host::is_zero(bx) && host::is_zero(by))) {
    XYZZ b1;
    b1.x = bx;
    b1.y = by;
    b1.zz = host::one(host::FP());
    b1.zzz = host::one(host::FP());
    XYZZ cur = b1;
    for (int d = 1; d <= 255; ++d) {
      HFp x, y;
      if (host::xyzz_to_affine(cur, x, y)) {
        memcpy(&row[12 * (d - 1)], x.l, 48);
        memcpy(&row[12 * (d - 1) + 6], y.l, 48);
      }
      cur = host::xyzz_add(cur, b1);
    }
  }
  pm_bases* tb = nullptr;
  int rc = pm_g1_bases_upload(ctx, row.data(), 255, &tb);
  if (rc) return rc;
  rc = pm_g1_bases_precompute(ctx, tb, 8);
  if (rc) {
    pm_g1_bases_free(ctx, tb);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipError_t e = hipSetDevice(ctx->device);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    if (e == hipSuccess) rc = ensure_buffer(ctx, ctx->msm_ws, n * (256 + 64));
    if (e == hipSuccess && !rc) {
      u32x4* scratch = (u32x4*)ctx->msm_ws.ptr;
      u32x4* prefix = scratch + 16 * n;
      const size_t want = std::max<size_t>((n + 63) / 64, std::min<size_t>(n, (size_t)ctx->num_cus * 512));
      const unsigned blocks = (unsigned)((want + 127) / 128);
      const u32 per = (u32)((n + (size_t)blocks * 128 - 1) / ((size_t)blocks * 128));
      ProfScope prof(ctx, st, "g1_fixed_base_mul");
      hipLaunchKernelGGL(fixed_base_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, (const u32x4*)d_scalars, n,
                         scalar_form, (const u32x4*)tb->d_table, scratch);
      hipLaunchKernelGGL(precompute_affine_kernel, dim3(blocks), dim3(128), 0, st, (const u32x4*)scratch, n, per, prefix,
                         (u32x4*)d_out_xy, 1u);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(st);   // the table is freed below
    }
    if (e != hipSuccess) rc = set_err(ctx, PM_ERR_HIP, std::string("fixed-base mul: ") + hipGetErrorString(e));
  }
  pm_g1_bases_free(ctx, tb);
  return rc;
}
