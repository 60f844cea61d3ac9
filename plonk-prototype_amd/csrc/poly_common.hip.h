// Shared by poly.hip and plonk_rounds.hip: the canonical (ABI) <-> device-form boundary of the
// polynomial helpers.  ABI form: x * 2^256 in 4 x u64; device form: x * 2^261 in 9 x 29-bit limbs.
// fe_mul of forms 2^a and 2^b gives 2^(a + b - 261): ABI x device -> ABI, device x device -> device.
#pragma once
#include <cstring>

#include "host_field.h"
#include "ntt_kernels.hip.h"

namespace pm {

using host::HFr;

// value * 2^5 for a normalised value < 2r: turns the device product a*b/2^261 into the ABI
// product a*b/2^256.  Output limbs normalised, value < 64 r.
PM_DEV Fr fr_shl5(const Fr& t) {
  constexpr u32 MASK = Consts<FrP>::MASK;
  Fr r;
  r.l[0] = (t.l[0] << 5) & MASK;
#pragma unroll
  for (int i = 1; i < 8; ++i) r.l[i] = ((t.l[i] << 5) & MASK) | (t.l[i - 1] >> 24);
  r.l[8] = (t.l[8] << 5) | (t.l[7] >> 24);
  return r;
}
// product of two ABI-form values, ABI form, value < r + r/2^16
PM_DEV Fr fr_abi_mul(const Fr& a, const Fr& b) { return fe_reduce_weak<FrP>(fr_shl5(fe_mul<FrP>(a, b))); }

PM_DEV Fr ld_canon(const u32x4* p, size_t i) { return fe_load<FrP>(p + 2 * i); }
PM_DEV void st_canon(u32x4* p, size_t i, const Fr& v) { fe_store<FrP>(p + 2 * i, v); }

// ------------------------------------------------------------------ host helpers
static inline void to_limbs29(u32* dst, HFr v) {  // ABI Montgomery -> device Montgomery limbs
  for (int i = 0; i < 5; ++i) v = host::add(v, v, host::FR());
  for (int i = 0; i < 9; ++i) {
    const int lo = 29 * i, j = lo / 64, sh = lo % 64;
    u64 x = v.l[j] >> sh;
    if (sh + 29 > 64 && j + 1 < 4) x |= v.l[j + 1] << (64 - sh);
    dst[i] = (u32)(x & ((1u << 29) - 1));
  }
}
static inline HFr hfr_pow_u64(HFr b, u64 e) {
  host::u64 ee[1] = {(host::u64)e};
  return host::pow(b, ee, 1, host::FR());
}
// x * 2^(256 + 5 k) as 29-bit limbs: k = 0 keeps the ABI scaling, 1 = device form, 2 = device form
// of 32 x (so that ABI x this = device form)
static inline void to_limbs29_shift(u32* dst, HFr v, int k) {
  for (int i = 0; i < 5 * (k - 1); ++i) v = host::add(v, v, host::FR());
  if (k >= 1) {
    to_limbs29(dst, v);
    return;
  }
  for (int i = 0; i < 9; ++i) {
    const int lo = 29 * i, j = lo / 64, sh = lo % 64;
    u64 x = v.l[j] >> sh;
    if (sh + 29 > 64 && j + 1 < 4) x |= v.l[j + 1] << (64 - sh);
    dst[i] = (u32)(x & ((1u << 29) - 1));
  }
}

}  // namespace pm
