/* Plain C against include/plonk_mi355x.h -- no Python, no torch: what a cgo / Rust -sys binding sees.
 *   gcc -O2 examples/abi_demo.c -Iinclude -Lplonk-prototype_amd/lib -lplonk_mi355x \
 *       -Wl,-rpath,$PWD/plonk-prototype_amd/lib -o abi_demo && ./abi_demo [log_n]
 * 1. iNTT(NTT(a)) == a and coset_ifft(coset_fft(a)) == a on a 2^log_n vector (host-pointer calls);
 * 2. an MSM over n copies of the generator equals (sum of the scalars) * G computed by a second MSM;
 * 3. the same NTT on device-resident data through pm_dev_* and pm_fr_ntt_dev;
 * 4. the ark-ec call shape: bases as `G1Affine { x, y, infinity: bool }` records marshalled
 *    to the packed 96-byte form (infinity -> all zero), scalars as canonical `BigInteger256` (PM_SCALAR_CANONICAL) --
 *    the same point as the dusk call with the Montgomery forms of the same scalars, an infinity base adding nothing.
 * Exit code 0 and "abi_demo OK" on success. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "plonk_mi355x.h"

static const uint64_t R_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                  0x73eda753299d7d48ULL};
/* dusk_bls12_381::G1Affine::generator(), Montgomery limbs */
static const uint64_t G1_GEN[12] = {0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL,
                                    0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL,
                                    0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL,
                                    0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL};

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t next_u64(void) { /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* r = (a + b) mod R_MOD for a, b < R_MOD (Montgomery forms add like the values they stand for) */
static void fr_add(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  unsigned __int128 c = 0;
  uint64_t t[4], d[4];
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a[i] + b[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  unsigned __int128 bw = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 x = (unsigned __int128)t[i] - R_MOD[i] - (uint64_t)bw;
    d[i] = (uint64_t)x;
    bw = (x >> 64) & 1;
  }
  memcpy(r, (c || !bw) ? d : t, 32);
}
/* ark_ff: Fr::into_repr() -- the canonical integer of a Montgomery form (one REDC: x R -> x) */
static void fr_from_mont(uint64_t out[4], const uint64_t a[4]) {
  const uint64_t INV = 0xfffffffeffffffffULL; /* -r^-1 mod 2^64 */
  uint64_t t[5] = {a[0], a[1], a[2], a[3], 0};
  for (int i = 0; i < 4; ++i) {
    const uint64_t m = t[0] * INV;
    unsigned __int128 c = (unsigned __int128)m * R_MOD[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (unsigned __int128)m * R_MOD[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  memcpy(out, t, 32); /* < r for a < r */
}
/* ark_ec::short_weierstrass_jacobian::GroupAffine<g1::Parameters>: Fq = BigInteger384 Montgomery limbs, then the flag */
struct ark_g1_affine {
  uint64_t x[6], y[6];
  uint8_t infinity;
};
#define CHECK(call)                                                                        \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ != PM_OK) {                                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? pm_last_error(ctx) : "no context"); \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 16;
  const size_t n = (size_t)1 << log_n;
  pm_ctx* ctx = NULL;
  printf("%s\n", pm_version());
  CHECK(pm_init(0, &ctx));

  uint64_t* a = malloc(n * 32), *b = malloc(n * 32), *c = malloc(n * 32);
  for (size_t i = 0; i < n; ++i) {
    for (int l = 0; l < 4; ++l) a[4 * i + l] = next_u64();
    a[4 * i + 3] &= ((uint64_t)1 << 62) - 1; /* < 2^254 < r: a valid BlsScalar */
  }
  CHECK(pm_fr_ntt(ctx, a, n, b, log_n, 0));
  CHECK(pm_fr_ntt(ctx, b, n, c, log_n, PM_NTT_INVERSE));
  if (memcmp(a, c, n * 32) || !memcmp(a, b, n * 32)) return fprintf(stderr, "NTT round trip failed\n"), 1;
  CHECK(pm_fr_ntt(ctx, a, n, b, log_n, PM_NTT_COSET));
  CHECK(pm_fr_ntt(ctx, b, n, c, log_n, PM_NTT_COSET | PM_NTT_INVERSE));
  if (memcmp(a, c, n * 32)) return fprintf(stderr, "coset NTT round trip failed\n"), 1;
  /* zero padding: a shorter input is the same as the padded one */
  memset(c, 0, n * 32);
  memcpy(c, a, (n / 2 + 1) * 32);
  uint64_t* d = malloc(n * 32);
  CHECK(pm_fr_ntt(ctx, c, n, b, log_n, 0));
  CHECK(pm_fr_ntt(ctx, a, n / 2 + 1, d, log_n, 0));
  if (memcmp(b, d, n * 32)) return fprintf(stderr, "zero padding differs\n"), 1;
  if (pm_fr_ntt(ctx, a, n, b, 40, 0) != PM_ERR_DOMAIN_TOO_LARGE) return fprintf(stderr, "no domain error\n"), 1;

  /* device-resident path */
  void *d_in = NULL, *d_out = NULL;
  CHECK(pm_dev_alloc(ctx, n * 32, &d_in));
  CHECK(pm_dev_alloc(ctx, n * 32, &d_out));
  CHECK(pm_dev_upload(ctx, d_in, a, n * 32));
  CHECK(pm_fr_ntt_dev(ctx, d_in, n, n, d_out, n, log_n, 1, 0, NULL));
  CHECK(pm_dev_download(ctx, c, d_out, n * 32));
  CHECK(pm_fr_ntt(ctx, a, n, b, log_n, 0));
  if (memcmp(b, c, n * 32)) return fprintf(stderr, "device NTT differs from the host-pointer call\n"), 1;

  /* MSM: sum_i s_i G == (sum_i s_i) G */
  const size_t m = n < 4096 ? n : 4096;
  uint64_t* pts = malloc(m * 96);
  uint64_t sum[4] = {0, 0, 0, 0}, lhs[18], rhs[18], lhs_xy[12], rhs_xy[12];
  for (size_t i = 0; i < m; ++i) {
    memcpy(pts + 12 * i, G1_GEN, 96);
    fr_add(sum, sum, a + 4 * i);
  }
  pm_bases* bases = NULL;
  CHECK(pm_g1_bases_upload(ctx, pts, m, &bases));
  CHECK(pm_g1_msm(ctx, bases, m, a, PM_SCALAR_MONTGOMERY, lhs));
  CHECK(pm_g1_msm(ctx, bases, 1, sum, PM_SCALAR_MONTGOMERY, rhs));
  int inf_l = 0, inf_r = 0;
  CHECK(pm_g1_to_affine(lhs, lhs_xy, &inf_l));
  CHECK(pm_g1_to_affine(rhs, rhs_xy, &inf_r));
  if (inf_l || inf_r || memcmp(lhs_xy, rhs_xy, 96)) return fprintf(stderr, "MSM identity failed\n"), 1;
  if (pm_g1_msm(ctx, bases, m + 1, a, PM_SCALAR_MONTGOMERY, lhs) != PM_ERR_LENGTH)
    return fprintf(stderr, "no length error\n"), 1;
  pm_g1_bases_free(ctx, bases);

  /* the ark call: VariableBaseMSM::multi_scalar_mul(&[G1Affine], &[BigInteger256]) */
  {
    struct ark_g1_affine* ark = calloc(m + 1, sizeof *ark);
    uint64_t* canon = malloc((m + 1) * 32);
    for (size_t i = 0; i < m; ++i) {
      memcpy(ark[i].x, G1_GEN, 48);
      memcpy(ark[i].y, G1_GEN + 6, 48);
      fr_from_mont(canon + 4 * i, a + 4 * i);
    }
    ark[m].infinity = 1; /* ark's G1Affine::zero(): (0, 1, infinity = true) -- the coordinates are not looked at */
    ark[m].y[0] = 1;
    fr_from_mont(canon + 4 * m, a); /* any scalar */
    uint64_t* packed = malloc((m + 1) * 96);
    for (size_t i = 0; i <= m; ++i) {
      if (ark[i].infinity) {
        memset(packed + 12 * i, 0, 96);
      } else {
        memcpy(packed + 12 * i, ark[i].x, 48);
        memcpy(packed + 12 * i + 6, ark[i].y, 48);
      }
    }
    uint64_t got[18], got_xy[12];
    int inf = 0;
    CHECK(pm_g1_bases_upload(ctx, packed, m + 1, &bases));
    CHECK(pm_g1_msm(ctx, bases, m + 1, canon, PM_SCALAR_CANONICAL, got));
    CHECK(pm_g1_to_affine(got, got_xy, &inf));
    if (inf || memcmp(got_xy, lhs_xy, 96)) return fprintf(stderr, "canonical-scalar (ark) MSM differs from the Montgomery one\n"), 1;
    /* the result is normalised (Z = 1), so (X, Y, Z) is ark's Jacobian GroupProjective { x, y, z } as it stands */
    if (memcmp(got, got_xy, 96)) return fprintf(stderr, "MSM result is not normalised\n"), 1;
    pm_g1_bases_free(ctx, bases);
    free(ark);
    free(canon);
    free(packed);
  }
  CHECK(pm_dev_free(ctx, d_in));
  CHECK(pm_dev_free(ctx, d_out));
  pm_shutdown(ctx);
  printf("abi_demo OK (2^%u NTT round trips, %zu-point MSM identity, ark-shaped MSM with canonical scalars)\n", log_n, m);
  return 0;
}
