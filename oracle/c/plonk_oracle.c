/*
 * CPU restatement of the reference's NTT / MSM hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (plonk-prototype_amd/) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference tree (/root/reference) holds no NTT/MSM code and no
 * tests; the path lives in un-vendored crates dusk-plonk 0.8.2 (ref:Cargo.toml:19) and
 * dusk-bls12_381 0.8 (ref:Cargo.toml:20) which cannot be fetched or built here (no
 * Rust toolchain, no network).  This file restates their published algorithms:
 *
 *   orc_fr_ntt            = EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
 *                           -> best_fft -> serial_fft            
 *   orc_fr_ntt_parallel   = the `std`-feature parallel_fft (rayon coset split, CS-3)
 *   orc_g1_msm            = msm_variable_base / pippenger        
 *                           == ark-ec 0.2 VariableBaseMSM::multi_scalar_mul
 *
 * It is checked against oracle/bigint_oracle.py (different algorithms, big-int
 * arithmetic) in tests/test_oracle_*.py.  Representation matches the Rust types:
 * Fr = 4 x u64 little-endian Montgomery (R = 2^256), Fp = 6 x u64 Montgomery
 * (R = 2^384), always fully reduced.
 *
 * Derived constants (R, R^2, -m^-1, root of unity) are computed at start-up from the
 * moduli alone, so they are an independent check on the tables in SURVEY.md sec. 8c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define ALWAYS_INLINE static inline __attribute__((always_inline))

/* ------------------------------------------------------------------ generic Montgomery */
typedef struct {
  int n;           /* limbs */
  u64 m[6];        /* modulus */
  u64 inv;         /* -m^-1 mod 2^64 */
  u64 one[6];      /* R mod m */
  u64 r2[6];       /* R^2 mod m */
} field_t;

static field_t FR, FP;

ALWAYS_INLINE int ge_n(const u64 *a, const u64 *b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
ALWAYS_INLINE u64 add_n(u64 *r, const u64 *a, const u64 *b, int n) {
  u64 c = 0;
  for (int i = 0; i < n; ++i) {
    u128 t = (u128)a[i] + b[i] + c;
    r[i] = (u64)t;
    c = (u64)(t >> 64);
  }
  return c;
}
ALWAYS_INLINE u64 sub_n(u64 *r, const u64 *a, const u64 *b, int n) {
  u64 bw = 0;
  for (int i = 0; i < n; ++i) {
    u128 t = (u128)a[i] - b[i] - bw;
    r[i] = (u64)t;
    bw = (u64)(t >> 64) & 1;
  }
  return bw;
}
ALWAYS_INLINE void f_add(u64 *r, const u64 *a, const u64 *b, const field_t *F, int n) {
  u64 t[6];
  u64 c = add_n(t, a, b, n);
  if (c || ge_n(t, F->m, n)) sub_n(t, t, F->m, n);
  memcpy(r, t, 8 * n);
}
ALWAYS_INLINE void f_sub(u64 *r, const u64 *a, const u64 *b, const field_t *F, int n) {
  u64 t[6];
  if (sub_n(t, a, b, n)) add_n(t, t, F->m, n);
  memcpy(r, t, 8 * n);
}
/* CIOS Montgomery multiplication, r = a*b/R mod m */
ALWAYS_INLINE void f_mul(u64 *r, const u64 *a, const u64 *b, const field_t *F, int n) {
  u64 t[8] = {0};
  for (int i = 0; i < n; ++i) {
    u64 c = 0;
    for (int j = 0; j < n; ++j) {
      u128 p = (u128)a[j] * b[i] + t[j] + c;
      t[j] = (u64)p;
      c = (u64)(p >> 64);
    }
    u128 s = (u128)t[n] + c;
    t[n] = (u64)s;
    t[n + 1] = (u64)(s >> 64);
    u64 q = t[0] * F->inv;
    u128 p = (u128)q * F->m[0] + t[0];
    c = (u64)(p >> 64);
    for (int j = 1; j < n; ++j) {
      p = (u128)q * F->m[j] + t[j] + c;
      t[j - 1] = (u64)p;
      c = (u64)(p >> 64);
    }
    s = (u128)t[n] + c;
    t[n - 1] = (u64)s;
    t[n] = t[n + 1] + (u64)(s >> 64);
  }
  if (t[n] || ge_n(t, F->m, n)) sub_n(t, t, F->m, n);
  memcpy(r, t, 8 * n);
}
static int is_zero_n(const u64 *a, int n) {
  u64 x = 0;
  for (int i = 0; i < n; ++i) x |= a[i];
  return x == 0;
}
static int eq_n(const u64 *a, const u64 *b, int n) { return memcmp(a, b, 8 * n) == 0; }

/* r = a^e, e little-endian limbs (plain integer) */
static void f_pow(u64 *r, const u64 *a, const u64 *e, int elimbs, const field_t *F) {
  int n = F->n;
  u64 acc[6], base[6];
  memcpy(acc, F->one, 8 * n);
  memcpy(base, a, 8 * n);
  for (int i = 0; i < elimbs * 64; ++i) {
    if ((e[i / 64] >> (i % 64)) & 1) f_mul(acc, acc, base, F, n);
    f_mul(base, base, base, F, n);
  }
  memcpy(r, acc, 8 * n);
}
static void f_inv(u64 *r, const u64 *a, const field_t *F) {
  u64 e[6], two[6] = {2, 0, 0, 0, 0, 0};
  sub_n(e, F->m, two, F->n);
  f_pow(r, a, e, F->n, F);
}
static void field_init(field_t *F, int n, const u64 *m) {
  F->n = n;
  memset(F->m, 0, sizeof F->m);
  memcpy(F->m, m, 8 * n);
  /* -m^-1 mod 2^64 by Newton iteration */
  u64 x = 1;
  for (int i = 0; i < 6; ++i) x *= 2 - m[0] * x;
  F->inv = (u64)0 - x;
  /* R mod m and R^2 mod m by repeated doubling of 1 */
  u64 t[6] = {1, 0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * 64 * n; ++i) {
    u64 c = add_n(t, t, t, n);
    if (c || ge_n(t, F->m, n)) sub_n(t, t, F->m, n);
    if (i == 64 * n - 1) memcpy(F->one, t, 8 * n);
  }
  memcpy(F->r2, t, 8 * n);
}

#define FRN 4
#define FPN 6
ALWAYS_INLINE void fr_mul(u64 *r, const u64 *a, const u64 *b) { f_mul(r, a, b, &FR, FRN); }
ALWAYS_INLINE void fr_add(u64 *r, const u64 *a, const u64 *b) { f_add(r, a, b, &FR, FRN); }
ALWAYS_INLINE void fr_sub(u64 *r, const u64 *a, const u64 *b) { f_sub(r, a, b, &FR, FRN); }
ALWAYS_INLINE void fp_mul(u64 *r, const u64 *a, const u64 *b) { f_mul(r, a, b, &FP, FPN); }
ALWAYS_INLINE void fp_add(u64 *r, const u64 *a, const u64 *b) { f_add(r, a, b, &FP, FPN); }
ALWAYS_INLINE void fp_sub(u64 *r, const u64 *a, const u64 *b) { f_sub(r, a, b, &FP, FPN); }
ALWAYS_INLINE void fp_sqr(u64 *r, const u64 *a) { f_mul(r, a, a, &FP, FPN); }
ALWAYS_INLINE void fp_dbl(u64 *r, const u64 *a) { f_add(r, a, a, &FP, FPN); }

/* ------------------------------------------------------------------ constants */
static const u64 FR_MODULUS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL,
                                  0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const u64 FP_MODULUS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL,
                                  0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
                                  0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
/* canonical (non-Montgomery) generator coordinates */
static const u64 G1_GEN_X[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL,
                                0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL,
                                0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const u64 G1_GEN_Y[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL,
                                0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL,
                                0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
#define TWO_ADICITY 32
static u64 FR_ROOT_OF_UNITY[4]; /* Montgomery, order 2^32 */
static u64 FR_GENERATOR[4];     /* Montgomery 7 */
static u64 FR_GENERATOR_INV[4];
static int g_inited = 0;

static void fr_from_u64(u64 *r, u64 v) {
  u64 t[4] = {v, 0, 0, 0};
  fr_mul(r, t, FR.r2);
}

void orc_init(void) {
  if (g_inited) return;
  field_init(&FR, FRN, FR_MODULUS);
  field_init(&FP, FPN, FP_MODULUS);
  fr_from_u64(FR_GENERATOR, 7);
  f_inv(FR_GENERATOR_INV, FR_GENERATOR, &FR);
  /* t = (r-1) >> 32 ; ROOT = 7^t */
  u64 e[4], one[4] = {1, 0, 0, 0};
  sub_n(e, FR_MODULUS, one, 4);
  for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
  f_pow(FR_ROOT_OF_UNITY, FR_GENERATOR, e, 4, &FR);
  g_inited = 1;
}

/* constant export so tests can pin them against SURVEY.md sec. 8c */
void orc_constants(u64 *fr_one, u64 *fr_r2, u64 *fr_inv, u64 *fr_root, u64 *fr_gen, u64 *fp_one,
                   u64 *fp_inv) {
  orc_init();
  memcpy(fr_one, FR.one, 32);
  memcpy(fr_r2, FR.r2, 32);
  *fr_inv = FR.inv;
  memcpy(fr_root, FR_ROOT_OF_UNITY, 32);
  memcpy(fr_gen, FR_GENERATOR, 32);
  memcpy(fp_one, FP.one, 48);
  *fp_inv = FP.inv;
}

/* ------------------------------------------------------------------ Fr helpers for tests */
void orc_fr_to_mont(u64 *io, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i) fr_mul(io + 4 * i, io + 4 * i, FR.r2);
}
void orc_fr_from_mont(u64 *io, size_t n) {
  orc_init();
  u64 one[4] = {1, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) fr_mul(io + 4 * i, io + 4 * i, one);
}
void orc_fp_to_mont(u64 *io, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i) fp_mul(io + 6 * i, io + 6 * i, FP.r2);
}
void orc_fp_from_mont(u64 *io, size_t n) {
  orc_init();
  u64 one[6] = {1, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) fp_mul(io + 6 * i, io + 6 * i, one);
}
void orc_fr_mul(u64 *r, const u64 *a, const u64 *b, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i) fr_mul(r + 4 * i, a + 4 * i, b + 4 * i);
}
void orc_fp_mul(u64 *r, const u64 *a, const u64 *b, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i) fp_mul(r + 6 * i, a + 6 * i, b + 6 * i);
}

/* splitmix64-driven uniform Fr sampling, Montgomery output (matches bigint_oracle.sample_fr
 * after to_mont): 4 words -> 256-bit value -> reduced mod r by Montgomery mul with R^2
 * (value*R^2/R = value*R mod r, i.e. to_mont of (value mod r)). */
static inline u64 splitmix64_next(u64 *st) {
  u64 z = (*st += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
void orc_fr_sample(u64 seed, size_t n, u64 *out_mont) {
  orc_init();
  u64 st = seed;
  for (size_t i = 0; i < n; ++i) {
    u64 v[4];
    for (int j = 0; j < 4; ++j) v[j] = splitmix64_next(&st);
    /* v < 2^256 may exceed r several times; CIOS needs only one operand < m */
    fr_mul(out_mont + 4 * i, v, FR.r2);
  }
}

/* ------------------------------------------------------------------ EvaluationDomain */
#define ORC_INVERSE 1u
#define ORC_COSET 2u

static void domain_gen(u64 *gen, unsigned log_n) {
  memcpy(gen, FR_ROOT_OF_UNITY, 32);
  for (unsigned i = log_n; i < TWO_ADICITY; ++i) fr_mul(gen, gen, gen);
}
static uint32_t bitreverse(uint32_t n, unsigned l) {
  uint32_t r = 0;
  for (unsigned i = 0; i < l; ++i) {
    r = (r << 1) | (n & 1);
    n >>= 1;
  }
  return r;
}
/* serial_fft: bit-reversal permutation, then log_n stages of DIT butterflies */
static void serial_fft(u64 *a, const u64 *omega, unsigned log_n) {
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n; ++k) {
    size_t rk = bitreverse((uint32_t)k, log_n);
    if (k < rk) {
      u64 t[4];
      memcpy(t, a + 4 * k, 32);
      memcpy(a + 4 * k, a + 4 * rk, 32);
      memcpy(a + 4 * rk, t, 32);
    }
  }
  size_t m = 1;
  for (unsigned s = 0; s < log_n; ++s) {
    u64 w_m[4], e[1] = {(u64)(n / (2 * m))};
    f_pow(w_m, omega, e, 1, &FR);
    for (size_t k = 0; k < n; k += 2 * m) {
      u64 w[4];
      memcpy(w, FR.one, 32);
      for (size_t j = 0; j < m; ++j) {
        u64 t[4];
        u64 *lo = a + 4 * (k + j), *hi = a + 4 * (k + j + m);
        fr_mul(t, hi, w);
        fr_sub(hi, lo, t);
        fr_add(lo, lo, t);
        fr_mul(w, w, w_m);
      }
    }
    m *= 2;
  }
}
/* rayon parallel_fft shape: split into 2^log_cpus cosets, serial_fft each, interleave */
static void parallel_fft(u64 *a, const u64 *omega, unsigned log_n, unsigned log_cpus) {
  size_t num_cpus = (size_t)1 << log_cpus;
  unsigned log_new_n = log_n - log_cpus;
  size_t new_n = (size_t)1 << log_new_n, n = (size_t)1 << log_n;
  u64 *tmp = (u64 *)calloc(n, 32);
  u64 new_omega[4], e[1] = {num_cpus};
  f_pow(new_omega, omega, e, 1, &FR);
#pragma omp parallel for schedule(static)
  for (size_t j = 0; j < num_cpus; ++j) {
    u64 *tj = tmp + 4 * j * new_n;
    u64 omega_j[4], omega_step[4], elt[4], ej[1] = {j}, es[1] = {j << log_new_n};
    f_pow(omega_j, omega, ej, 1, &FR);
    f_pow(omega_step, omega, es, 1, &FR);
    memcpy(elt, FR.one, 32);
    for (size_t i = 0; i < new_n; ++i) {
      for (size_t s = 0; s < num_cpus; ++s) {
        size_t idx = (i + (s << log_new_n)) & (n - 1);
        u64 t[4];
        fr_mul(t, a + 4 * idx, elt);
        fr_add(tj + 4 * i, tj + 4 * i, t);
        fr_mul(elt, elt, omega_step);
      }
      fr_mul(elt, elt, omega_j);
    }
    serial_fft(tj, new_omega, log_new_n);
  }
#pragma omp parallel for schedule(static)
  for (size_t idx = 0; idx < n; ++idx)
    memcpy(a + 4 * idx, tmp + 4 * ((idx & (num_cpus - 1)) * new_n + (idx >> log_cpus)), 32);
  free(tmp);
}
static void distribute_powers(u64 *a, size_t n, const u64 *g) {
  u64 p[4];
  memcpy(p, FR.one, 32);
  for (size_t i = 0; i < n; ++i) {
    fr_mul(a + 4 * i, a + 4 * i, p);
    fr_mul(p, p, g);
  }
}
/* a: 2^log_n Montgomery Fr, already zero-padded, transformed in place.
 * threads <= 1 -> serial_fft (the no-std default of ref:Cargo.toml:19);
 * threads  > 1 -> parallel_fft with log_cpus = floor(log2(threads)) (the `std` feature). */
int orc_fr_ntt(u64 *a, unsigned log_n, unsigned flags, int threads) {
  orc_init();
  if (log_n >= TWO_ADICITY) return -2;
  size_t n = (size_t)1 << log_n;
  u64 omega[4];
  domain_gen(omega, log_n);
  if (flags & ORC_INVERSE) f_inv(omega, omega, &FR);
  if ((flags & ORC_COSET) && !(flags & ORC_INVERSE)) distribute_powers(a, n, FR_GENERATOR);
  unsigned log_cpus = 0;
  while (threads > 1 && (2u << log_cpus) <= (unsigned)threads) ++log_cpus;
  if (log_cpus == 0 || log_n <= log_cpus) {
    serial_fft(a, omega, log_n);
  } else {
#ifdef _OPENMP
    omp_set_num_threads(1 << log_cpus);
#endif
    parallel_fft(a, omega, log_n, log_cpus);
  }
  if (flags & ORC_INVERSE) {
    u64 ninv[4];
    fr_from_u64(ninv, (u64)n);
    f_inv(ninv, ninv, &FR);
    for (size_t i = 0; i < n; ++i) fr_mul(a + 4 * i, a + 4 * i, ninv);
    if (flags & ORC_COSET) distribute_powers(a, n, FR_GENERATOR_INV);
  }
  return 0;
}

/* ------------------------------------------------------------------ G1 (Jacobian) */
typedef struct { u64 x[6], y[6], z[6]; } g1j_t;   /* z == 0 <=> identity */
typedef struct { u64 x[6], y[6]; } g1a_t;         /* (0,0) <=> identity at the ABI */

static void g1j_set_identity(g1j_t *p) {
  memcpy(p->x, FP.one, 48);
  memcpy(p->y, FP.one, 48);
  memset(p->z, 0, 48);
}
static int g1a_is_identity(const g1a_t *p) { return is_zero_n(p->x, 6) && is_zero_n(p->y, 6); }

static void g1j_double(g1j_t *r, const g1j_t *p) {
  if (is_zero_n(p->z, 6)) { *r = *p; return; }
  /* dbl-2009-l, a = 0 */
  u64 A[6], B[6], C[6], D[6], E[6], F[6], t[6], x3[6], y3[6], z3[6];
  fp_sqr(A, p->x);
  fp_sqr(B, p->y);
  fp_sqr(C, B);
  fp_add(t, p->x, B);
  fp_sqr(t, t);
  fp_sub(t, t, A);
  fp_sub(t, t, C);
  fp_dbl(D, t);
  fp_dbl(E, A);
  fp_add(E, E, A);
  fp_sqr(F, E);
  fp_dbl(t, D);
  fp_sub(x3, F, t);
  fp_mul(z3, p->y, p->z);
  fp_dbl(z3, z3);
  fp_sub(t, D, x3);
  fp_mul(y3, E, t);
  fp_dbl(C, C);
  fp_dbl(C, C);
  fp_dbl(C, C);
  fp_sub(y3, y3, C);
  memcpy(r->x, x3, 48);
  memcpy(r->y, y3, 48);
  memcpy(r->z, z3, 48);
}
static void g1j_add_mixed(g1j_t *r, const g1j_t *p, const g1a_t *q) {
  if (g1a_is_identity(q)) { *r = *p; return; }
  if (is_zero_n(p->z, 6)) {
    memcpy(r->x, q->x, 48);
    memcpy(r->y, q->y, 48);
    memcpy(r->z, FP.one, 48);
    return;
  }
  /* madd-2007-bl */
  u64 Z1Z1[6], U2[6], S2[6], H[6], HH[6], I[6], J[6], rr[6], V[6], t[6], x3[6], y3[6], z3[6];
  fp_sqr(Z1Z1, p->z);
  fp_mul(U2, q->x, Z1Z1);
  fp_mul(S2, q->y, p->z);
  fp_mul(S2, S2, Z1Z1);
  if (eq_n(U2, p->x, 6)) {
    if (eq_n(S2, p->y, 6)) { g1j_double(r, p); return; }
    g1j_set_identity(r);
    return;
  }
  fp_sub(H, U2, p->x);
  fp_sqr(HH, H);
  fp_dbl(I, HH);
  fp_dbl(I, I);
  fp_mul(J, H, I);
  fp_sub(rr, S2, p->y);
  fp_dbl(rr, rr);
  fp_mul(V, p->x, I);
  fp_sqr(x3, rr);
  fp_sub(x3, x3, J);
  fp_sub(x3, x3, V);
  fp_sub(x3, x3, V);
  fp_sub(t, V, x3);
  fp_mul(y3, rr, t);
  fp_mul(t, p->y, J);
  fp_dbl(t, t);
  fp_sub(y3, y3, t);
  fp_add(z3, p->z, H);
  fp_sqr(z3, z3);
  fp_sub(z3, z3, Z1Z1);
  fp_sub(z3, z3, HH);
  memcpy(r->x, x3, 48);
  memcpy(r->y, y3, 48);
  memcpy(r->z, z3, 48);
}
static void g1j_add(g1j_t *r, const g1j_t *p, const g1j_t *q) {
  if (is_zero_n(p->z, 6)) { *r = *q; return; }
  if (is_zero_n(q->z, 6)) { *r = *p; return; }
  /* add-2007-bl */
  u64 Z1Z1[6], Z2Z2[6], U1[6], U2[6], S1[6], S2[6], H[6], I[6], J[6], rr[6], V[6], t[6];
  u64 x3[6], y3[6], z3[6];
  fp_sqr(Z1Z1, p->z);
  fp_sqr(Z2Z2, q->z);
  fp_mul(U1, p->x, Z2Z2);
  fp_mul(U2, q->x, Z1Z1);
  fp_mul(S1, p->y, q->z);
  fp_mul(S1, S1, Z2Z2);
  fp_mul(S2, q->y, p->z);
  fp_mul(S2, S2, Z1Z1);
  if (eq_n(U1, U2, 6)) {
    if (eq_n(S1, S2, 6)) { g1j_double(r, p); return; }
    g1j_set_identity(r);
    return;
  }
  fp_sub(H, U2, U1);
  fp_dbl(I, H);
  fp_sqr(I, I);
  fp_mul(J, H, I);
  fp_sub(rr, S2, S1);
  fp_dbl(rr, rr);
  fp_mul(V, U1, I);
  fp_sqr(x3, rr);
  fp_sub(x3, x3, J);
  fp_sub(x3, x3, V);
  fp_sub(x3, x3, V);
  fp_sub(t, V, x3);
  fp_mul(y3, rr, t);
  fp_mul(t, S1, J);
  fp_dbl(t, t);
  fp_sub(y3, y3, t);
  fp_add(z3, p->z, q->z);
  fp_sqr(z3, z3);
  fp_sub(z3, z3, Z1Z1);
  fp_sub(z3, z3, Z2Z2);
  fp_mul(z3, z3, H);
  memcpy(r->x, x3, 48);
  memcpy(r->y, y3, 48);
  memcpy(r->z, z3, 48);
}
static void g1j_to_affine(g1a_t *r, const g1j_t *p) {
  if (is_zero_n(p->z, 6)) { memset(r, 0, sizeof *r); return; }
  u64 zi[6], zi2[6], zi3[6];
  f_inv(zi, p->z, &FP);
  fp_sqr(zi2, zi);
  fp_mul(zi3, zi2, zi);
  fp_mul(r->x, p->x, zi2);
  fp_mul(r->y, p->y, zi3);
}

/* Jacobian (X,Y,Z) -> affine (x,y) Montgomery; identity -> (0,0).  in: 18 limbs, out: 12 */
void orc_g1_jacobian_to_affine(const u64 *xyz, u64 *xy) {
  orc_init();
  g1j_to_affine((g1a_t *)xy, (const g1j_t *)xyz);
}
/* homogeneous projective (x = X/Z, y = Y/Z; dusk/zkcrypto G1Projective) -> affine */
void orc_g1_projective_to_affine(const u64 *xyz, u64 *xy) {
  orc_init();
  const u64 *X = xyz, *Y = xyz + 6, *Z = xyz + 12;
  if (is_zero_n(Z, 6)) { memset(xy, 0, 96); return; }
  u64 zi[6];
  f_inv(zi, Z, &FP);
  fp_mul(xy, X, zi);
  fp_mul(xy + 6, Y, zi);
}
int orc_g1_is_on_curve(const u64 *xy) {
  orc_init();
  const g1a_t *p = (const g1a_t *)xy;
  if (g1a_is_identity(p)) return 1;
  u64 l[6], r[6], four[6] = {4, 0, 0, 0, 0, 0};
  fp_mul(four, four, FP.r2);
  fp_sqr(l, p->y);
  fp_sqr(r, p->x);
  fp_mul(r, r, p->x);
  fp_add(r, r, four);
  return eq_n(l, r, 6);
}
void orc_g1_generator(u64 *xy) {
  orc_init();
  fp_mul(xy, G1_GEN_X, FP.r2);
  fp_mul(xy + 6, G1_GEN_Y, FP.r2);
}
/* canonical (non-Montgomery) little-endian scalar k, 4 limbs; affine in/out */
static void g1_scalar_mul(g1j_t *r, const g1a_t *p, const u64 *k) {
  g1j_t acc;
  g1j_set_identity(&acc);
  for (int i = 255; i >= 0; --i) {
    g1j_double(&acc, &acc);
    if ((k[i / 64] >> (i % 64)) & 1) g1j_add_mixed(&acc, &acc, p);
  }
  *r = acc;
}
void orc_g1_mul(const u64 *xy, const u64 *k_canonical, u64 *out_xy) {
  orc_init();
  g1j_t r;
  g1_scalar_mul(&r, (const g1a_t *)xy, k_canonical);
  g1j_to_affine((g1a_t *)out_xy, &r);
}
void orc_g1_add_affine(const u64 *a, const u64 *b, u64 *out_xy) {
  orc_init();
  g1j_t p;
  g1j_set_identity(&p);
  g1j_add_mixed(&p, &p, (const g1a_t *)a);
  g1j_add_mixed(&p, &p, (const g1a_t *)b);
  g1j_to_affine((g1a_t *)out_xy, &p);
}

/* batch Jacobian -> affine with one inversion (Montgomery's trick) */
static void batch_to_affine(g1a_t *out, const g1j_t *in, size_t n) {
  u64 *pref = (u64 *)malloc(48 * (n + 1));
  u64 acc[6];
  memcpy(acc, FP.one, 48);
  for (size_t i = 0; i < n; ++i) {
    memcpy(pref + 6 * i, acc, 48);
    if (!is_zero_n(in[i].z, 6)) fp_mul(acc, acc, in[i].z);
  }
  u64 inv[6];
  f_inv(inv, acc, &FP);
  for (size_t i = n; i-- > 0;) {
    if (is_zero_n(in[i].z, 6)) { memset(&out[i], 0, sizeof(g1a_t)); continue; }
    u64 zi[6], zi2[6], zi3[6];
    fp_mul(zi, inv, pref + 6 * i);
    fp_mul(inv, inv, in[i].z);
    fp_sqr(zi2, zi);
    fp_mul(zi3, zi2, zi);
    fp_mul(out[i].x, in[i].x, zi2);
    fp_mul(out[i].y, in[i].y, zi3);
  }
  free(pref);
}

/* Synthetic SRS-like bases with known discrete logs: P_i = (k0 + i*d) * G.
 * k0, d canonical 4-limb scalars.  Expected MSM result = (sum_i s_i*(k0+i*d)) * G, which a
 * test can evaluate with one independent scalar multiplication.  OpenMP over chunks. */
void orc_g1_bases_arith(const u64 *k0, const u64 *d, size_t n, u64 *out_xy, int threads) {
  orc_init();
  g1a_t G, D;
  orc_g1_generator((u64 *)&G);
  orc_g1_mul((const u64 *)&G, d, (u64 *)&D);
  const size_t CH = 4096;
  size_t nch = (n + CH - 1) / CH;
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
  for (size_t c = 0; c < nch; ++c) {
    size_t lo = c * CH, hi = lo + CH > n ? n : lo + CH;
    /* k = k0 + lo*d mod r, via Montgomery Fr arithmetic */
    u64 km[4], dm[4], lm[4], one[4] = {1, 0, 0, 0};
    fr_mul(km, k0, FR.r2);
    fr_mul(dm, d, FR.r2);
    fr_from_u64(lm, (u64)lo);
    fr_mul(lm, lm, dm);
    fr_add(km, km, lm);
    fr_mul(km, km, one);
    g1j_t *buf = (g1j_t *)malloc(sizeof(g1j_t) * (hi - lo));
    g1_scalar_mul(&buf[0], &G, km);
    for (size_t i = 1; i < hi - lo; ++i) g1j_add_mixed(&buf[i], &buf[i - 1], &D);
    batch_to_affine((g1a_t *)out_xy + lo, buf, hi - lo);
    free(buf);
  }
}

/* ------------------------------------------------------------------ msm_variable_base */
#define ORC_SCALAR_MONTGOMERY 0u
#define ORC_SCALAR_CANONICAL 1u

static unsigned ln_without_floats(size_t a) {
  /* ark/dusk: (log2(a) * 69 / 100), log2 = ceil for non powers of two */
  unsigned l = 0;
  while (((size_t)1 << l) < a) ++l;
  return l * 69 / 100;
}
unsigned orc_msm_window_bits(size_t n) { return n < 32 ? 3 : ln_without_floats(n) + 2; }

/* points: n x (x[6], y[6]) Montgomery, (0,0) = identity.  scalars: n x 4 limbs.
 * out: Jacobian X,Y,Z (18 limbs).  threads>1 parallelises over windows (the rayon shape). */
int orc_g1_msm(const u64 *points_xy, const u64 *scalars, size_t n, unsigned scalar_form,
               u64 *out_xyz, int threads) {
  orc_init();
  const g1a_t *pts = (const g1a_t *)points_xy;
  /* scalar.reduce(): Montgomery -> canonical integer */
  u64 *sc = (u64 *)malloc(32 * (n ? n : 1));
  memcpy(sc, scalars, 32 * n);
  if (scalar_form == ORC_SCALAR_MONTGOMERY) orc_fr_from_mont(sc, n);
  unsigned c = orc_msm_window_bits(n);
  unsigned num_bits = 255;
  unsigned nwin = (num_bits + c - 1) / c;
  g1j_t *wsum = (g1j_t *)malloc(sizeof(g1j_t) * nwin);
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
  for (unsigned w = 0; w < nwin; ++w) {
    unsigned w_start = w * c;
    size_t nb = ((size_t)1 << c) - 1;
    g1j_t *buckets = (g1j_t *)malloc(sizeof(g1j_t) * nb);
    for (size_t b = 0; b < nb; ++b) g1j_set_identity(&buckets[b]);
    g1j_t res;
    g1j_set_identity(&res);
    for (size_t i = 0; i < n; ++i) {
      const u64 *s = sc + 4 * i;
      if (is_zero_n(s, 4)) continue;
      if (s[0] == 1 && s[1] == 0 && s[2] == 0 && s[3] == 0) {
        if (w_start == 0) g1j_add_mixed(&res, &res, &pts[i]);   /* unit-scalar shortcut */
        continue;
      }
      /* (s >> w_start) mod 2^c */
      unsigned limb = w_start / 64, off = w_start % 64;
      u64 d = s[limb] >> off;
      if (off + c > 64 && limb + 1 < 4) d |= s[limb + 1] << (64 - off);
      d &= ((u64)1 << c) - 1;
      if (d) g1j_add_mixed(&buckets[d - 1], &buckets[d - 1], &pts[i]);
    }
    g1j_t running;
    g1j_set_identity(&running);
    for (size_t b = nb; b-- > 0;) {
      g1j_add(&running, &running, &buckets[b]);
      g1j_add(&res, &res, &running);
    }
    wsum[w] = res;
    free(buckets);
  }
  /* fold windows high -> low */
  g1j_t total;
  g1j_set_identity(&total);
  for (unsigned w = nwin; w-- > 1;) {
    g1j_add(&total, &total, &wsum[w]);
    for (unsigned k = 0; k < c; ++k) g1j_double(&total, &total);
  }
  g1j_add(&total, &total, &wsum[0]);
  memcpy(out_xyz, &total, sizeof total);
  free(wsum);
  free(sc);
  return 0;
}

/* sum_i s_i * (k0 + i*d) mod r for canonical s (Montgomery if scalar_form==0): the discrete log
 * of the MSM result over orc_g1_bases_arith bases.  out: canonical 4 limbs. */
void orc_expected_dlog(const u64 *scalars, size_t n, unsigned scalar_form, const u64 *k0,
                       const u64 *d, u64 *out_canonical) {
  orc_init();
  u64 acc[4] = {0, 0, 0, 0}, km[4], dm[4], one[4] = {1, 0, 0, 0};
  fr_mul(km, k0, FR.r2);
  fr_mul(dm, d, FR.r2);
  for (size_t i = 0; i < n; ++i) {
    u64 s[4], t[4];
    memcpy(s, scalars + 4 * i, 32);
    if (scalar_form == ORC_SCALAR_CANONICAL) fr_mul(s, s, FR.r2);
    fr_mul(t, s, km);
    fr_add(acc, acc, t);
    fr_add(km, km, dm);
  }
  fr_mul(out_canonical, acc, one);
}

/* ------------------------------------------------------------------ polynomial helpers (next rows N1/N2)
 * Restatements of dusk_plonk::fft::{Polynomial, Evaluations} / util helpers the prover rounds use
 * around the NTT/MSM calls (dusk-plonk 0.8.2 pinned at
 * ref:Cargo.toml:19).  All vectors are Montgomery Fr (4 limbs). */
/* Evaluations / Polynomial coefficient-wise ops: op 0 add, 1 sub, 2 mul; b_len == 1 broadcasts */
void orc_fr_vec_op(int op, const u64 *a, const u64 *b, size_t b_len, u64 *out, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i) {
    const u64 *y = b + 4 * (b_len == 1 ? 0 : i);
    if (op == 0) fr_add(out + 4 * i, a + 4 * i, y);
    else if (op == 1) fr_sub(out + 4 * i, a + 4 * i, y);
    else fr_mul(out + 4 * i, a + 4 * i, y);
  }
}
/* util::batch_inversion: zeros stay zero */
void orc_fr_batch_inverse(u64 *v, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; ++i)
    if (!is_zero_n(v + 4 * i, 4)) f_inv(v + 4 * i, v + 4 * i, &FR);
}
/* Polynomial::evaluate: Horner */
void orc_fr_poly_evaluate(const u64 *coeffs, size_t n, const u64 *point, u64 *out) {
  orc_init();
  u64 acc[4] = {0, 0, 0, 0};
  for (size_t i = n; i-- > 0;) {
    fr_mul(acc, acc, point);
    fr_add(acc, acc, coeffs + 4 * i);
  }
  memcpy(out, acc, 32);
}
/* Polynomial::ruffini: quotient of division by (X - z); out has n-1 coefficients (n >= 1) */
void orc_fr_poly_ruffini(const u64 *coeffs, size_t n, const u64 *z, u64 *out) {
  orc_init();
  u64 q[4] = {0, 0, 0, 0};
  for (size_t i = n; i-- > 1;) {
    u64 t[4];
    fr_mul(t, q, z);
    fr_add(q, t, coeffs + 4 * i);
    memcpy(out + 4 * (i - 1), q, 32);
  }
}
/* grand product of the permutation argument: out[0] = 1, out[i] = prod_{j<i} a[j] */
void orc_fr_prefix_product(const u64 *a, size_t n, u64 *out) {
  orc_init();
  u64 acc[4];
  memcpy(acc, FR.one, 32);
  for (size_t i = 0; i < n; ++i) {
    memcpy(out + 4 * i, acc, 32);
    fr_mul(acc, acc, a + 4 * i);
  }
}

/* ------------------------------------------------------------------ PLONK prover rounds (N1)
 * The pointwise steps of dusk-plonk 0.8.2's prover (ref:Cargo.toml:19; permutation::
 * compute_permutation_poly, quotient_poly::compute, linearisation_poly::compute) for the arithmetic
 * gate and the 4-wire permutation, written from the protocol equations.  Used by
 * oracle/cpu_prover.py: the CPU baseline of a full proof and a fast second oracle for the GPU
 * prover at sizes the big-int oracle cannot reach.  PARITY UNPINNED (no upstream vectors). */
static void fr_from_small(u64 *r, u64 v) { fr_from_u64(r, v); }

/* out[i] = scale * base^i */
void orc_fr_powers(const u64 *base, const u64 *scale, size_t n, u64 *out) {
  orc_init();
  u64 cur[4];
  memcpy(cur, scale, 32);
  for (size_t i = 0; i < n; ++i) {
    memcpy(out + 4 * i, cur, 32);
    fr_mul(cur, cur, base);
  }
}
/* out = sum_j coeffs[j] * vecs[j] */
void orc_fr_lincomb(unsigned k, const u64 *const *vecs, const u64 *coeffs, size_t n, u64 *out, int threads) {
  orc_init();
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    for (unsigned j = 0; j < k; ++j) {
      fr_mul(t, vecs[j] + 4 * i, coeffs + 4 * j);
      fr_add(acc, acc, t);
    }
    memcpy(out + 4 * i, acc, 32);
  }
}
/* util::batch_inversion with Montgomery's trick (zeros stay zero); one inversion per thread block */
void orc_fr_batch_inverse_trick(u64 *v, size_t n, int threads) {
  orc_init();
  const int T = threads > 0 ? threads : 1;
#pragma omp parallel for num_threads(T) schedule(static)
  for (int b = 0; b < T; ++b) {
    const size_t lo = n * (size_t)b / T, hi = n * (size_t)(b + 1) / T;
    if (hi <= lo) continue;
    u64 *pre = (u64 *)malloc(32 * (hi - lo));
    u64 acc[4];
    memcpy(acc, FR.one, 32);
    for (size_t i = lo; i < hi; ++i) {
      memcpy(pre + 4 * (i - lo), acc, 32);
      if (!is_zero_n(v + 4 * i, 4)) fr_mul(acc, acc, v + 4 * i);
    }
    f_inv(acc, acc, &FR);
    for (size_t i = hi; i-- > lo;) {
      if (is_zero_n(v + 4 * i, 4)) continue;
      u64 t[4];
      fr_mul(t, acc, pre + 4 * (i - lo));
      fr_mul(acc, acc, v + 4 * i);
      memcpy(v + 4 * i, t, 32);
    }
    free(pre);
  }
}
/* num[i] = prod_j (w_j[i] + beta k_j roots[i] + gamma), den[i] = prod_j (w_j[i] + beta sigma_j[i] + gamma);
 * k = {1, 7, 13, 17} */
void orc_plonk_perm_terms(const u64 *const *w, const u64 *const *sig, const u64 *roots, const u64 *beta,
                          const u64 *gamma, size_t n, u64 *num, u64 *den, int threads) {
  orc_init();
  u64 bk[4][4];
  const u64 ks[4] = {1, 7, 13, 17};
  for (int j = 0; j < 4; ++j) {
    u64 kj[4];
    fr_from_small(kj, ks[j]);
    fr_mul(bk[j], beta, kj);
  }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    u64 a[4], b[4], t[4];
    memcpy(a, FR.one, 32);
    memcpy(b, FR.one, 32);
    for (int j = 0; j < 4; ++j) {
      fr_mul(t, bk[j], roots + 4 * i);
      fr_add(t, t, w[j] + 4 * i);
      fr_add(t, t, gamma);
      fr_mul(a, a, t);
      fr_mul(t, beta, sig[j] + 4 * i);
      fr_add(t, t, w[j] + 4 * i);
      fr_add(t, t, gamma);
      fr_mul(b, b, t);
    }
    memcpy(num + 4 * i, a, 32);
    memcpy(den + 4 * i, b, 32);
  }
}
/* ---- the widget identities of dusk-plonk 0.8 (proof_system::widget::{range, logic, ecc::scalar_mul::fixed_base,
 * ecc::curve_addition}), restated from the published design: each returns the factor its selector multiplies,
 * separation challenge included.  Same formulas as oracle/plonk_rounds_oracle.py (the big-int restatement). */
static void w_delta(u64 *r, const u64 *f) { /* f (f-1)(f-2)(f-3) */
  u64 c[4], t[4], acc[4];
  memcpy(acc, f, 32);
  for (u64 k = 1; k <= 3; ++k) {
    fr_from_small(c, k);
    fr_sub(t, f, c);
    fr_mul(acc, acc, t);
  }
  memcpy(r, acc, 32);
}
static void w_quad(u64 *r, const u64 *hi, const u64 *lo) { /* hi - 4 lo */
  u64 four[4], t[4];
  fr_from_small(four, 4);
  fr_mul(t, four, lo);
  fr_sub(r, hi, t);
}
static void w_range(u64 *r, const u64 *sep, const u64 *a, const u64 *b, const u64 *c, const u64 *d, const u64 *dn) {
  u64 k[4], kp[4], q[4], t[4], acc[4];
  fr_mul(k, sep, sep);
  w_quad(q, c, d); w_delta(acc, q);
  memcpy(kp, k, 32);
  w_quad(q, b, c); w_delta(t, q); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(kp, kp, k);
  w_quad(q, a, b); w_delta(t, q); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(kp, kp, k);
  w_quad(q, dn, a); w_delta(t, q); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(r, acc, sep);
}
static void w_logic(u64 *r, const u64 *sep, const u64 *a, const u64 *an, const u64 *b, const u64 *bn, const u64 *c,
                    const u64 *d, const u64 *dn, const u64 *qc) {
  u64 k[4], kp[4], qa[4], qb[4], qd[4], t[4], u[4], acc[4], s[4], in[4], f[4], e[4], bb[4], cst[4];
  fr_mul(k, sep, sep);
  w_quad(qa, an, a); w_quad(qb, bn, b); w_quad(qd, dn, d);
  w_delta(acc, qa);
  memcpy(kp, k, 32);
  w_delta(t, qb); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(kp, kp, k);
  w_delta(t, qd); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(kp, kp, k);
  fr_mul(t, qa, qb); fr_sub(t, c, t); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(kp, kp, k);
  /* delta_xor_and(qa, qb, w = c, qd, q_c) */
  fr_add(s, qa, qb);
  fr_from_small(cst, 4); fr_mul(in, cst, c);
  fr_from_small(cst, 18); fr_mul(t, cst, s); fr_sub(in, in, t);
  fr_from_small(cst, 81); fr_add(in, in, cst);                         /* 4w - 18(a+b) + 81 */
  fr_mul(in, c, in);
  fr_mul(t, qa, qa); fr_mul(u, qb, qb); fr_add(t, t, u);
  fr_from_small(cst, 18); fr_mul(t, cst, t); fr_add(in, in, t);        /* + 18 (a^2 + b^2) */
  fr_from_small(cst, 81); fr_mul(t, cst, s); fr_sub(in, in, t);
  fr_from_small(cst, 83); fr_add(in, in, cst);                         /* - 81(a+b) + 83 */
  fr_mul(f, c, in);
  fr_add(t, s, qd); fr_from_small(cst, 3); fr_mul(e, cst, t);
  fr_add(t, f, f); fr_sub(e, e, t);                                    /* 3(a+b+c) - 2f */
  fr_from_small(cst, 9); fr_mul(bb, cst, qd);
  fr_from_small(cst, 3); fr_mul(t, cst, s); fr_sub(bb, bb, t);
  fr_mul(bb, qc, bb);                                                  /* q_c (9c - 3(a+b)) */
  fr_add(t, bb, e); fr_mul(t, t, kp); fr_add(acc, acc, t);
  fr_mul(r, acc, sep);
}
static void w_edwards_d(u64 *r) {
  u64 a[4], b[4];
  fr_from_small(a, 10240); fr_from_small(b, 10241);
  f_inv(b, b, &FR);
  fr_mul(a, a, b);
  u64 z[4] = {0, 0, 0, 0};
  fr_sub(r, z, a);
}
static void w_fixed(u64 *r, const u64 *sep, const u64 *edw, const u64 *a, const u64 *an, const u64 *b, const u64 *bn,
                    const u64 *c, const u64 *d, const u64 *dn, const u64 *ql, const u64 *qr, const u64 *qc) {
  u64 k[4], k2[4], k3[4], bit[4], t[4], u[4], acc[4], ya[4], xa[4], dxy[4];
  fr_mul(k, sep, sep); fr_mul(k2, k, k); fr_mul(k3, k2, k);
  fr_add(t, d, d); fr_sub(bit, dn, t);
  fr_sub(t, bit, FR.one); fr_mul(acc, bit, t); fr_add(t, bit, FR.one); fr_mul(acc, acc, t);   /* bit (bit-1)(bit+1) */
  fr_mul(t, bit, bit); fr_sub(u, qr, FR.one); fr_mul(ya, t, u); fr_add(ya, ya, FR.one);
  fr_mul(xa, ql, bit);
  fr_mul(t, bit, qc); fr_sub(t, t, c); fr_mul(t, t, k); fr_add(acc, acc, t);
  fr_mul(dxy, c, a); fr_mul(dxy, dxy, b); fr_mul(dxy, dxy, edw);
  fr_mul(t, an, dxy); fr_add(t, an, t); fr_mul(u, a, ya); fr_sub(t, t, u); fr_mul(u, b, xa); fr_sub(t, t, u);
  fr_mul(t, t, k2); fr_add(acc, acc, t);
  fr_mul(t, bn, dxy); fr_sub(t, bn, t); fr_mul(u, b, ya); fr_sub(t, t, u); fr_mul(u, a, xa); fr_sub(t, t, u);
  fr_mul(t, t, k3); fr_add(acc, acc, t);
  fr_mul(r, acc, sep);
}
static void w_var(u64 *r, const u64 *sep, const u64 *edw, const u64 *a, const u64 *an, const u64 *b, const u64 *bn,
                  const u64 *c, const u64 *d, const u64 *dn) {
  u64 k[4], k2[4], y1x2[4], y1y2[4], x1x2[4], t[4], u[4], acc[4], dd[4];
  fr_mul(k, sep, sep); fr_mul(k2, k, k);
  fr_mul(y1x2, b, c); fr_mul(y1y2, b, d); fr_mul(x1x2, a, c);
  fr_mul(acc, a, d); fr_sub(acc, acc, dn);
  fr_mul(dd, dn, y1x2); fr_mul(dd, dd, edw);
  fr_add(t, dn, y1x2); fr_mul(u, an, dd); fr_add(u, an, u); fr_sub(t, t, u); fr_mul(t, t, k); fr_add(acc, acc, t);
  fr_add(t, y1y2, x1x2); fr_mul(u, bn, dd); fr_sub(u, bn, u); fr_sub(t, t, u); fr_mul(t, t, k2); fr_add(acc, acc, t);
  fr_mul(r, acc, sep);
}
/* the four widget factors at one row / one set of evaluations: out[0..3] = range, logic, fixed, var */
void orc_plonk_widget_values(const u64 *seps /* range logic fixed var */, const u64 *row /* a b c d an bn dn ql qr qc */,
                             u64 *out) {
  orc_init();
  u64 edw[4];
  w_edwards_d(edw);
  const u64 *a = row, *b = row + 4, *c = row + 8, *d = row + 12, *an = row + 16, *bn = row + 20, *dn = row + 24,
            *ql = row + 28, *qr = row + 32, *qc = row + 36;
  w_range(out, seps, a, b, c, d, dn);
  w_logic(out + 4, seps + 4, a, an, b, bn, c, d, dn, qc);
  w_fixed(out + 8, seps + 8, edw, a, an, b, bn, c, d, dn, ql, qr, qc);
  w_var(out + 12, seps + 12, edw, a, an, b, bn, c, d, dn);
}

/* Quotient on the 4n coset, x[i] = 7 w_4n^i.  ptrs: w0..w3, z, q_m, q_l, q_r, q_o, q_4, q_c, pi, s0..s3, l1, x,
 * q_arith, q_range, q_logic, q_fixed_group_add, q_variable_group_add (23 arrays of 4n); seps: the range, logic,
 * fixed-base and variable-base separation challenges.
 * t[i] = (q_arith arith + widgets + pi + alpha (z prod id - z(wX) prod copy) + alpha^2 (z - 1) l1) / (x^n - 1) */
void orc_plonk_quotient(const u64 *const *p, size_t n, const u64 *alpha, const u64 *beta, const u64 *gamma,
                        const u64 *seps, u64 *out, int threads) {
  orc_init();
  const size_t n4 = 4 * n;
  const u64 *w[4] = {p[0], p[1], p[2], p[3]}, *z = p[4], *qm = p[5], *ql = p[6], *qr = p[7], *qo = p[8], *q4 = p[9],
            *qc = p[10], *pi = p[11], *s[4] = {p[12], p[13], p[14], p[15]}, *l1 = p[16], *x = p[17], *qarith = p[18],
            *qrange = p[19], *qlogic = p[20], *qfixed = p[21], *qvar = p[22];
  u64 bk[4][4], alpha2[4], zh_inv[4][4], edw[4];
  const u64 ks[4] = {1, 7, 13, 17};
  for (int j = 0; j < 4; ++j) {
    u64 kj[4];
    fr_from_small(kj, ks[j]);
    fr_mul(bk[j], beta, kj);
  }
  fr_mul(alpha2, alpha, alpha);
  w_edwards_d(edw);
  for (int j = 0; j < 4 && (size_t)j < n4; ++j) { /* x^n - 1 only depends on i mod 4 */
    u64 e[1] = {(u64)n}, t[4];
    f_pow(t, x + 4 * j, e, 1, &FR);
    fr_sub(t, t, FR.one);
    f_inv(zh_inv[j], t, &FR);
  }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (size_t i = 0; i < n4; ++i) {
    const size_t nx = (i + 4) % n4;
    const u64 *a = w[0] + 4 * i, *b = w[1] + 4 * i, *c = w[2] + 4 * i, *d = w[3] + 4 * i;
    const u64 *an = w[0] + 4 * nx, *bn = w[1] + 4 * nx, *dn = w[3] + 4 * nx;
    u64 g[4], t[4], id[4], cp[4];
    fr_mul(t, a, b);
    fr_mul(g, t, qm + 4 * i);
    fr_mul(t, ql + 4 * i, a); fr_add(g, g, t);
    fr_mul(t, qr + 4 * i, b); fr_add(g, g, t);
    fr_mul(t, qo + 4 * i, c); fr_add(g, g, t);
    fr_mul(t, q4 + 4 * i, d); fr_add(g, g, t);
    fr_add(g, g, qc + 4 * i);
    fr_mul(g, g, qarith + 4 * i);
    w_range(t, seps, a, b, c, d, dn); fr_mul(t, t, qrange + 4 * i); fr_add(g, g, t);
    w_logic(t, seps + 4, a, an, b, bn, c, d, dn, qc + 4 * i); fr_mul(t, t, qlogic + 4 * i); fr_add(g, g, t);
    w_fixed(t, seps + 8, edw, a, an, b, bn, c, d, dn, ql + 4 * i, qr + 4 * i, qc + 4 * i);
    fr_mul(t, t, qfixed + 4 * i); fr_add(g, g, t);
    w_var(t, seps + 12, edw, a, an, b, bn, c, d, dn); fr_mul(t, t, qvar + 4 * i); fr_add(g, g, t);
    fr_add(g, g, pi + 4 * i);
    memcpy(id, z + 4 * i, 32);
    memcpy(cp, z + 4 * nx, 32);
    for (int j = 0; j < 4; ++j) {
      fr_mul(t, bk[j], x + 4 * i);
      fr_add(t, t, w[j] + 4 * i);
      fr_add(t, t, gamma);
      fr_mul(id, id, t);
      fr_mul(t, beta, s[j] + 4 * i);
      fr_add(t, t, w[j] + 4 * i);
      fr_add(t, t, gamma);
      fr_mul(cp, cp, t);
    }
    fr_sub(id, id, cp);
    fr_mul(id, id, alpha);
    fr_add(g, g, id);
    fr_sub(t, z + 4 * i, FR.one);
    fr_mul(t, t, l1 + 4 * i);
    fr_mul(t, t, alpha2);
    fr_add(g, g, t);
    fr_mul(out + 4 * i, g, zh_inv[i & 3]);
  }
}
