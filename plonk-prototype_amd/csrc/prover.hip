// Host-only C++: the five PLONK prover rounds behind ONE C-ABI call (pm_plonk_preprocess /
// pm_plonk_prove), so that a Rust (or Go, or C) prover needs no host logic of its own between the
// NTT and MSM kernels -- SURVEY.md section 8f rows N1 + N2 + N3, BASELINE.json configs[3].
//
// Restates dusk_plonk::proof_system::Prover::prove_with_preprocessed (dusk-plonk 0.8.2,
// ref:Cargo.toml:19; not in the reference tree) for the arithmetic gate and the 4-wire permutation,
// with a Merlin / STROBE-128 transcript (merlin is a dependency of dusk-plonk).  The same sequence
// exists in Python (plonk-prototype_amd/prover.py); tests require the two to produce identical
// proofs.  Everything here is sequencing and scalar arithmetic on a dozen field elements: the
// vector work is done by the library's own entry points, called like any client would call them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plonk_mi355x.h"
#include "host_field.h"

namespace {
using pm::host::HFp;
using pm::host::HFr;
typedef uint64_t u64;
const pm::host::Field<4>& FRF() { return pm::host::FR(); }

HFr fr_u64(u64 v) { return pm::host::from_u64(v, FRF()); }
HFr fmul(const HFr& a, const HFr& b) { return pm::host::mul(a, b, FRF()); }
HFr fadd(const HFr& a, const HFr& b) { return pm::host::add(a, b, FRF()); }
HFr fsub(const HFr& a, const HFr& b) { return pm::host::sub(a, b, FRF()); }
HFr finv(const HFr& a) { return pm::host::inv(a, FRF()); }
HFr fpow(const HFr& a, u64 e) { return pm::host::pow(a, &e, 1, FRF()); }
HFr fone() { return pm::host::one(FRF()); }
HFr fneg(const HFr& a) { return fsub(pm::host::zero<4>(), a); }
// Montgomery -> canonical limbs
HFr fr_canonical(const HFr& a) {
  HFr raw1 = pm::host::zero<4>();
  raw1.l[0] = 1;
  return fmul(a, raw1);
}

// ------------------------------------------------------------------ Merlin over STROBE-128
struct Strobe128 {
  static const int R = 166;
  uint8_t st[200];
  uint8_t pos = 0, pos_begin = 0, cur_flags = 0;
  explicit Strobe128(const std::string& label) {
    memset(st, 0, sizeof st);
    const uint8_t head[6] = {1, R + 2, 1, 0, 1, 96};
    memcpy(st, head, 6);
    memcpy(st + 6, "STROBEv1.0.2", 12);
    pm_keccak_f1600(st);
    meta_ad((const uint8_t*)label.data(), label.size(), false);
  }
  void run_f() {
    st[pos] ^= pos_begin;
    st[pos + 1] ^= 0x04;
    st[R + 1] ^= 0x80;
    pm_keccak_f1600(st);
    pos = 0;
    pos_begin = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      st[pos++] ^= d[i];
      if (pos == R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      d[i] = st[pos];
      st[pos++] = 0;
      if (pos == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;                       // continued operation (same flags by construction here)
    const uint8_t old_begin = pos_begin;
    pos_begin = (uint8_t)(pos + 1);
    cur_flags = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    if ((flags & (4 | 32)) && pos != 0) run_f();   // C or K
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(16 | 2, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(2, more); absorb(d, n); }
  void prf(uint8_t* d, size_t n) { begin_op(1 | 2 | 4, false); squeeze(d, n); }
};

struct Transcript {
  Strobe128 s;
  explicit Transcript(const std::string& label) : s("Merlin v1.0") { append("dom-sep", (const uint8_t*)label.data(), label.size()); }
  void append(const char* label, const uint8_t* msg, size_t n) {
    s.meta_ad((const uint8_t*)label, strlen(label), false);
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s.meta_ad(len, 4, true);
    s.ad(msg, n, false);
  }
  void append_u64(const char* label, u64 v) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * i));
    append(label, b, 8);
  }
  // 48-byte zcash compressed G1 (big-endian x; bit 7 compressed, bit 6 infinity, bit 5 = y > (p-1)/2)
  void append_commitment(const char* label, const u64 xy[12]) {
    uint8_t out[48];
    memset(out, 0, 48);
    bool any = false;
    for (int i = 0; i < 12; ++i) any = any || xy[i];
    if (!any) {
      out[0] = 0xC0;
    } else {
      HFp x, y, raw1 = pm::host::zero<6>();
      memcpy(x.l, xy, 48);
      memcpy(y.l, xy + 6, 48);
      raw1.l[0] = 1;
      x = pm::host::mul(x, raw1, pm::host::FP());
      y = pm::host::mul(y, raw1, pm::host::FP());
      for (int i = 0; i < 48; ++i) out[i] = (uint8_t)(x.l[(47 - i) / 8] >> (8 * ((47 - i) % 8)));
      out[0] |= 0x80;
      // y > (p - 1) / 2  <=>  2 y > p - 1  <=>  2 y >= p + 1 ... compare y with p - y
      HFp ny = pm::host::sub(pm::host::zero<6>(), y, pm::host::FP());   // canonical limbs: p - y
      if (pm::host::geq<6>(y.l, ny.l) && !pm::host::eq(y, ny)) out[0] |= 0x20;
    }
    append(label, out, 48);
  }
  void append_scalar(const char* label, const HFr& v) {
    const HFr c = fr_canonical(v);
    uint8_t b[32];
    for (int i = 0; i < 32; ++i) b[i] = (uint8_t)(c.l[i / 8] >> (8 * (i % 8)));
    append(label, b, 32);
  }
  // 64 challenge bytes as a little-endian integer mod r (BlsScalar::from_bytes_wide)
  HFr challenge_scalar(const char* label) {
    s.meta_ad((const uint8_t*)label, strlen(label), false);
    uint8_t len[4] = {64, 0, 0, 0};
    s.meta_ad(len, 4, true);
    uint8_t b[64];
    s.prf(b, 64);
    const HFr k256 = fr_u64(256);
    HFr acc = pm::host::zero<4>();
    for (int i = 63; i >= 0; --i) acc = fadd(fmul(acc, k256), fr_u64(b[i]));
    return acc;
  }
};

void put(u64 dst[4], const HFr& v) { memcpy(dst, v.l, 32); }
HFr get(const u64 src[4]) {
  HFr r;
  memcpy(r.l, src, 32);
  return r;
}
char* at(void* base, size_t elems) { return (char*)base + 32 * elems; }

const char* SEL_NAMES[6] = {"q_m", "q_l", "q_r", "q_o", "q_4", "q_c"};
}  // namespace

struct pm_prover_key {
  size_t n = 0;
  uint32_t log_n = 0;
  HFr omega, k[3], zh_inv[4];
  // device arrays (pm_dev_alloc)
  void *roots = nullptr, *x4 = nullptr, *sel_coeffs = nullptr, *sel_coset = nullptr, *sigma_evals = nullptr,
       *sigma_coeffs = nullptr, *sigma_coset = nullptr, *l1_coset = nullptr;
  // per-proof workspace: coeffs [a b c d z pi] 6n | num n | den n | coset 24n | t 4n | r n | agg n | wit 2n | pi0 n
  void *coeffs = nullptr, *num = nullptr, *den = nullptr, *coset = nullptr, *t = nullptr, *r = nullptr, *agg = nullptr,
       *wit = nullptr, *pi_zero = nullptr;
};

#define PK_TRY(call)            \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != PM_OK) return rc_; \
  } while (0)

extern "C" void pm_plonk_key_free(pm_ctx* ctx, pm_prover_key* pk) {
  if (!pk) return;
  for (void* p : {pk->roots, pk->x4, pk->sel_coeffs, pk->sel_coset, pk->sigma_evals, pk->sigma_coeffs, pk->sigma_coset,
                  pk->l1_coset, pk->coeffs, pk->num, pk->den, pk->coset, pk->t, pk->r, pk->agg, pk->wit, pk->pi_zero})
    if (p && ctx) (void)pm_dev_free(ctx, p);
  delete pk;
}

extern "C" int pm_plonk_preprocess(pm_ctx* ctx, const uint64_t* const selectors[6], const int64_t* sigma_index, size_t n,
                                   pm_prover_key** out) {
  if (!ctx || !selectors || !sigma_index || !out) return PM_ERR_BAD_ARG;
  *out = nullptr;
  if (n < 4 || (n & (n - 1))) return PM_ERR_LENGTH;
  pm_prover_key* pk = new pm_prover_key();
  pk->n = n;
  while (((size_t)1 << pk->log_n) < n) ++pk->log_n;
  const uint32_t lg = pk->log_n;
  u64 w[4], wi[4], si[4], w4[4];
  int rc = pm_domain_info(lg, w, wi, si);
  if (!rc) rc = pm_domain_info(lg + 2, w4, wi, wi);
  if (rc) {
    delete pk;
    return rc;
  }
  pk->omega = get(w);
  const HFr n_inv = get(si), omega4 = get(w4), one = fone(), g = fr_u64(7);
  pk->k[0] = fr_u64(7);
  pk->k[1] = fr_u64(13);
  pk->k[2] = fr_u64(17);
  struct Alloc { void** p; size_t elems; };
  const Alloc allocs[] = {{&pk->roots, n},         {&pk->x4, 4 * n},         {&pk->sel_coeffs, 6 * n}, {&pk->sel_coset, 24 * n},
                          {&pk->sigma_evals, 4 * n}, {&pk->sigma_coeffs, 4 * n}, {&pk->sigma_coset, 16 * n}, {&pk->l1_coset, 4 * n},
                          {&pk->coeffs, 6 * n},    {&pk->num, n},            {&pk->den, n},            {&pk->coset, 24 * n},
                          {&pk->t, 4 * n},         {&pk->r, n},              {&pk->agg, n},            {&pk->wit, 2 * n},
                          {&pk->pi_zero, n}};
  for (const Alloc& a : allocs)
    if ((rc = pm_dev_alloc(ctx, a.elems * 32, a.p)) != PM_OK) break;
  void* tmp = nullptr;
  std::vector<u64> table, gathered;
  if (!rc) rc = pm_fr_powers_dev(ctx, pk->omega.l, one.l, n, pk->roots, nullptr);
  if (!rc) rc = pm_fr_powers_dev(ctx, omega4.l, g.l, 4 * n, pk->x4, nullptr);
  // selectors: evaluations -> coefficients -> 4n coset
  if (!rc) rc = pm_dev_alloc(ctx, 6 * n * 32, &tmp);
  for (int s = 0; s < 6 && !rc; ++s) rc = pm_dev_upload(ctx, at(tmp, s * n), selectors[s], n * 32);
  if (!rc) rc = pm_fr_ntt_dev(ctx, tmp, n, n, pk->sel_coeffs, n, lg, 6, PM_NTT_INVERSE, nullptr);
  if (!rc) rc = pm_fr_ntt_dev(ctx, pk->sel_coeffs, n, n, pk->sel_coset, 4 * n, lg + 2, 6, PM_NTT_COSET, nullptr);
  // sigma_j(w^i) = k_j' w^i': gather from the table of the 4n points of the cosets k_j H
  if (!rc) {
    const HFr ks[4] = {one, pk->k[0], pk->k[1], pk->k[2]};
    for (int j = 0; j < 4 && !rc; ++j) rc = pm_fr_powers_dev(ctx, pk->omega.l, ks[j].l, n, at(tmp, j * n), nullptr);
  }
  if (!rc) {
    table.resize(16 * n);
    gathered.resize(16 * n);
    rc = pm_dev_download(ctx, table.data(), tmp, 4 * n * 32);
  }
  if (!rc) {
    std::vector<uint8_t> seen(4 * n, 0);
    for (size_t p = 0; p < 4 * n && !rc; ++p) {
      const int64_t q = sigma_index[p];
      if (q < 0 || (size_t)q >= 4 * n || seen[q]) rc = PM_ERR_BAD_ARG;   // not a permutation
      else {
        seen[q] = 1;
        memcpy(&gathered[4 * p], &table[4 * (size_t)q], 32);
      }
    }
  }
  if (!rc) rc = pm_dev_upload(ctx, pk->sigma_evals, gathered.data(), 4 * n * 32);
  if (!rc) rc = pm_fr_ntt_dev(ctx, pk->sigma_evals, n, n, pk->sigma_coeffs, n, lg, 4, PM_NTT_INVERSE, nullptr);
  if (!rc) rc = pm_fr_ntt_dev(ctx, pk->sigma_coeffs, n, n, pk->sigma_coset, 4 * n, lg + 2, 4, PM_NTT_COSET, nullptr);
  // L_1 = (1/n) sum X^i on the coset
  if (!rc) rc = pm_fr_powers_dev(ctx, one.l, n_inv.l, n, tmp, nullptr);
  if (!rc) rc = pm_fr_ntt_dev(ctx, tmp, n, n, pk->l1_coset, 4 * n, lg + 2, 1, PM_NTT_COSET, nullptr);
  if (!rc) rc = pm_sync(ctx);
  if (tmp) (void)pm_dev_free(ctx, tmp);
  if (!rc) {
    // Z_H(g w4^i) = g^n (w4^n)^i - 1, period 4
    const HFr gn = fpow(g, n), i4 = fpow(omega4, n);
    HFr p = one;
    for (int k = 0; k < 4; ++k) {
      pk->zh_inv[k] = finv(fsub(fmul(gn, p), one));
      p = fmul(p, i4);
    }
  }
  if (rc) {
    pm_plonk_key_free(ctx, pk);
    return rc;
  }
  *out = pk;
  return PM_OK;
}

// Commitments to `batch` coefficient vectors of n elements.  With the SRS split over ranks (shard.fn set)
// this rank's bases cover coefficients [shard.lo, shard.lo + len(ck)): it computes the partial sums of its
// slice and the exchange callback returns the sums over all ranks (all-gather + group-law fold).
struct Shard {
  size_t lo = 0;
  pm_exchange_fn fn = nullptr;
  void* user = nullptr;
};
static int commit_batch(pm_ctx* ctx, const pm_bases* ck, const Shard& sh, const void* d, size_t n, size_t stride,
                        uint32_t batch, u64 (*out_xy)[12]) {
  u64 xyz[4 * 18];
  const size_t have = pm_g1_bases_len(ck);
  size_t cnt = n;
  if (sh.fn) cnt = sh.lo < n ? std::min(have, n - sh.lo) : 0;
  if (cnt > 0) {
    PK_TRY(pm_g1_msm_batch_dev(ctx, ck, 0, cnt, at((void*)d, sh.lo), stride, batch, PM_SCALAR_MONTGOMERY, xyz, nullptr));
  } else {
    memset(xyz, 0, sizeof xyz);
    for (uint32_t b = 0; b < batch; ++b) memcpy(xyz + 18 * b + 6, pm::host::FP().one, 48);   // (0, 1, 0)
  }
  if (sh.fn && sh.fn(sh.user, xyz, batch) != 0) return PM_ERR_BAD_ARG;
  for (uint32_t b = 0; b < batch; ++b) {
    int ident = 0;
    PK_TRY(pm_g1_to_affine(xyz + 18 * b, out_xy[b], &ident));
  }
  return PM_OK;
}

static int prove_impl(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard, const void* d_witness,
                      const void* d_public_inputs, const char* transcript_label, pm_plonk_proof* out);

extern "C" int pm_plonk_prove_sharded(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck_slice, size_t first_coefficient,
                                      const void* d_witness, const void* d_public_inputs, const char* transcript_label,
                                      pm_exchange_fn exchange, void* user, pm_plonk_proof* out) {
  if (!exchange) return PM_ERR_BAD_ARG;
  Shard sh;
  sh.lo = first_coefficient;
  sh.fn = exchange;
  sh.user = user;
  return prove_impl(ctx, pk, ck_slice, sh, d_witness, d_public_inputs, transcript_label, out);
}

extern "C" int pm_plonk_prove(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const void* d_witness,
                              const void* d_public_inputs, const char* transcript_label, pm_plonk_proof* out) {
  return prove_impl(ctx, pk, ck, Shard(), d_witness, d_public_inputs, transcript_label, out);
}

static int prove_impl(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard, const void* d_witness,
                      const void* d_public_inputs, const char* transcript_label, pm_plonk_proof* out) {
  if (!ctx || !pk || !ck || !d_witness || !out) return PM_ERR_BAD_ARG;
  const size_t n = pk->n;
  const uint32_t lg = pk->log_n;
  if (!shard.fn && pm_g1_bases_len(ck) < n) return PM_ERR_LENGTH;
  Transcript ts(transcript_label ? transcript_label : "plonk");
  ts.append("dom-sep", (const uint8_t*)"circuit_size", 12);
  ts.append_u64("n", n);
  const HFr one = fone();
  // ---- round 1 --------------------------------------------------------------------------------
  PK_TRY(pm_fr_ntt_dev(ctx, d_witness, n, n, pk->coeffs, n, lg, 4, PM_NTT_INVERSE, nullptr));
  PK_TRY(commit_batch(ctx, ck, shard, pk->coeffs, n, n, 4, &out->commitments[0]));
  const char* wl[4] = {"w_a", "w_b", "w_c", "w_d"};
  for (int j = 0; j < 4; ++j) ts.append_commitment(wl[j], out->commitments[j]);
  // ---- round 2 --------------------------------------------------------------------------------
  const HFr beta = ts.challenge_scalar("beta"), gamma = ts.challenge_scalar("gamma");
  pm_plonk_perm_args pa;
  memset(&pa, 0, sizeof pa);
  for (int j = 0; j < 4; ++j) {
    pa.wires[j] = at((void*)d_witness, j * n);
    pa.sigmas[j] = at(pk->sigma_evals, j * n);
  }
  pa.roots = pk->roots;
  put(pa.beta, beta);
  put(pa.gamma, gamma);
  for (int j = 0; j < 3; ++j) put(pa.k[j], pk->k[j]);
  PK_TRY(pm_plonk_perm_terms_dev(ctx, &pa, n, pk->num, pk->den, nullptr));
  PK_TRY(pm_fr_batch_inverse_dev(ctx, pk->den, n, nullptr));
  PK_TRY(pm_fr_vec_op_dev(ctx, 2, pk->num, pk->den, n, pk->num, n, nullptr));
  PK_TRY(pm_fr_prefix_product_dev(ctx, pk->num, n, pk->den, nullptr));
  void* z_coeffs = at(pk->coeffs, 4 * n);
  PK_TRY(pm_fr_ntt_dev(ctx, pk->den, n, n, z_coeffs, n, lg, 1, PM_NTT_INVERSE, nullptr));
  PK_TRY(commit_batch(ctx, ck, shard, z_coeffs, n, n, 1, &out->commitments[4]));
  ts.append_commitment("z", out->commitments[4]);
  // ---- round 3 --------------------------------------------------------------------------------
  const HFr alpha = ts.challenge_scalar("alpha");
  void* pi_coeffs = at(pk->coeffs, 5 * n);
  const void* pi_ev = d_public_inputs;
  if (!pi_ev) {
    const HFr zero = pm::host::zero<4>();
    PK_TRY(pm_fr_powers_dev(ctx, zero.l, zero.l, n, pk->pi_zero, nullptr));
    pi_ev = pk->pi_zero;
  }
  PK_TRY(pm_fr_ntt_dev(ctx, pi_ev, n, n, pi_coeffs, n, lg, 1, PM_NTT_INVERSE, nullptr));
  PK_TRY(pm_fr_ntt_dev(ctx, pk->coeffs, n, n, pk->coset, 4 * n, lg + 2, 6, PM_NTT_COSET, nullptr));
  pm_plonk_quotient_args qa;
  memset(&qa, 0, sizeof qa);
  for (int j = 0; j < 4; ++j) {
    qa.wires[j] = at(pk->coset, 4 * n * j);
    qa.sigmas[j] = at(pk->sigma_coset, 4 * n * j);
  }
  qa.z = at(pk->coset, 4 * n * 4);
  qa.pi = at(pk->coset, 4 * n * 5);
  const void** sel[6] = {&qa.q_m, &qa.q_l, &qa.q_r, &qa.q_o, &qa.q_4, &qa.q_c};
  for (int s = 0; s < 6; ++s) *sel[s] = at(pk->sel_coset, 4 * n * s);
  qa.l1 = pk->l1_coset;
  qa.x = pk->x4;
  put(qa.alpha, alpha);
  put(qa.beta, beta);
  put(qa.gamma, gamma);
  for (int j = 0; j < 3; ++j) put(qa.k[j], pk->k[j]);
  for (int j = 0; j < 4; ++j) put(qa.zh_inv[j], pk->zh_inv[j]);
  PK_TRY(pm_plonk_quotient_dev(ctx, &qa, n, pk->t, nullptr));
  PK_TRY(pm_fr_ntt_dev(ctx, pk->t, 4 * n, 4 * n, pk->t, 4 * n, lg + 2, 1, PM_NTT_INVERSE | PM_NTT_COSET, nullptr));
  PK_TRY(commit_batch(ctx, ck, shard, pk->t, n, n, 4, &out->commitments[5]));
  const char* tl[4] = {"t_1", "t_2", "t_3", "t_4"};
  for (int i = 0; i < 4; ++i) ts.append_commitment(tl[i], out->commitments[5 + i]);
  // ---- round 4 --------------------------------------------------------------------------------
  const HFr zc = ts.challenge_scalar("z"), zw = fmul(zc, pk->omega);
  HFr ev[10];   // a b c d sigma_1 sigma_2 sigma_3 z_next t r
  for (int j = 0; j < 4; ++j) PK_TRY(pm_fr_poly_evaluate_dev(ctx, at(pk->coeffs, j * n), n, zc.l, ev[j].l, nullptr));
  for (int j = 0; j < 3; ++j) PK_TRY(pm_fr_poly_evaluate_dev(ctx, at(pk->sigma_coeffs, j * n), n, zc.l, ev[4 + j].l, nullptr));
  PK_TRY(pm_fr_poly_evaluate_dev(ctx, z_coeffs, n, zw.l, ev[7].l, nullptr));
  const HFr zn = fpow(zc, n);
  HFr tp[4];
  for (int i = 0; i < 4; ++i) PK_TRY(pm_fr_poly_evaluate_dev(ctx, at(pk->t, i * n), n, zc.l, tp[i].l, nullptr));
  ev[8] = fadd(tp[0], fmul(zn, fadd(tp[1], fmul(zn, fadd(tp[2], fmul(zn, tp[3]))))));
  const HFr &a_ = ev[0], &b_ = ev[1], &c_ = ev[2], &d_ = ev[3], &s1 = ev[4], &s2 = ev[5], &s3 = ev[6], &z_next = ev[7];
  const HFr l1_z = fmul(fsub(zn, one), finv(fmul(fr_u64(n), fsub(zc, one))));
  const HFr bz = fmul(beta, zc);
  HFr ident = fadd(fadd(a_, bz), gamma);
  const HFr* wv[3] = {&b_, &c_, &d_};
  for (int j = 0; j < 3; ++j) ident = fmul(ident, fadd(fadd(*wv[j], fmul(bz, pk->k[j])), gamma));
  const HFr copy3 = fmul(fmul(fadd(fadd(a_, fmul(beta, s1)), gamma), fadd(fadd(b_, fmul(beta, s2)), gamma)),
                         fadd(fadd(c_, fmul(beta, s3)), gamma));
  const HFr alpha2 = fmul(alpha, alpha);
  const void* lin_v[8];
  u64 lin_c[8][4];
  const HFr lc[8] = {fmul(a_, b_), a_, b_, c_, d_, one, fadd(fmul(alpha, ident), fmul(alpha2, l1_z)),
                     fneg(fmul(fmul(fmul(alpha, copy3), beta), z_next))};
  for (int s = 0; s < 6; ++s) lin_v[s] = at(pk->sel_coeffs, s * n);
  lin_v[6] = z_coeffs;
  lin_v[7] = at(pk->sigma_coeffs, 3 * n);
  for (int i = 0; i < 8; ++i) put(lin_c[i], lc[i]);
  PK_TRY(pm_fr_lincomb_dev(ctx, 8, lin_v, &lin_c[0][0], n, pk->r, nullptr));
  PK_TRY(pm_fr_poly_evaluate_dev(ctx, pk->r, n, zc.l, ev[9].l, nullptr));
  const char* el[10] = {"a_eval", "b_eval", "c_eval", "d_eval", "sigma_1_eval", "sigma_2_eval", "sigma_3_eval",
                        "z_next_eval", "t_eval", "r_eval"};
  for (int i = 0; i < 10; ++i) {
    ts.append_scalar(el[i], ev[i]);
    put(out->evaluations[i], ev[i]);
  }
  // ---- round 5 --------------------------------------------------------------------------------
  const HFr v = ts.challenge_scalar("v");
  const void* agg_v[12];
  u64 agg_c[12][4];
  HFr ac[12];
  ac[0] = one;
  ac[1] = zn;
  ac[2] = fmul(zn, zn);
  ac[3] = fmul(ac[2], zn);
  HFr vp = one;
  for (int e = 0; e < 8; ++e) {
    vp = fmul(vp, v);
    ac[4 + e] = vp;
  }
  for (int i = 0; i < 4; ++i) agg_v[i] = at(pk->t, i * n);
  agg_v[4] = pk->r;
  for (int j = 0; j < 4; ++j) agg_v[5 + j] = at(pk->coeffs, j * n);
  for (int j = 0; j < 3; ++j) agg_v[9 + j] = at(pk->sigma_coeffs, j * n);
  for (int i = 0; i < 12; ++i) put(agg_c[i], ac[i]);
  PK_TRY(pm_fr_lincomb_dev(ctx, 12, agg_v, &agg_c[0][0], n, pk->agg, nullptr));
  PK_TRY(pm_fr_poly_ruffini_dev(ctx, pk->agg, n, zc.l, pk->wit, nullptr));
  PK_TRY(pm_fr_poly_ruffini_dev(ctx, z_coeffs, n, zw.l, at(pk->wit, n), nullptr));
  PK_TRY(commit_batch(ctx, ck, shard, pk->wit, n - 1, n, 2, &out->commitments[9]));
  ts.append_commitment("w_z", out->commitments[9]);
  ts.append_commitment("w_zw", out->commitments[10]);
  const HFr u = ts.challenge_scalar("u");
  const HFr chal[6] = {beta, gamma, alpha, zc, v, u};
  for (int i = 0; i < 6; ++i) put(out->challenges[i], chal[i]);
  (void)SEL_NAMES;
  return PM_OK;
}
