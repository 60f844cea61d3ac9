// Host-only C++: the five PLONK prover rounds behind ONE C-ABI call (pm_plonk_preprocess /
// pm_plonk_prove), so that a Rust (or Go, or C) prover needs no host logic of its own between the
// NTT and MSM kernels -- SURVEY.md section 8f rows N1 + N2 + N3, BASELINE.json configs[3].
//
// Restates dusk_plonk::proof_system::{Prover::preprocess, Prover::prove_with_preprocessed} (dusk-plonk
// 0.8.2, ref:Cargo.toml:19; not in the reference tree) from the published design: 11 selector
// polynomials (arithmetic + q_arith, range, logic, fixed-base, variable-base), the 4-wire
// permutation, a Merlin / STROBE-128 transcript seeded with the verifier key, the 16-evaluation
// Proof, aggregate opening witnesses.  These are the gate kinds the reference's own gadgets emit
// (ref:src/zk/gadgets.rs:34,37,40,88-91,211; ref:src/zk/circuits.rs:64-70).  PARITY UNPINNED:
// formulas and transcript labels are restated, not compared with upstream bytes.  Everything here is
// sequencing and scalar arithmetic on a few dozen field elements: the vector work is done by the
// library's own entry points, called like any client would call them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "context.h"
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
  static void scalar_bytes(uint8_t b[32], const HFr& v) {
    const HFr c = fr_canonical(v);
    for (int i = 0; i < 32; ++i) b[i] = (uint8_t)(c.l[i / 8] >> (8 * (i % 8)));
  }
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
  void append_commitment(const char* label, const u64 xy[12]) {
    uint8_t out[48];
    g1_compress(out, xy);
    append(label, out, 48);
  }
  // 48-byte zcash compressed G1 (big-endian x; bit 7 compressed, bit 6 infinity, bit 5 = y > (p-1)/2)
  static void g1_compress(uint8_t out[48], const u64 xy[12]) {
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

// selector order of dusk's VerifierKey::seed_transcript (and of the ABI)
enum { Q_M, Q_L, Q_R, Q_O, Q_C, Q_4, Q_ARITH, Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR, NSEL };
const int SEL_SEED_ORDER[NSEL] = {Q_M, Q_L, Q_R, Q_O, Q_C, Q_4, Q_ARITH, Q_RANGE, Q_LOGIC, Q_VAR, Q_FIXED};

// ================================================================================================================
// THE transcript table: every label string and the order of every message of a proof's Fiat-Shamir transcript, in
// the order they are absorbed / squeezed.  Restated from the published dusk-plonk 0.8 design (ref:Cargo.toml:19); the
// crate is not in the reference tree and no upstream proof bytes exist here, so these strings are PARITY-UNPINNED:
// byte-equality of a proof with dusk's stands or falls with them.  THIS IS THE SINGLE PLACE TO EDIT when upstream
// vectors become available -- the prover below only refers to this table, and the verifier side in Python
// (plonk-prototype_amd/prover.py: derive_challenges) reads the same table through pm_plonk_transcript_labels().
// ================================================================================================================
namespace tl {
const char* const PROTOCOL = "plonk";                       // default Transcript::new(label)
// VerifierKey::seed_transcript: the 11 selector commitments in SEL_SEED_ORDER (variable before fixed), the 4 sigmas
const char* const SELECTORS[NSEL] = {"q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic",
                                     "q_variable_group_add", "q_fixed_group_add"};
const char* const SIGMAS[4] = {"left_sigma", "right_sigma", "out_sigma", "fourth_sigma"};
// circuit_domain_sep(n)
const char* const DOM_SEP = "dom-sep";
const char* const DOM_SEP_VALUE = "circuit_size";
const char* const CIRCUIT_SIZE = "n";
// this library's public-input binding (flags = 0; absent with PM_PLONK_UPSTREAM_TRANSCRIPT): count, then (position, value)
const char* const PI_LEN = "pi_len";
const char* const PI_POS = "pi_pos";
const char* const PI_VALUE = "pi";
// round 1: the wire commitments
const char* const WIRES[4] = {"w_l", "w_r", "w_o", "w_4"};
// round 2: beta (re-absorbed under its own label), gamma, then the permutation commitment
const char* const BETA = "beta";
const char* const GAMMA = "gamma";
const char* const PERM = "z";
// round 3: the quotient's challenges, then the four quotient commitments
const char* const ALPHA = "alpha";
const char* const RANGE_SEP = "range separation challenge";
const char* const LOGIC_SEP = "logic separation challenge";
const char* const FIXED_SEP = "fixed base separation challenge";
const char* const VAR_SEP = "variable base separation challenge";
const char* const QUOTIENT[4] = {"t_1", "t_2", "t_3", "t_4"};
// round 4: the evaluation challenge and the 17 evaluations in transcript order (pm_plonk_proof.evaluations)
const char* const Z_CHALLENGE = "z";
const char* const EVALS[17] = {"a_eval", "b_eval", "c_eval", "d_eval", "a_next_eval", "b_next_eval", "d_next_eval",
                               "left_sig_eval", "right_sig_eval", "out_sig_eval", "q_arith_eval", "q_c_eval", "q_l_eval",
                               "q_r_eval", "perm_eval", "t_eval", "r_eval"};
// round 5: the two aggregation challenges (same label twice), then the opening commitments; the verifier's batch challenge
const char* const AGGREGATE = "aggregate_witness";
const char* const W_Z = "w_z";
const char* const W_ZW = "w_z_w";
const char* const BATCH = "batch";
}  // namespace tl
const char* const* const SEL_LABELS = tl::SELECTORS;
const char* const* const SIGMA_LABELS = tl::SIGMAS;

// ---- the widgets' linearisation scalars: the same identities as plonk_rounds.hip's quotient kernel,
// on the opening evaluations (widget::*::ProverKey::compute_linearisation)
HFr small(u64 v) { return fr_u64(v); }
HFr wdelta(const HFr& f) {
  return fmul(fmul(fmul(f, fsub(f, small(1))), fsub(f, small(2))), fsub(f, small(3)));
}
HFr edwards_d() { return fneg(fmul(small(10240), finv(small(10241)))); }
struct RowEvals {
  HFr a, b, c, d, an, bn, dn, q_l, q_r, q_c;
};
HFr widget_range(const HFr& sep, const RowEvals& e) {
  const HFr k = fmul(sep, sep), k2 = fmul(k, k), k3 = fmul(k2, k), four = small(4);
  HFr t = wdelta(fsub(e.c, fmul(four, e.d)));
  t = fadd(t, fmul(wdelta(fsub(e.b, fmul(four, e.c))), k));
  t = fadd(t, fmul(wdelta(fsub(e.a, fmul(four, e.b))), k2));
  t = fadd(t, fmul(wdelta(fsub(e.dn, fmul(four, e.a))), k3));
  return fmul(t, sep);
}
HFr widget_logic(const HFr& sep, const RowEvals& e) {
  const HFr k = fmul(sep, sep), k2 = fmul(k, k), k3 = fmul(k2, k), k4 = fmul(k2, k2), four = small(4);
  const HFr qa = fsub(e.an, fmul(four, e.a)), qb = fsub(e.bn, fmul(four, e.b)), qd = fsub(e.dn, fmul(four, e.d));
  const HFr s = fadd(qa, qb), w = e.c;
  HFr in = fadd(fsub(fmul(four, w), fmul(small(18), s)), small(81));
  in = fadd(fmul(w, in), fmul(small(18), fadd(fmul(qa, qa), fmul(qb, qb))));
  in = fadd(fsub(in, fmul(small(81), s)), small(83));
  const HFr f = fmul(w, in);
  const HFr ee = fsub(fmul(small(3), fadd(s, qd)), fadd(f, f));
  const HFr bb = fmul(e.q_c, fsub(fmul(small(9), qd), fmul(small(3), s)));
  HFr t = wdelta(qa);
  t = fadd(t, fmul(wdelta(qb), k));
  t = fadd(t, fmul(wdelta(qd), k2));
  t = fadd(t, fmul(fsub(w, fmul(qa, qb)), k3));
  t = fadd(t, fmul(fadd(bb, ee), k4));
  return fmul(t, sep);
}
HFr widget_fixed(const HFr& sep, const RowEvals& e) {
  const HFr k = fmul(sep, sep), k2 = fmul(k, k), k3 = fmul(k2, k), one = fone();
  const HFr bit = fsub(e.dn, fadd(e.d, e.d));
  HFr t = fmul(fmul(bit, fsub(bit, one)), fadd(bit, one));
  const HFr ya = fadd(fmul(fmul(bit, bit), fsub(e.q_r, one)), one), xa = fmul(e.q_l, bit);
  t = fadd(t, fmul(fsub(fmul(bit, e.q_c), e.c), k));
  const HFr dxy = fmul(fmul(fmul(e.c, e.a), e.b), edwards_d());
  t = fadd(t, fmul(fsub(fadd(e.an, fmul(e.an, dxy)), fadd(fmul(e.a, ya), fmul(e.b, xa))), k2));
  t = fadd(t, fmul(fsub(fsub(e.bn, fmul(e.bn, dxy)), fadd(fmul(e.b, ya), fmul(e.a, xa))), k3));
  return fmul(t, sep);
}
HFr widget_var(const HFr& sep, const RowEvals& e) {
  const HFr k = fmul(sep, sep), k2 = fmul(k, k);
  const HFr y1x2 = fmul(e.b, e.c), y1y2 = fmul(e.b, e.d), x1x2 = fmul(e.a, e.c);
  HFr t = fsub(fmul(e.a, e.d), e.dn);
  const HFr dd = fmul(fmul(e.dn, y1x2), edwards_d());
  t = fadd(t, fmul(fsub(fadd(e.dn, y1x2), fadd(e.an, fmul(e.an, dd))), k));
  t = fadd(t, fmul(fsub(fadd(y1y2, x1x2), fsub(e.bn, fmul(e.bn, dd))), k2));
  return fmul(t, sep);
}
}  // namespace

struct pm_prover_key {
  size_t n = 0;
  uint32_t log_n = 0;
  HFr omega, k[3], zh_inv[4];
  // device arrays (pm_dev_alloc)
  void *roots = nullptr, *x4 = nullptr, *sel_coeffs = nullptr /* 11 n */, *sigma_evals = nullptr,
       *sigma_coeffs = nullptr, *sigma_coset = nullptr, *l1_coset = nullptr;
  void* sel_coset[NSEL] = {};         // 4n each; nullptr where the quotient kernel does not read it
  bool sel_zero[NSEL] = {};           // identically zero
  bool arith_is_one = false;          // q_arith = 1 everywhere: the kernel skips the multiplication
  // per-proof workspace: coeffs [a b c d z pi] 6n | num n | den n | coset 24n | t 4n | r n | agg n | wit 2n | pi n
  void *coeffs = nullptr, *num = nullptr, *den = nullptr, *coset = nullptr, *t = nullptr, *r = nullptr, *agg = nullptr,
       *wit = nullptr, *pi_evals = nullptr;
  // verifier key (commitments to the 11 selector and 4 sigma polynomials) and the transcript seeded with it
  bool committed = false;
  u64 vk[NSEL + 4][12] = {};
  Transcript base{std::string("plonk")};   // overwritten by key_commit (tl::PROTOCOL or the caller's label)
  std::atomic<bool> busy{false};       // a proof is running on this key's workspace
  hipStream_t side = nullptr;         // second stream: work that does not depend on the next challenge
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
};

// side stream <- everything submitted on the context's stream so far / the reverse
static int pm_stream_fork(pm_ctx* ctx, hipStream_t side, hipEvent_t ev) {
  PM_HIP(ctx, hipEventRecord(ev, ctx->stream));
  PM_HIP(ctx, hipStreamWaitEvent(side, ev, 0));
  return PM_OK;
}
static int pm_stream_join(pm_ctx* ctx, hipStream_t side, hipEvent_t ev) {
  PM_HIP(ctx, hipEventRecord(ev, side));
  PM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
  return PM_OK;
}

#define PK_TRY(call)            \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != PM_OK) return rc_; \
  } while (0)

extern "C" void pm_plonk_key_free(pm_ctx* ctx, pm_prover_key* pk) {
  if (!pk) return;
  if (ctx) (void)pm_sync(ctx);
  if (pk->side) {
    (void)hipStreamSynchronize(pk->side);
    (void)hipStreamDestroy(pk->side);
  }
  if (pk->ev_main) (void)hipEventDestroy(pk->ev_main);
  if (pk->ev_side) (void)hipEventDestroy(pk->ev_side);
  for (void* p : {pk->roots, pk->x4, pk->sel_coeffs, pk->sigma_evals, pk->sigma_coeffs, pk->sigma_coset,
                  pk->l1_coset, pk->coeffs, pk->num, pk->den, pk->coset, pk->t, pk->r, pk->agg, pk->wit, pk->pi_evals})
    if (p && ctx) (void)pm_dev_free(ctx, p);
  for (void* p : pk->sel_coset)
    if (p && ctx) (void)pm_dev_free(ctx, p);
  delete pk;
}

extern "C" int pm_plonk_preprocess(pm_ctx* ctx, const uint64_t* const selectors[PM_PLONK_SELECTORS],
                                   const int64_t* sigma_index, size_t n, pm_prover_key** out) {
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
  // which selector polynomials are trivial: the quotient kernel skips what it can
  pk->arith_is_one = selectors[Q_ARITH] != nullptr;
  for (int s = 0; s < NSEL; ++s) {
    bool zero = true;
    if (selectors[s]) {
      for (size_t i = 0; i < 4 * n && zero; ++i) zero = selectors[s][i] == 0;
    }
    pk->sel_zero[s] = zero;
  }
  if (selectors[Q_ARITH])
    for (size_t i = 0; i < n && pk->arith_is_one; ++i) pk->arith_is_one = memcmp(selectors[Q_ARITH] + 4 * i, one.l, 32) == 0;
  struct Alloc { void** p; size_t elems; };
  std::vector<Alloc> allocs = {{&pk->roots, n},          {&pk->x4, 4 * n},           {&pk->sel_coeffs, (size_t)NSEL * n},
                               {&pk->sigma_evals, 4 * n}, {&pk->sigma_coeffs, 4 * n}, {&pk->sigma_coset, 16 * n},
                               {&pk->l1_coset, 4 * n},    {&pk->coeffs, 6 * n},       {&pk->num, n},
                               {&pk->den, n},             {&pk->coset, 24 * n},       {&pk->t, 4 * n},
                               {&pk->r, n},               {&pk->agg, n},              {&pk->wit, 2 * n},
                               {&pk->pi_evals, n}};
  for (int s = 0; s < NSEL; ++s) {
    const bool need = s <= Q_4 || (s == Q_ARITH ? !pk->arith_is_one : !pk->sel_zero[s]);
    if (need) allocs.push_back({&pk->sel_coset[s], 4 * n});
  }
  for (const Alloc& a : allocs)
    if ((rc = pm_dev_alloc(ctx, a.elems * 32, a.p)) != PM_OK) break;
  void* tmp = nullptr;
  if (!rc) rc = pm_fr_powers_dev(ctx, pk->omega.l, one.l, n, pk->roots, nullptr);
  if (!rc) rc = pm_fr_powers_dev(ctx, omega4.l, g.l, 4 * n, pk->x4, nullptr);
  // selectors: evaluations -> coefficients -> 4n coset (only the coset forms the quotient kernel reads)
  if (!rc) rc = pm_dev_alloc(ctx, (size_t)NSEL * n * 32, &tmp);
  const HFr zero = pm::host::zero<4>();
  for (int s = 0; s < NSEL && !rc; ++s) {
    if (pk->sel_zero[s]) rc = pm_fr_powers_dev(ctx, zero.l, zero.l, n, at(tmp, s * n), nullptr);
    else rc = pm_dev_upload(ctx, at(tmp, s * n), selectors[s], n * 32);
  }
  if (!rc) rc = pm_fr_ntt_dev(ctx, tmp, n, n, pk->sel_coeffs, n, lg, NSEL, PM_NTT_INVERSE, nullptr);
  for (int s = 0; s < NSEL && !rc; ++s)
    if (pk->sel_coset[s])
      rc = pm_fr_ntt_dev(ctx, at(pk->sel_coeffs, s * n), n, n, pk->sel_coset[s], 4 * n, lg + 2, 1, PM_NTT_COSET, nullptr);
  // sigma_j(w^i) = k_j' w^i': the indices are checked here (a permutation of the 4n wire positions), the values gathered on
  // the device (r01 - r04: a 4n x 32-byte table went to the host and back)
  if (!rc) {
    std::vector<uint8_t> seen(4 * n, 0);
    for (size_t p = 0; p < 4 * n && !rc; ++p) {
      const int64_t q = sigma_index[p];
      if (q < 0 || (size_t)q >= 4 * n || seen[q]) rc = PM_ERR_BAD_ARG;   // not a permutation
      else seen[q] = 1;
    }
  }
  if (!rc) {
    u64 kk[3][4];
    for (int j = 0; j < 3; ++j) put(kk[j], pk->k[j]);
    rc = pm::sigma_evals_from_index(ctx, sigma_index, 4 * n, lg, pk->omega.l, kk, pk->sigma_evals);
  }
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
  if (!rc && hipStreamCreateWithFlags(&pk->side, hipStreamNonBlocking) != hipSuccess) rc = PM_ERR_HIP;
  if (!rc && hipEventCreateWithFlags(&pk->ev_main, hipEventDisableTiming) != hipSuccess) rc = PM_ERR_HIP;
  if (!rc && hipEventCreateWithFlags(&pk->ev_side, hipEventDisableTiming) != hipSuccess) rc = PM_ERR_HIP;
  if (rc) {
    pm_plonk_key_free(ctx, pk);
    return rc;
  }
  *out = pk;
  return PM_OK;
}

// Commitments to `batch` coefficient vectors of n elements.  With the SRS split over ranks (shard.fn set)
// this rank's bases cover coefficients [shard.lo, shard.lo + len(ck)): it computes the partial sums of its
// slice and the exchange callback returns the sums over all ranks (all-gather + group-law fold).  A rank
// whose local MSM fails still enters the exchange (with an abort marker) so that its peers do not block.
struct Shard {
  size_t lo = 0;
  bool on = false;             // the SRS is split: partial sums are exchanged
  pm_exchange_fn fn = nullptr; // nullptr with on = true: the context's RCCL communicator (pm_g1_allgather_fold)
  void* user = nullptr;
  mutable bool aborted = false;   // an exchange of this call already carried / returned the abort marker
  // How many exchanges the call makes on every rank (key commit: slice cover + two batches; proof: four commit batches)
  // and how many this rank has completed: an error AFTER the last one must not put one more collective on the wire
  // -- no peer would be waiting for it, and it would pair with the first exchange of the peers' next call.
  int expect = 0;
  mutable int done = 0;
};
// One exchange of k partial points (k = 0: the abort marker).
static int shard_exchange(pm_ctx* ctx, const Shard& sh, u64* xyz, uint32_t k) {
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ++ctx->stat_allgather_calls;
  }
  const int rc = sh.fn ? (sh.fn(sh.user, xyz, k) != 0 ? PM_ERR_EXCHANGE : PM_OK) : pm_g1_allgather_fold(ctx, xyz, k);
  ++sh.done;
  if (k == 0 || rc != PM_OK) sh.aborted = true;
  return k == 0 && rc == PM_OK ? PM_ERR_EXCHANGE : rc;
}
// A sharded call that fails anywhere OUTSIDE an exchange -- an NTT, the quotient, an allocation, an argument that is
// wrong on this rank only -- must still meet its peers in their next collective, or they block in it for ever
// (ncclAllGather has no timeout): one abort marker, after which every rank returns an error.
static int shard_leave(pm_ctx* ctx, const Shard& sh, int rc) {
  if (rc != PM_OK && sh.on && ctx && !sh.aborted && sh.done < sh.expect) {
    u64 xyz[18] = {0};
    (void)shard_exchange(ctx, sh, xyz, 0);
  }
  return rc;
}
static int commit_batch(pm_ctx* ctx, const pm_bases* ck, const Shard& sh, const void* d, size_t n, size_t stride,
                        uint32_t batch, u64 (*out_xy)[12]) {
  u64 xyz[16 * 18];
  if (batch > 16) return PM_ERR_BAD_ARG;
  const size_t have = pm_g1_bases_len(ck);
  size_t cnt = n;
  if (sh.on) cnt = sh.lo < n ? std::min(have, n - sh.lo) : 0;
  int rc = PM_OK;
  if (cnt > 0) {
    rc = pm_g1_msm_batch_dev(ctx, ck, 0, cnt, at((void*)d, sh.lo), stride, batch, PM_SCALAR_MONTGOMERY, xyz, nullptr);
  } else {
    memset(xyz, 0, sizeof xyz);
    for (uint32_t b = 0; b < batch; ++b) memcpy(xyz + 18 * b + 6, pm::host::FP().one, 48);   // (0, 1, 0)
  }
  if (sh.on) {
    // k = 0 tells the peers that this rank gave up: they get an error back and stop too
    const int xrc = shard_exchange(ctx, sh, xyz, rc == PM_OK ? batch : 0);
    if (rc == PM_OK && xrc != PM_OK) rc = xrc;
  }
  if (rc) return rc;
  return pm_g1_to_affine_batch(xyz, batch, &out_xy[0][0], nullptr);   // one host inversion per batch
}

// The ranks' SRS slices must tile [0, n): a gap or an overlap would give a well-formed but wrong commitment that no
// rank can see.  One extra exchange per key: every rank contributes len * G and S_rank * G with
//     S_rank = sum over its coefficient indices i of rho^i      (G the group generator, host-side scalar multiplications),
// rho a fixed element of Fr of no special structure; the folded sums must be n * G and (rho^n - 1) / (rho - 1) * G.
// sum_ranks S_rank is the polynomial  sum_i m_i X^i  at rho, m_i = how many ranks hold coefficient i: it equals
// 1 + X + ... + X^(n-1) as a polynomial exactly when every m_i is 1, and two different polynomials of degree < 2^32
// agree at a point that was fixed without looking at them with probability < 2^-220 -- this is a guard against
// mis-configured ranks, not against an adversary.  (r03 compared two moments, count and index sum: slices shifted
// symmetrically about a centre passed both)
static const u64 COVER_RHO[4] = {0x9e3779b97f4a7c15ULL, 0xbf58476d1ce4e5b9ULL, 0x94d049bb133111ebULL, 0x2545f4914f6cdd1dULL};
static HFr cover_rho() {                      // the 254-bit constant above (< r), into Montgomery form
  HFr raw, r2;
  memcpy(raw.l, COVER_RHO, 32);
  memcpy(r2.l, FRF().r2, 32);
  return fmul(raw, r2);
}
// sum_{i = lo}^{lo + cnt - 1} rho^i = rho^lo (rho^cnt - 1) / (rho - 1)
static HFr cover_sum(u64 lo, u64 cnt) {
  const HFr rho = cover_rho();
  return fmul(fmul(fpow(rho, lo), fsub(fpow(rho, cnt), fone())), finv(fsub(rho, fone())));
}
static pm::host::XYZZ host_mul_generator(const u64 k[4]) {
  using namespace pm::host;
  // the BLS12-381 G1 generator, canonical limbs -> Montgomery
  static const u64 GX[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                            0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
  static const u64 GY[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                            0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
  const Field<6>& F = FP();
  HFp r2, x, y;
  memcpy(r2.l, F.r2, 48);
  memcpy(x.l, GX, 48);
  memcpy(y.l, GY, 48);
  XYZZ g;
  g.x = mul(x, r2, F);
  g.y = mul(y, r2, F);
  g.zz = one(F);
  g.zzz = one(F);
  XYZZ acc = xyzz_identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = xyzz_double(acc);
    if ((k[bit >> 6] >> (bit & 63)) & 1) acc = xyzz_add(acc, g);
  }
  return acc;
}
static pm::host::XYZZ host_mul_generator(u64 k) {
  const u64 kk[4] = {k, 0, 0, 0};
  return host_mul_generator(kk);
}
static pm::host::XYZZ host_mul_generator(const HFr& k) { return host_mul_generator(fr_canonical(k).l); }
static void host_to_projective(const pm::host::XYZZ& p, u64 out[18]) {
  pm::host::HFp x, y;
  memset(out, 0, 144);
  if (!pm::host::xyzz_to_affine(p, x, y)) {
    memcpy(out + 6, pm::host::FP().one, 48);   // (0, 1, 0)
    return;
  }
  memcpy(out, x.l, 48);
  memcpy(out + 6, y.l, 48);
  memcpy(out + 12, pm::host::FP().one, 48);
}
static int check_slice_cover(pm_ctx* ctx, const pm_bases* ck, const Shard& sh, size_t n) {
  const size_t have = pm_g1_bases_len(ck);
  const u64 cnt = sh.lo < n ? std::min<size_t>(have, n - sh.lo) : 0;
  u64 xyz[2 * 18], want[2 * 18], got_xy[2 * 12], want_xy[2 * 12];
  host_to_projective(host_mul_generator(cnt), xyz);
  host_to_projective(host_mul_generator(cover_sum(sh.lo, cnt)), xyz + 18);
  host_to_projective(host_mul_generator((u64)n), want);
  host_to_projective(host_mul_generator(cover_sum(0, n)), want + 18);
  PK_TRY(shard_exchange(ctx, sh, xyz, 2));
  PK_TRY(pm_g1_to_affine_batch(xyz, 2, got_xy, nullptr));
  PK_TRY(pm_g1_to_affine_batch(want, 2, want_xy, nullptr));
  if (memcmp(got_xy, want_xy, sizeof got_xy) != 0) {
    ctx->err = "the ranks' commit-key slices do not tile [0, n): a gap or an overlap";
    return PM_ERR_LENGTH;
  }
  return PM_OK;
}

static int key_commit_body(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard,
                           const char* transcript_label, uint64_t (*vk_out)[12]);
static int key_commit_impl(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard,
                           const char* transcript_label, uint64_t (*vk_out)[12]) {
  return shard_leave(ctx, shard, key_commit_body(ctx, pk, ck, shard, transcript_label, vk_out));
}
static int key_commit_body(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard,
                           const char* transcript_label, uint64_t (*vk_out)[12]) {
  if (!ctx || !pk || !ck) return PM_ERR_BAD_ARG;
  const size_t n = pk->n;
  if (!shard.on && pm_g1_bases_len(ck) < n) return PM_ERR_LENGTH;
  if (shard.on) PK_TRY(check_slice_cover(ctx, ck, shard, n));
  // the verifier key: commitments to the 11 selector and the 4 sigma polynomials
  PK_TRY(commit_batch(ctx, ck, shard, pk->sel_coeffs, n, n, NSEL, &pk->vk[0]));
  PK_TRY(commit_batch(ctx, ck, shard, pk->sigma_coeffs, n, n, 4, &pk->vk[NSEL]));
  // Prover::preprocess: Transcript::new(label), VerifierKey::seed_transcript, circuit_domain_sep(n)
  Transcript ts(transcript_label ? transcript_label : tl::PROTOCOL);
  for (int i = 0; i < NSEL; ++i) ts.append_commitment(SEL_LABELS[i], pk->vk[SEL_SEED_ORDER[i]]);
  for (int j = 0; j < 4; ++j) ts.append_commitment(SIGMA_LABELS[j], pk->vk[NSEL + j]);
  ts.append(tl::DOM_SEP, (const uint8_t*)tl::DOM_SEP_VALUE, strlen(tl::DOM_SEP_VALUE));
  ts.append_u64(tl::CIRCUIT_SIZE, n);
  pk->base = ts;
  pk->committed = true;
  if (vk_out) memcpy(vk_out, pk->vk, sizeof pk->vk);
  return PM_OK;
}

// The table above as text, one "key=label" line per entry in transcript order (the verifier side in Python reads it:
// one table for both sides).
extern "C" const char* pm_plonk_transcript_labels(void) {
  static const std::string text = [] {
    std::string t;
    auto add = [&](const std::string& k, const char* v) { t += k + "=" + v + "\n"; };
    add("protocol", tl::PROTOCOL);
    for (int i = 0; i < NSEL; ++i) add("selector_" + std::to_string(i), tl::SELECTORS[i]);
    for (int i = 0; i < 4; ++i) add("sigma_" + std::to_string(i), tl::SIGMAS[i]);
    add("dom_sep", tl::DOM_SEP);
    add("dom_sep_value", tl::DOM_SEP_VALUE);
    add("circuit_size", tl::CIRCUIT_SIZE);
    add("pi_len", tl::PI_LEN);
    add("pi_pos", tl::PI_POS);
    add("pi_value", tl::PI_VALUE);
    for (int i = 0; i < 4; ++i) add("wire_" + std::to_string(i), tl::WIRES[i]);
    add("beta", tl::BETA);
    add("gamma", tl::GAMMA);
    add("perm", tl::PERM);
    add("alpha", tl::ALPHA);
    add("range_sep", tl::RANGE_SEP);
    add("logic_sep", tl::LOGIC_SEP);
    add("fixed_sep", tl::FIXED_SEP);
    add("var_sep", tl::VAR_SEP);
    for (int i = 0; i < 4; ++i) add("quotient_" + std::to_string(i), tl::QUOTIENT[i]);
    add("z_challenge", tl::Z_CHALLENGE);
    for (int i = 0; i < 17; ++i) add("eval_" + std::to_string(i), tl::EVALS[i]);
    add("aggregate", tl::AGGREGATE);
    add("w_z", tl::W_Z);
    add("w_zw", tl::W_ZW);
    add("batch", tl::BATCH);
    return t;
  }();
  return text.c_str();
}

extern "C" int pm_plonk_key_commit(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key, const char* transcript_label,
                                   uint64_t (*verifier_key_out)[12]) {
  return key_commit_impl(ctx, key, commit_key, Shard(), transcript_label, verifier_key_out);
}
extern "C" int pm_plonk_key_commit_sharded(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key_slice,
                                           size_t first_coefficient, pm_exchange_fn exchange, void* user,
                                           const char* transcript_label, uint64_t (*verifier_key_out)[12]) {
  Shard sh;
  sh.on = true;
  sh.lo = first_coefficient;
  sh.fn = exchange;
  sh.user = user;
  sh.expect = 3;
  return key_commit_impl(ctx, key, commit_key_slice, sh, transcript_label, verifier_key_out);
}

// pi_evals <- 0, then the sparse public inputs (a repeated position keeps its last value).  A handful goes up
// element by element; longer lists are staged as compact (position, value) arrays in the round-2 scratch
// (num / den are not written before round 2, which is ordered after this on the same stream) and scattered by
// one kernel, so a statement with thousands of public inputs costs two copies, not thousands.
__global__ void pi_scatter_kernel(const uint4* vals, const unsigned long long* pos, size_t cnt, uint4* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  out[2 * pos[i]] = vals[2 * i];
  out[2 * pos[i] + 1] = vals[2 * i + 1];
}
static int scatter_public_inputs(pm_ctx* ctx, pm_prover_key* pk, const uint64_t* pos, const uint64_t* vals, size_t n_pi) {
  PM_HIP(ctx, hipSetDevice(ctx->device));
  PM_HIP(ctx, hipMemsetAsync(pk->pi_evals, 0, pk->n * 32, ctx->stream));
  if (n_pi <= 16) {
    for (size_t i = 0; i < n_pi; ++i)
      PM_HIP(ctx, hipMemcpyAsync(at(pk->pi_evals, pos[i]), vals + 4 * i, 32, hipMemcpyHostToDevice, ctx->stream));
    return PM_OK;
  }
  std::unordered_map<uint64_t, size_t> last;
  last.reserve(2 * n_pi);
  for (size_t i = 0; i < n_pi; ++i) last[pos[i]] = i;
  std::vector<unsigned long long> hp;
  std::vector<uint64_t> hv;
  hp.reserve(last.size());
  hv.reserve(4 * last.size());
  for (size_t i = 0; i < n_pi; ++i) {
    if (last[pos[i]] != i) continue;
    hp.push_back(pos[i]);
    hv.insert(hv.end(), vals + 4 * i, vals + 4 * i + 4);
  }
  const size_t cnt = hp.size();   // <= n: fits the n-element scratch arrays
  PM_HIP(ctx, hipMemcpyAsync(pk->num, hv.data(), cnt * 32, hipMemcpyHostToDevice, ctx->stream));
  PM_HIP(ctx, hipMemcpyAsync(pk->den, hp.data(), cnt * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(pi_scatter_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint4*)pk->num, (const unsigned long long*)pk->den, cnt, (uint4*)pk->pi_evals);
  PM_HIP(ctx, hipGetLastError());
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging vectors die with this scope
  return PM_OK;
}

static int prove_body(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard, const void* d_witness,
                      const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                      pm_plonk_proof* out);
static int prove_impl(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard, const void* d_witness,
                      const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                      pm_plonk_proof* out) {
  if (ctx && ctx->marks_on) ctx->marks.clear();
  pm::host_mark(ctx, "prove: start");
  const int rc = shard_leave(ctx, shard, prove_body(ctx, pk, ck, shard, d_witness, pi_positions, pi_values, n_pi, flags, out));
  pm::host_mark(ctx, "prove: end");
  if (ctx && ctx->marks_on && !ctx->marks.empty()) {   // PM_HOST_MARKS=1: where the host's time between the kernels goes
    const double t0 = ctx->marks.front().second;
    double prev = t0;
    for (const auto& mk : ctx->marks) {
      fprintf(stderr, "[host] %9.1f us  +%7.1f  %s\n", mk.second - t0, mk.second - prev, mk.first);
      prev = mk.second;
    }
  }
  return rc;
}

extern "C" int pm_plonk_prove_sharded(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck_slice, size_t first_coefficient,
                                      const void* d_witness, const uint64_t* pi_positions, const uint64_t* pi_values,
                                      size_t n_pi, uint32_t flags, pm_exchange_fn exchange, void* user,
                                      pm_plonk_proof* out) {
  Shard sh;
  sh.on = true;
  sh.lo = first_coefficient;
  sh.fn = exchange;
  sh.user = user;
  sh.expect = 4;
  return prove_impl(ctx, pk, ck_slice, sh, d_witness, pi_positions, pi_values, n_pi, flags, out);
}

extern "C" int pm_plonk_prove(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const void* d_witness,
                              const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                              pm_plonk_proof* out) {
  return prove_impl(ctx, pk, ck, Shard(), d_witness, pi_positions, pi_values, n_pi, flags, out);
}

namespace {
struct BusyGuard {   // one proof at a time per key: the key owns the per-proof workspace
  pm_prover_key* pk;
  bool ok;
  explicit BusyGuard(pm_prover_key* k) : pk(k), ok(!k->busy.exchange(true)) {}
  ~BusyGuard() {
    if (ok) pk->busy.store(false);
  }
};
}  // namespace

static int prove_body(pm_ctx* ctx, pm_prover_key* pk, const pm_bases* ck, const Shard& shard, const void* d_witness,
                      const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                      pm_plonk_proof* out) {
  if (!ctx || !pk || !ck || !d_witness || !out) return PM_ERR_BAD_ARG;
  if (n_pi && (!pi_positions || !pi_values)) return PM_ERR_BAD_ARG;
  if (flags & ~(PM_PLONK_BIND_PUBLIC_INPUTS | PM_PLONK_UPSTREAM_TRANSCRIPT)) return PM_ERR_BAD_ARG;
  if (flags == (PM_PLONK_BIND_PUBLIC_INPUTS | PM_PLONK_UPSTREAM_TRANSCRIPT)) return PM_ERR_BAD_ARG;   // the two modes exclude each other
  if (!pk->committed) return PM_ERR_BAD_ARG;   // pm_plonk_key_commit first: the transcript starts from the verifier key
  BusyGuard guard(pk);
  if (!guard.ok) return PM_ERR_BUSY;
  const size_t n = pk->n;
  const uint32_t lg = pk->log_n;
  if (!shard.on && pm_g1_bases_len(ck) < n) return PM_ERR_LENGTH;
  for (size_t i = 0; i < n_pi; ++i)
    if (pi_positions[i] >= n) return PM_ERR_LENGTH;
  Transcript ts = pk->base;
  if (!(flags & PM_PLONK_UPSTREAM_TRANSCRIPT)) {
    // the default; not in dusk-plonk 0.8.2 (its transcript never sees the public inputs): binds the statement to
    // the challenges so that it cannot be chosen after them
    ts.append_u64(tl::PI_LEN, n_pi);
    for (size_t i = 0; i < n_pi; ++i) {
      ts.append_u64(tl::PI_POS, pi_positions[i]);
      ts.append_scalar(tl::PI_VALUE, get(pi_values + 4 * i));
    }
  }
  const HFr one = fone();
  // ---- round 1 --------------------------------------------------------------------------------
  pm::host_mark(ctx, "round 1");
  PK_TRY(pm_fr_ntt_dev(ctx, d_witness, n, n, pk->coeffs, n, lg, 4, PM_NTT_INVERSE, nullptr));
  // work no challenge depends on goes to the side stream and runs under the MSMs of rounds 1 and 2:
  // the public-input polynomial and the wire polynomials on the 4n coset (round 3 reads them)
  hipStream_t side = pk->side;
  void* pi_coeffs = at(pk->coeffs, 5 * n);
  {
    PK_TRY(scatter_public_inputs(ctx, pk, pi_positions, pi_values, n_pi));
    PK_TRY(pm_fr_ntt_dev(ctx, pk->pi_evals, n, n, pi_coeffs, n, lg, 1, PM_NTT_INVERSE, nullptr));
    PK_TRY(pm_stream_fork(ctx, side, pk->ev_main));
    PK_TRY(pm_fr_ntt_dev(ctx, pk->coeffs, n, n, pk->coset, 4 * n, lg + 2, 4, PM_NTT_COSET, side));
    PK_TRY(pm_fr_ntt_dev(ctx, pi_coeffs, n, n, at(pk->coset, 4 * n * 5), 4 * n, lg + 2, 1, PM_NTT_COSET, side));
  }
  PK_TRY(commit_batch(ctx, ck, shard, pk->coeffs, n, n, 4, &out->commitments[0]));
  for (int j = 0; j < 4; ++j) ts.append_commitment(tl::WIRES[j], out->commitments[j]);
  // ---- round 2 --------------------------------------------------------------------------------
  pm::host_mark(ctx, "round 2");
  const HFr beta = ts.challenge_scalar(tl::BETA);
  ts.append_scalar(tl::BETA, beta);
  const HFr gamma = ts.challenge_scalar(tl::GAMMA);
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
  PK_TRY(pm::fr_batch_inverse_mul(ctx, pk->den, pk->num, n, nullptr));       // den <- num / den (r06: one kernel, was two)
  PK_TRY(pm_fr_prefix_product_dev(ctx, pk->den, n, pk->num, nullptr));       // num <- z on H
  void* z_coeffs = at(pk->coeffs, 4 * n);
  PK_TRY(pm_fr_ntt_dev(ctx, pk->num, n, n, z_coeffs, n, lg, 1, PM_NTT_INVERSE, nullptr));
  // z on the 4n coset depends on no further challenge: side stream, under the commitment to z
  PK_TRY(pm_stream_fork(ctx, side, pk->ev_main));
  PK_TRY(pm_fr_ntt_dev(ctx, z_coeffs, n, n, at(pk->coset, 4 * n * 4), 4 * n, lg + 2, 1, PM_NTT_COSET, side));
  PK_TRY(commit_batch(ctx, ck, shard, z_coeffs, n, n, 1, &out->commitments[4]));
  ts.append_commitment(tl::PERM, out->commitments[4]);
  // ---- round 3 --------------------------------------------------------------------------------
  pm::host_mark(ctx, "round 3");
  const HFr alpha = ts.challenge_scalar(tl::ALPHA);
  const HFr range_sep = ts.challenge_scalar(tl::RANGE_SEP);
  const HFr logic_sep = ts.challenge_scalar(tl::LOGIC_SEP);
  const HFr fixed_sep = ts.challenge_scalar(tl::FIXED_SEP);
  const HFr var_sep = ts.challenge_scalar(tl::VAR_SEP);
  PK_TRY(pm_stream_join(ctx, side, pk->ev_side));   // the wire, PI and z coset forms are ready
  pm_plonk_quotient_args qa;
  memset(&qa, 0, sizeof qa);
  for (int j = 0; j < 4; ++j) {
    qa.wires[j] = at(pk->coset, 4 * n * j);
    qa.sigmas[j] = at(pk->sigma_coset, 4 * n * j);
  }
  qa.z = at(pk->coset, 4 * n * 4);
  qa.pi = at(pk->coset, 4 * n * 5);
  qa.q_m = pk->sel_coset[Q_M];
  qa.q_l = pk->sel_coset[Q_L];
  qa.q_r = pk->sel_coset[Q_R];
  qa.q_o = pk->sel_coset[Q_O];
  qa.q_c = pk->sel_coset[Q_C];
  qa.q_4 = pk->sel_coset[Q_4];
  qa.q_arith = pk->sel_coset[Q_ARITH];            // nullptr when q_arith = 1
  qa.q_range = pk->sel_coset[Q_RANGE];
  qa.q_logic = pk->sel_coset[Q_LOGIC];
  qa.q_fixed_group_add = pk->sel_coset[Q_FIXED];
  qa.q_variable_group_add = pk->sel_coset[Q_VAR];
  qa.l1 = pk->l1_coset;
  qa.x = pk->x4;
  put(qa.alpha, alpha);
  put(qa.beta, beta);
  put(qa.gamma, gamma);
  put(qa.range_sep, range_sep);
  put(qa.logic_sep, logic_sep);
  put(qa.fixed_sep, fixed_sep);
  put(qa.var_sep, var_sep);
  for (int j = 0; j < 3; ++j) put(qa.k[j], pk->k[j]);
  for (int j = 0; j < 4; ++j) put(qa.zh_inv[j], pk->zh_inv[j]);
  PK_TRY(pm_plonk_quotient_dev(ctx, &qa, n, pk->t, nullptr));
  PK_TRY(pm_fr_ntt_dev(ctx, pk->t, 4 * n, 4 * n, pk->t, 4 * n, lg + 2, 1, PM_NTT_INVERSE | PM_NTT_COSET, nullptr));
  PK_TRY(commit_batch(ctx, ck, shard, pk->t, n, n, 4, &out->commitments[5]));
  for (int i = 0; i < 4; ++i) ts.append_commitment(tl::QUOTIENT[i], out->commitments[5 + i]);
  // ---- round 4 --------------------------------------------------------------------------------
  pm::host_mark(ctx, "round 4");
  const HFr zc = ts.challenge_scalar(tl::Z_CHALLENGE), zw = fmul(zc, pk->omega);
  enum { E_A, E_B, E_C, E_D, E_AN, E_BN, E_DN, E_S1, E_S2, E_S3, E_QARITH, E_QC, E_QL, E_QR, E_ZN, E_T, E_R, NEV };
  HFr ev[NEV];
  // r(z) needs no pass over r: r is a linear combination of key and round polynomials, so r(z) is the same combination of
  // their values at z -- the ones the proof does not open anyway (q_m, q_o, q_4, z, sigma_4, the widget selectors) ride along
  // as a second group.  ONE host synchronisation for all openings (r01 - r04: three, with r's own evaluation behind the
  // linear combination).
  enum { X_QM, X_QO, X_Q4, X_Z, X_S4, X_RANGE, X_LOGIC, X_FIXED, X_VAR, NX };
  HFr xv[NX];
  {
    const void* at_z[15];
    const void* at_x[NX];
    u64 out_z[15][4], out_x[NX][4], out_zw[4][4];
    for (int j = 0; j < 4; ++j) at_z[j] = at(pk->coeffs, j * n);
    for (int j = 0; j < 3; ++j) at_z[4 + j] = at(pk->sigma_coeffs, j * n);
    at_z[7] = at(pk->sel_coeffs, Q_ARITH * n);
    at_z[8] = at(pk->sel_coeffs, Q_C * n);
    at_z[9] = at(pk->sel_coeffs, Q_L * n);
    at_z[10] = at(pk->sel_coeffs, Q_R * n);
    for (int i = 0; i < 4; ++i) at_z[11 + i] = at(pk->t, i * n);
    at_x[X_QM] = at(pk->sel_coeffs, Q_M * n);
    at_x[X_QO] = at(pk->sel_coeffs, Q_O * n);
    at_x[X_Q4] = at(pk->sel_coeffs, Q_4 * n);
    at_x[X_Z] = z_coeffs;
    at_x[X_S4] = at(pk->sigma_coeffs, 3 * n);
    uint32_t nx = X_RANGE;
    const int wsel[4] = {Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR};
    int xslot[4] = {-1, -1, -1, -1};
    for (int w = 0; w < 4; ++w)
      if (!pk->sel_zero[wsel[w]]) {
        xslot[w] = (int)nx;
        at_x[nx++] = at(pk->sel_coeffs, wsel[w] * n);
      }
    const void* at_zw[4] = {at(pk->coeffs, 0), at(pk->coeffs, n), at(pk->coeffs, 3 * n), z_coeffs};
    const uint32_t gk[3] = {15, nx, 4};
    const void* const* gp[3] = {at_z, at_x, at_zw};
    const uint64_t* gpt[3] = {zc.l, zc.l, zw.l};
    uint64_t* gout[3] = {&out_z[0][0], &out_x[0][0], &out_zw[0][0]};
    PK_TRY(pm::poly_evaluate_groups(ctx, 3, gk, gp, gpt, gout, n));
    pm::host_mark(ctx, "openings at z, z w");
    for (int j = 0; j < 4; ++j) ev[E_A + j] = get(out_z[j]);
    for (int j = 0; j < 3; ++j) ev[E_S1 + j] = get(out_z[4 + j]);
    ev[E_QARITH] = get(out_z[7]);
    ev[E_QC] = get(out_z[8]);
    ev[E_QL] = get(out_z[9]);
    ev[E_QR] = get(out_z[10]);
    ev[E_AN] = get(out_zw[0]);
    ev[E_BN] = get(out_zw[1]);
    ev[E_DN] = get(out_zw[2]);
    ev[E_ZN] = get(out_zw[3]);
    const HFr zn_ = fpow(zc, n);
    ev[E_T] = fadd(get(out_z[11]), fmul(zn_, fadd(get(out_z[12]), fmul(zn_, fadd(get(out_z[13]), fmul(zn_, get(out_z[14])))))));
    for (int j = 0; j < X_RANGE; ++j) xv[j] = get(out_x[j]);
    for (int w = 0; w < 4; ++w) xv[X_RANGE + w] = xslot[w] >= 0 ? get(out_x[xslot[w]]) : pm::host::zero<4>();
  }
  const HFr zn = fpow(zc, n);
  const HFr &a_ = ev[E_A], &b_ = ev[E_B], &c_ = ev[E_C], &d_ = ev[E_D], &s1 = ev[E_S1], &s2 = ev[E_S2], &s3 = ev[E_S3],
            &z_next = ev[E_ZN], &qar = ev[E_QARITH];
  const HFr l1_z = fmul(fsub(zn, one), finv(fmul(fr_u64(n), fsub(zc, one))));
  const HFr bz = fmul(beta, zc);
  HFr ident = fadd(fadd(a_, bz), gamma);
  const HFr* wv[3] = {&b_, &c_, &d_};
  for (int j = 0; j < 3; ++j) ident = fmul(ident, fadd(fadd(*wv[j], fmul(bz, pk->k[j])), gamma));
  const HFr copy3 = fmul(fmul(fadd(fadd(a_, fmul(beta, s1)), gamma), fadd(fadd(b_, fmul(beta, s2)), gamma)),
                         fadd(fadd(c_, fmul(beta, s3)), gamma));
  const HFr alpha2 = fmul(alpha, alpha);
  RowEvals re{a_, b_, c_, d_, ev[E_AN], ev[E_BN], ev[E_DN], ev[E_QL], ev[E_QR], ev[E_QC]};
  {
    const void* lin_v[12];
    u64 lin_c[12][4];
    uint32_t k = 0;
    HFr r_z = pm::host::zero<4>();   // r(z) = sum of coefficient x value at z, term by term
    auto term = [&](const void* v, const HFr& c, const HFr& value_at_z) {
      lin_v[k] = v;
      put(lin_c[k], c);
      r_z = fadd(r_z, fmul(c, value_at_z));
      ++k;
    };
    // arithmetic: q_arith(z) (a b q_m + a q_l + b q_r + c q_o + d q_4 + q_c)
    term(at(pk->sel_coeffs, Q_M * n), fmul(qar, fmul(a_, b_)), xv[X_QM]);
    term(at(pk->sel_coeffs, Q_L * n), fmul(qar, a_), ev[E_QL]);
    term(at(pk->sel_coeffs, Q_R * n), fmul(qar, b_), ev[E_QR]);
    term(at(pk->sel_coeffs, Q_O * n), fmul(qar, c_), xv[X_QO]);
    term(at(pk->sel_coeffs, Q_4 * n), fmul(qar, d_), xv[X_Q4]);
    term(at(pk->sel_coeffs, Q_C * n), qar, ev[E_QC]);
    if (!pk->sel_zero[Q_RANGE]) term(at(pk->sel_coeffs, Q_RANGE * n), widget_range(range_sep, re), xv[X_RANGE]);
    if (!pk->sel_zero[Q_LOGIC]) term(at(pk->sel_coeffs, Q_LOGIC * n), widget_logic(logic_sep, re), xv[X_LOGIC]);
    if (!pk->sel_zero[Q_FIXED]) term(at(pk->sel_coeffs, Q_FIXED * n), widget_fixed(fixed_sep, re), xv[X_FIXED]);
    if (!pk->sel_zero[Q_VAR]) term(at(pk->sel_coeffs, Q_VAR * n), widget_var(var_sep, re), xv[X_VAR]);
    term(z_coeffs, fadd(fmul(alpha, ident), fmul(alpha2, l1_z)), xv[X_Z]);
    term(at(pk->sigma_coeffs, 3 * n), fneg(fmul(fmul(fmul(alpha, copy3), beta), z_next)), xv[X_S4]);
    PK_TRY(pm_fr_lincomb_dev(ctx, k, lin_v, &lin_c[0][0], n, pk->r, nullptr));   // r itself: round 5 divides it
    ev[E_R] = r_z;
  }
  static_assert(NEV == 17, "tl::EVALS lists the evaluations in this enum's order");
  for (int i = 0; i < NEV; ++i) {
    ts.append_scalar(tl::EVALS[i], ev[i]);
    put(out->evaluations[i], ev[i]);
  }
  // ---- round 5: CommitKey::compute_aggregate_witness at z and at z w ------------------------------
  const HFr aw = ts.challenge_scalar(tl::AGGREGATE);
  {
    const void* agg_v[12];
    u64 agg_c[12][4];
    HFr ac[12];
    ac[0] = one;                      // quot = t_1 + z^n t_2 + z^2n t_3 + z^3n t_4 comes first (power 0)
    ac[1] = zn;
    ac[2] = fmul(zn, zn);
    ac[3] = fmul(ac[2], zn);
    HFr vp = one;
    for (int e = 0; e < 8; ++e) {     // then lin, w_l, w_r, w_o, w_4, left, right, out sigma
      vp = fmul(vp, aw);
      ac[4 + e] = vp;
    }
    for (int i = 0; i < 4; ++i) agg_v[i] = at(pk->t, i * n);
    agg_v[4] = pk->r;
    for (int j = 0; j < 4; ++j) agg_v[5 + j] = at(pk->coeffs, j * n);
    for (int j = 0; j < 3; ++j) agg_v[9 + j] = at(pk->sigma_coeffs, j * n);
    for (int i = 0; i < 12; ++i) put(agg_c[i], ac[i]);
    PK_TRY(pm_fr_lincomb_dev(ctx, 12, agg_v, &agg_c[0][0], n, pk->agg, nullptr));
    PK_TRY(pm_fr_poly_ruffini_dev(ctx, pk->agg, n, zc.l, pk->wit, nullptr));
  }
  const HFr aws = ts.challenge_scalar(tl::AGGREGATE);
  {
    const void* sh_v[4] = {z_coeffs, at(pk->coeffs, 0), at(pk->coeffs, n), at(pk->coeffs, 3 * n)};
    u64 sh_c[4][4];
    HFr vp = one;
    for (int e = 0; e < 4; ++e) {
      put(sh_c[e], vp);
      vp = fmul(vp, aws);
    }
    PK_TRY(pm_fr_lincomb_dev(ctx, 4, sh_v, &sh_c[0][0], n, pk->agg, nullptr));
    PK_TRY(pm_fr_poly_ruffini_dev(ctx, pk->agg, n, zw.l, at(pk->wit, n), nullptr));
  }
  PK_TRY(commit_batch(ctx, ck, shard, pk->wit, n - 1, n, 2, &out->commitments[9]));
  ts.append_commitment(tl::W_Z, out->commitments[9]);
  ts.append_commitment(tl::W_ZW, out->commitments[10]);
  const HFr chal[10] = {beta, gamma, alpha, range_sep, logic_sep, fixed_sep, var_sep, zc, aw, aws};
  for (int i = 0; i < 10; ++i) put(out->challenges[i], chal[i]);
  return PM_OK;
}

// Proof::to_bytes of dusk-plonk 0.8: 11 compressed G1 (a b c d z t_1..t_4 w_z w_zw) then the 16 scalars of
// ProofEvaluations::to_bytes (a b c d a_next b_next d_next q_arith q_c q_l q_r left right out sigma, lin_poly, perm).
extern "C" int pm_plonk_proof_to_bytes(const pm_plonk_proof* proof, uint8_t out[PM_PLONK_PROOF_BYTES]) {
  if (!proof || !out) return PM_ERR_BAD_ARG;
  for (int i = 0; i < 11; ++i) Transcript::g1_compress(out + 48 * i, proof->commitments[i]);
  static const int order[16] = {0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 7, 8, 9, 16, 14};
  for (int i = 0; i < 16; ++i) Transcript::scalar_bytes(out + 528 + 32 * i, get(proof->evaluations[order[i]]));
  return PM_OK;
}

extern "C" int pm_plonk_verifier_key(const pm_prover_key* key, uint64_t (*out)[12]) {
  if (!key || !out || !key->committed) return PM_ERR_BAD_ARG;
  memcpy(out, key->vk, sizeof key->vk);
  return PM_OK;
}

#include "prover_dist.hip.h"
