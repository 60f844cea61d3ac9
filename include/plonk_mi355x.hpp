// C++17 host-side mirror of the reference's dependency interface for the hot path, header-only over
// the C ABI of plonk_mi355x.h.  The reference is compiled code (Rust: dusk-plonk 0.8.2 /
// dusk-bls12_381 0.8, ref:Cargo.toml:19-20) and no Rust toolchain exists in the build image, so this
// is the host language the boundary is written in; names, argument meaning and error behaviour
// follow the crates:
//
//   dusk_plonk::fft::EvaluationDomain::{new, fft, ifft, coset_fft, coset_ifft, *_in_place, elements}
//   dusk_bls12_381::multiscalar_mul::msm_variable_base(points, scalars) -> G1Projective
//   dusk_plonk::commitment_scheme::kzg10::CommitKey::{commit, max_degree}   (+ setup, N4)
//   dusk_plonk::fft::Polynomial::{evaluate, ruffini} and coefficient-wise + - *   (device resident)
//
// Memory layouts are the Rust types' own: Fr = BlsScalar([u64; 4]) Montgomery limbs, G1Affine =
// (x, y) of 6 Montgomery limbs each with (0, 0) for the identity, G1Projective = (X, Y, Z).
// Errors: plonk_mi355x::Error carries the pm_status code (InvalidEvalDomainSize = PM_ERR_DOMAIN_TOO_LARGE,
// PolynomialDegreeTooLarge = PM_ERR_LENGTH).  There is no CPU fallback: Context() throws without a
// gfx950 device.  examples/host_demo.cpp exercises everything below.
#ifndef PLONK_MI355X_HPP
#define PLONK_MI355X_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "plonk_mi355x.h"

namespace plonk_mi355x {

using Fr = std::array<uint64_t, 4>;             // dusk_bls12_381::BlsScalar
using G1Affine = std::array<uint64_t, 12>;      // x | y, (0, 0) = identity
using G1Projective = std::array<uint64_t, 18>;  // X | Y | Z, normalised by the library (Z = 1 or (0, 1, 0))

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class Context {
 public:
  explicit Context(int device = 0) {
    int rc = pm_init(device, &h_);
    if (rc != PM_OK) throw Error(rc, "pm_init: no usable gfx950 device (there is no CPU fallback)");
  }
  ~Context() { if (h_) pm_shutdown(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  pm_ctx* get() const { return h_; }
  void check(int rc) const { if (rc != PM_OK) throw Error(rc, pm_last_error(h_)); }
  void sync() const { check(pm_sync(h_)); }
  // release cached workspaces / twiddle tables (regrown on demand); returns the bytes given back
  size_t trim() const { size_t f = 0; check(pm_trim(h_, &f)); return f; }
  // exchange counters since the last reset (pm_comm_stats): what a multi-GPU deployment pays per proof, countable on one GPU
  struct CommStats { uint64_t alltoall_calls, alltoall_bytes, allgather_calls, transpose_steps; };
  CommStats comm_stats(bool reset = false) const {
    uint64_t v[4];
    check(pm_comm_stats(h_, v, reset ? 1 : 0));
    return CommStats{v[0], v[1], v[2], v[3]};
  }

 private:
  pm_ctx* h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
class EvaluationDomain {
 public:
  uint64_t size = 0;
  uint32_t log_size_of_group = 0;
  Fr size_inv{}, group_gen{}, group_gen_inv{};

  // EvaluationDomain::new(num_coeffs): next power of two; Error{PM_ERR_DOMAIN_TOO_LARGE} when
  // log2(size) >= TWO_ADICITY (upstream: Error::InvalidEvalDomainSize)
  EvaluationDomain(Context& ctx, size_t num_coeffs) : ctx_(&ctx) {
    size = 1;
    while (size < num_coeffs) { size <<= 1; ++log_size_of_group; }
    int rc = pm_domain_info(log_size_of_group, group_gen.data(), group_gen_inv.data(), size_inv.data());
    if (rc != PM_OK) throw Error(rc, "InvalidEvalDomainSize: log_size_of_group >= TWO_ADICITY");
  }
  std::vector<Fr> fft(const std::vector<Fr>& coeffs) const { return run(coeffs, 0); }
  std::vector<Fr> ifft(const std::vector<Fr>& evals) const { return run(evals, PM_NTT_INVERSE); }
  std::vector<Fr> coset_fft(const std::vector<Fr>& coeffs) const { return run(coeffs, PM_NTT_COSET); }
  std::vector<Fr> coset_ifft(const std::vector<Fr>& evals) const { return run(evals, PM_NTT_INVERSE | PM_NTT_COSET); }
  void fft_in_place(std::vector<Fr>& a) const { a = run(a, 0); }
  void ifft_in_place(std::vector<Fr>& a) const { a = run(a, PM_NTT_INVERSE); }
  void coset_fft_in_place(std::vector<Fr>& a) const { a = run(a, PM_NTT_COSET); }
  void coset_ifft_in_place(std::vector<Fr>& a) const { a = run(a, PM_NTT_INVERSE | PM_NTT_COSET); }
  // several equal-length polynomials in one call (a prover round): uploads, transforms and downloads overlap
  std::vector<std::vector<Fr>> fft_many(const std::vector<std::vector<Fr>>& polys, uint32_t flags = 0) const {
    if (polys.empty()) return {};
    const size_t len = polys[0].size();
    std::vector<Fr> in(polys.size() * len), out(polys.size() * size);
    for (size_t b = 0; b < polys.size(); ++b) {
      if (polys[b].size() != len) throw Error(PM_ERR_LENGTH, "fft_many: polynomials differ in length");
      std::copy(polys[b].begin(), polys[b].end(), in.begin() + b * len);
    }
    ctx_->check(pm_fr_ntt_batch(ctx_->get(), in.empty() ? nullptr : in[0].data(), len, len, out[0].data(), size,
                                log_size_of_group, (uint32_t)polys.size(), flags));
    std::vector<std::vector<Fr>> res(polys.size());
    for (size_t b = 0; b < polys.size(); ++b) res[b].assign(out.begin() + b * size, out.begin() + (b + 1) * size);
    return res;
  }
  // all domain elements 1, g, g^2, ...
  std::vector<Fr> elements() const {
    std::vector<Fr> x(size > 1 ? 2 : 1, Fr{0, 0, 0, 0});
    x.back() = one();
    return fft(x);
  }
  // evaluate_vanishing_polynomial(tau) = tau^size - 1
  Fr evaluate_vanishing_polynomial(const Fr& tau) const {
    Fr out{};
    int rc = pm_domain_evaluate_vanishing_polynomial(log_size_of_group, tau.data(), out.data());
    if (rc != PM_OK) throw Error(rc, "pm_domain_evaluate_vanishing_polynomial");
    return out;
  }
  // evaluate_all_lagrange_coefficients(tau): [L_0(tau), .., L_(size-1)(tau)]
  std::vector<Fr> evaluate_all_lagrange_coefficients(const Fr& tau) const {
    std::vector<Fr> out(size);
    int rc = pm_domain_evaluate_all_lagrange_coefficients(log_size_of_group, tau.data(), out[0].data());
    if (rc != PM_OK) throw Error(rc, "pm_domain_evaluate_all_lagrange_coefficients");
    return out;
  }
  // compute_vanishing_poly_over_coset(*this, poly_degree): X^poly_degree - 1 on GENERATOR * H, over this domain
  // (upstream asserts size > poly_degree: Error{PM_ERR_BAD_ARG})
  std::vector<Fr> compute_vanishing_poly_over_coset(uint64_t poly_degree) const {
    std::vector<Fr> out(size);
    int rc = pm_domain_vanishing_poly_over_coset(log_size_of_group, poly_degree, out[0].data());
    if (rc != PM_OK) throw Error(rc, "compute_vanishing_poly_over_coset: domain size > poly_degree violated");
    return out;
  }
  static Fr one() { return Fr{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}; }

 private:
  std::vector<Fr> run(const std::vector<Fr>& a, uint32_t flags) const {
    if (a.size() > size) throw Error(PM_ERR_LENGTH, "input longer than the domain");
    std::vector<Fr> out(size);
    const Fr zero{0, 0, 0, 0};
    ctx_->check(pm_fr_ntt(ctx_->get(), a.empty() ? zero.data() : a[0].data(), a.size(), out[0].data(),
                          log_size_of_group, flags));
    return out;
  }
  Context* ctx_;
};

// ------------------------------------------------------------------------------------------------
inline G1Affine to_affine(const G1Projective& p, bool* is_identity = nullptr) {
  G1Affine out{};
  int ident = 0;
  int rc = pm_g1_to_affine(p.data(), out.data(), &ident);
  if (rc != PM_OK) throw Error(rc, "pm_g1_to_affine");
  if (is_identity) *is_identity = ident != 0;
  return out;
}

// msm_variable_base(points, scalars): bases uploaded for this one call
inline G1Projective msm_variable_base(Context& ctx, const std::vector<G1Affine>& points, const std::vector<Fr>& scalars) {
  if (points.size() != scalars.size()) throw Error(PM_ERR_LENGTH, "points and scalars differ in length");
  G1Projective out{};
  pm_bases* b = nullptr;
  const G1Affine dummy{};
  ctx.check(pm_g1_bases_upload(ctx.get(), points.empty() ? dummy.data() : points[0].data(), points.size(), &b));
  const Fr zero{0, 0, 0, 0};
  int rc = pm_g1_msm(ctx.get(), b, scalars.size(), scalars.empty() ? zero.data() : scalars[0].data(),
                     PM_SCALAR_MONTGOMERY, out.data());
  pm_g1_bases_free(ctx.get(), b);
  ctx.check(rc);
  return out;
}

class CommitKey {
 public:
  // CommitKey { powers_of_g }: the SRS stays resident in HBM; precompute = also its window table
  CommitKey(Context& ctx, const std::vector<G1Affine>& powers_of_g, bool precompute = false) : ctx_(&ctx) {
    const G1Affine dummy{};
    ctx.check(pm_g1_bases_upload(ctx.get(), powers_of_g.empty() ? dummy.data() : powers_of_g[0].data(),
                                 powers_of_g.size(), &bases_));
    if (precompute) ctx.check(pm_g1_bases_precompute(ctx.get(), bases_, 0));
  }
  ~CommitKey() { if (bases_) pm_g1_bases_free(ctx_->get(), bases_); }
  CommitKey(const CommitKey&) = delete;
  CommitKey& operator=(const CommitKey&) = delete;
  size_t max_degree() const { return pm_g1_bases_len(bases_) - 1; }
  // commit(polynomial): Error{PM_ERR_LENGTH} = PolynomialDegreeTooLarge
  G1Affine commit(const std::vector<Fr>& coeffs) const {
    if (coeffs.size() > pm_g1_bases_len(bases_)) throw Error(PM_ERR_LENGTH, "PolynomialDegreeTooLarge");
    G1Projective p{};
    const Fr zero{0, 0, 0, 0};
    ctx_->check(pm_g1_msm(ctx_->get(), bases_, coeffs.size(), coeffs.empty() ? zero.data() : coeffs[0].data(),
                          PM_SCALAR_MONTGOMERY, p.data()));
    return to_affine(p);
  }
  const pm_bases* bases() const { return bases_; }

 private:
  Context* ctx_;
  pm_bases* bases_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
// dusk_plonk::fft::Polynomial kept in device memory between NTT and MSM calls
class DevicePolynomial {
 public:
  DevicePolynomial(Context& ctx, size_t n) : ctx_(&ctx), n_(n) { ctx.check(pm_dev_alloc(ctx.get(), n * 32, &p_)); }
  DevicePolynomial(Context& ctx, const std::vector<Fr>& coeffs) : DevicePolynomial(ctx, coeffs.size()) {
    if (n_) ctx.check(pm_dev_upload(ctx.get(), p_, coeffs[0].data(), n_ * 32));
  }
  ~DevicePolynomial() { if (p_) pm_dev_free(ctx_->get(), p_); }
  DevicePolynomial(const DevicePolynomial&) = delete;
  DevicePolynomial& operator=(const DevicePolynomial&) = delete;
  DevicePolynomial(DevicePolynomial&& o) noexcept : ctx_(o.ctx_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
  size_t len() const { return n_; }
  void* data() const { return p_; }
  std::vector<Fr> to_host() const {
    std::vector<Fr> out(n_);
    if (n_) ctx_->check(pm_dev_download(ctx_->get(), out[0].data(), p_, n_ * 32));
    return out;
  }
  DevicePolynomial add(const DevicePolynomial& o) const { return op(0, o); }
  DevicePolynomial sub(const DevicePolynomial& o) const { return op(1, o); }
  DevicePolynomial mul(const DevicePolynomial& o) const { return op(2, o); }   // coefficient-wise / scalar (len 1)
  Fr evaluate(const Fr& point) const {
    Fr out{};
    ctx_->check(pm_fr_poly_evaluate_dev(ctx_->get(), p_, n_, point.data(), out.data(), nullptr));
    return out;
  }
  DevicePolynomial ruffini(const Fr& z) const {   // quotient by (X - z), remainder dropped
    DevicePolynomial q(*ctx_, n_ ? n_ - 1 : 0);
    ctx_->check(pm_fr_poly_ruffini_dev(ctx_->get(), p_, n_, z.data(), q.p_, nullptr));
    return q;
  }
  // NTT of the resident coefficients onto a domain of 2^log_n points (zero padded)
  DevicePolynomial ntt(uint32_t log_n, uint32_t flags) const {
    DevicePolynomial out(*ctx_, (size_t)1 << log_n);
    ctx_->check(pm_fr_ntt_dev(ctx_->get(), p_, n_, n_, out.p_, out.n_, log_n, 1, flags, nullptr));
    return out;
  }
  G1Affine commit(const CommitKey& ck) const {
    if (n_ > ck.max_degree() + 1) throw Error(PM_ERR_LENGTH, "PolynomialDegreeTooLarge");
    G1Projective p{};
    ctx_->check(pm_g1_msm_dev(ctx_->get(), ck.bases(), 0, n_, p_, PM_SCALAR_MONTGOMERY, p.data(), nullptr));
    return to_affine(p);
  }

 private:
  DevicePolynomial op(int code, const DevicePolynomial& o) const {
    if (o.n_ != 1 && o.n_ != n_) throw Error(PM_ERR_LENGTH, "operands differ in length");
    DevicePolynomial out(*ctx_, n_);
    ctx_->check(pm_fr_vec_op_dev(ctx_->get(), code, p_, o.p_, o.n_, out.p_, n_, nullptr));
    return out;
  }
  Context* ctx_;
  void* p_ = nullptr;
  size_t n_ = 0;
};

// ------------------------------------------------------------------------------------------------
// dusk_plonk::proof_system::{ProverKey, Prover::preprocess, Prover::prove_with_preprocessed, Proof}: the 11
// selector polynomials of dusk-plonk 0.8 (arithmetic, range, logic, fixed-base, variable-base) and the
// 4-wire permutation; one library call per proof (pm_plonk_prove), everything resident in HBM.
struct Proof {
  std::array<G1Affine, 11> commitments;                  // a b c d z t_1 t_2 t_3 t_4 w_z w_zw
  std::array<Fr, PM_PLONK_EVALS> evaluations;            // transcript order, see plonk_mi355x.h
  std::array<Fr, PM_PLONK_CHALLENGES> challenges;
  std::array<uint8_t, PM_PLONK_PROOF_BYTES> bytes;       // Proof::to_bytes
};
struct PublicInput {
  uint64_t position;   // gate index
  Fr value;
};

class ProverKey {
 public:
  // selectors: q_m q_l q_r q_o q_c q_4 q_arith q_range q_logic q_fixed_group_add q_variable_group_add, n
  // evaluations on H each (an empty vector = identically zero); sigma_index[j n + i] = successor position.
  // The verifier key is committed with `ck` and seeds the transcript every proof starts from.
  ProverKey(Context& ctx, const std::array<std::vector<Fr>, PM_PLONK_SELECTORS>& selectors,
            const std::vector<int64_t>& sigma_index, const CommitKey& ck, const char* transcript_label = nullptr)
      : ctx_(&ctx), n_(selectors[0].size()) {
    const uint64_t* ptrs[PM_PLONK_SELECTORS];
    for (int s = 0; s < PM_PLONK_SELECTORS; ++s) {
      if (selectors[s].empty()) {
        ptrs[s] = nullptr;
        continue;
      }
      if (selectors[s].size() != n_ || n_ == 0) throw Error(PM_ERR_LENGTH, "selectors differ in length");
      ptrs[s] = selectors[s][0].data();
    }
    if (sigma_index.size() != 4 * n_) throw Error(PM_ERR_LENGTH, "sigma_index must have 4n entries");
    ctx.check(pm_plonk_preprocess(ctx.get(), ptrs, sigma_index.data(), n_, &key_));
    uint64_t vk[PM_PLONK_VK_POINTS][12];
    int rc = pm_plonk_key_commit(ctx.get(), key_, ck.bases(), transcript_label, vk);
    if (rc) {
      pm_plonk_key_free(ctx.get(), key_);
      key_ = nullptr;
      ctx.check(rc);
    }
    for (int i = 0; i < PM_PLONK_VK_POINTS; ++i) std::copy(vk[i], vk[i] + 12, verifier_key_[i].begin());
  }
  ~ProverKey() { if (key_) pm_plonk_key_free(ctx_->get(), key_); }
  ProverKey(const ProverKey&) = delete;
  ProverKey& operator=(const ProverKey&) = delete;
  size_t n() const { return n_; }
  const std::array<G1Affine, PM_PLONK_VK_POINTS>& verifier_key() const { return verifier_key_; }
  // witness: [a | b | c | d] on the device (4n)
  Proof prove(const CommitKey& ck, const DevicePolynomial& witness, const std::vector<PublicInput>& public_inputs = {},
              bool bind_public_inputs = true) const {
    if (witness.len() != 4 * n_) throw Error(PM_ERR_LENGTH, "the witness must hold 4n wire values");
    std::vector<uint64_t> pos, val;
    for (const PublicInput& pi : public_inputs) {
      pos.push_back(pi.position);
      val.insert(val.end(), pi.value.begin(), pi.value.end());
    }
    pm_plonk_proof raw;
    ctx_->check(pm_plonk_prove(ctx_->get(), key_, ck.bases(), witness.data(), pos.data(), val.data(), pos.size(),
                               bind_public_inputs ? 0u : PM_PLONK_UPSTREAM_TRANSCRIPT, &raw));
    Proof p;
    for (int i = 0; i < 11; ++i) std::copy(raw.commitments[i], raw.commitments[i] + 12, p.commitments[i].begin());
    for (int i = 0; i < PM_PLONK_EVALS; ++i) std::copy(raw.evaluations[i], raw.evaluations[i] + 4, p.evaluations[i].begin());
    for (int i = 0; i < PM_PLONK_CHALLENGES; ++i) std::copy(raw.challenges[i], raw.challenges[i] + 4, p.challenges[i].begin());
    ctx_->check(pm_plonk_proof_to_bytes(&raw, p.bytes.data()));
    return p;
  }

 private:
  Context* ctx_;
  size_t n_;
  pm_prover_key* key_ = nullptr;
  std::array<G1Affine, PM_PLONK_VK_POINTS> verifier_key_;
};

// The same prover with every vector split over `dist.world` ranks by coefficient range (pm_plonk_*_dist, DESIGN.md
// section 7.6): this rank is given its m = n / world rows of every selector column and of the permutation, its slice
// [rank m, (rank + 1) m) of the commit key, and holds 1 / world of the key and workspace.  `dist` carries the two
// collectives (both null: the context's RCCL communicator, pm_comm_init); every rank gets the same proof.
class DistProverKey {
 public:
  DistProverKey(Context& ctx, const pm_dist& dist, const std::array<std::vector<Fr>, PM_PLONK_SELECTORS>& selector_slices,
                const std::vector<int64_t>& sigma_index_slices, size_t n, const CommitKey& ck_slice,
                const char* transcript_label = nullptr)
      : ctx_(&ctx), dist_(dist), n_(n) {
    const size_t m = n / dist.world;
    const uint64_t* ptrs[PM_PLONK_SELECTORS];
    for (int s = 0; s < PM_PLONK_SELECTORS; ++s) {
      if (selector_slices[s].empty()) {
        ptrs[s] = nullptr;
        continue;
      }
      if (selector_slices[s].size() != m) throw Error(PM_ERR_LENGTH, "a selector slice must hold n / world rows");
      ptrs[s] = selector_slices[s][0].data();
    }
    if (sigma_index_slices.size() != 4 * m) throw Error(PM_ERR_LENGTH, "sigma_index_slices must have 4 n / world entries");
    ctx.check(pm_plonk_preprocess_dist(ctx.get(), &dist_, ptrs, sigma_index_slices.data(), n, &key_));
    uint64_t vk[PM_PLONK_VK_POINTS][12];
    int rc = pm_plonk_key_commit_dist(ctx.get(), &dist_, key_, ck_slice.bases(), transcript_label, vk);
    if (rc) {
      pm_plonk_dist_key_free(ctx.get(), key_);
      key_ = nullptr;
      ctx.check(rc);
    }
    for (int i = 0; i < PM_PLONK_VK_POINTS; ++i) std::copy(vk[i], vk[i] + 12, verifier_key_[i].begin());
  }
  ~DistProverKey() { if (key_) pm_plonk_dist_key_free(ctx_->get(), key_); }
  DistProverKey(const DistProverKey&) = delete;
  DistProverKey& operator=(const DistProverKey&) = delete;
  size_t device_bytes() const { return pm_plonk_dist_key_bytes(key_); }
  const std::array<G1Affine, PM_PLONK_VK_POINTS>& verifier_key() const { return verifier_key_; }
  // witness_slices: this rank's [a | b | c | d] rows on the device (4 n / world); public inputs whole, the same on every rank
  Proof prove(const CommitKey& ck_slice, const DevicePolynomial& witness_slices, const std::vector<PublicInput>& public_inputs = {},
              bool bind_public_inputs = true) const {
    if (witness_slices.len() != 4 * (n_ / dist_.world)) throw Error(PM_ERR_LENGTH, "the witness must hold 4 n / world wire values");
    std::vector<uint64_t> pos, val;
    for (const PublicInput& pi : public_inputs) {
      pos.push_back(pi.position);
      val.insert(val.end(), pi.value.begin(), pi.value.end());
    }
    pm_plonk_proof raw;
    ctx_->check(pm_plonk_prove_dist(ctx_->get(), &dist_, key_, ck_slice.bases(), witness_slices.data(), pos.data(), val.data(),
                                    pos.size(), bind_public_inputs ? 0u : PM_PLONK_UPSTREAM_TRANSCRIPT, &raw));
    Proof p;
    for (int i = 0; i < 11; ++i) std::copy(raw.commitments[i], raw.commitments[i] + 12, p.commitments[i].begin());
    for (int i = 0; i < PM_PLONK_EVALS; ++i) std::copy(raw.evaluations[i], raw.evaluations[i] + 4, p.evaluations[i].begin());
    for (int i = 0; i < PM_PLONK_CHALLENGES; ++i) std::copy(raw.challenges[i], raw.challenges[i] + 4, p.challenges[i].begin());
    ctx_->check(pm_plonk_proof_to_bytes(&raw, p.bytes.data()));
    return p;
  }

 private:
  Context* ctx_;
  pm_dist dist_;
  size_t n_;
  pm_dist_key* key_ = nullptr;
  std::array<G1Affine, PM_PLONK_VK_POINTS> verifier_key_;
};

}  // namespace plonk_mi355x
#endif  // PLONK_MI355X_HPP
