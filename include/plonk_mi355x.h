/*
 * plonk_mi355x.h -- C ABI of the MI355X (gfx950) backend for the PLONK prover hot path:
 * BLS12-381 Fr radix-2 NTT / iNTT / coset-NTT and G1 variable-base MSM (KZG commit).
 *
 * Drop-in boundary.  The reference (Manta-Network/Plonk-Prototype) reaches this path only
 * through its dependencies (ref:Cargo.toml:19 `dusk-plonk = 0.8.2`, ref:Cargo.toml:20
 * `dusk-bls12_381 = 0.8`): there is no FFI or plugin interface upstream, `best_fft` and
 * `msm_variable_base` are plain Rust functions.  The entry points below are therefore
 * exactly what a `-sys` crate for a `[patch.crates-io]` fork of those two crates binds
 * (INTEGRATION.md shows the Rust side):
 *
 *   pm_fr_ntt / pm_fr_ntt_batch  <- dusk_plonk::fft::EvaluationDomain::{fft, ifft, coset_fft,
 *                                   coset_ifft}(_in_place)   [ark-poly: EvaluationDomain::
 *                                   {fft, ifft, coset_fft, coset_ifft}]   SURVEY.md 8a a3-a7
 *   pm_domain_info               <- EvaluationDomain::new  (group_gen, group_gen_inv,
 *                                   size_inv; error when log2(size) >= 32)  SURVEY.md 8a a2
 *   pm_g1_bases_upload           <- CommitKey { powers_of_g }  (device-resident SRS)   a10
 *   pm_g1_msm                    <- dusk_bls12_381::multiscalar_mul::msm_variable_base
 *                                   [ark-ec: VariableBaseMSM::multi_scalar_mul]        a9
 *   pm_g1_fold / pm_g1_to_affine <- G1Projective `+` / G1Affine::from (multi-GPU fold)  a8
 *   pm_fr_*_dev, pm_plonk_*_dev  <- fft::{Polynomial, Evaluations}, proof_system::{permutation,
 *                                   quotient_poly, linearisation_poly} pointwise work   8f N1/N2
 *   pm_plonk_preprocess / _prove <- proof_system::{ProverKey, Prover::prove_with_preprocessed}:
 *                                   the whole five-round prover behind one call        8f N1-N3
 *   pm_g1_fixed_base_mul_dev     <- PublicParameters::setup (powers of tau)             8f N4
 * (include/plonk_mi355x.hpp is the C++ mirror of the same interfaces.)
 *
 * Data layouts are the Rust types' memory, so slices can be passed without marshalling:
 *   Fr  (BlsScalar / ark Fr)  : 4 x uint64_t little-endian limbs, Montgomery form R = 2^256,
 *                               fully reduced.  32 bytes, 8-byte aligned.
 *   Fp                        : 6 x uint64_t little-endian limbs, Montgomery form R = 2^384.
 *   G1 affine base            : x[6] | y[6]  (96 bytes packed).  The point at infinity is
 *                               encoded as x = y = 0 (not on the curve, so unambiguous).
 *   G1 projective result      : X[6] | Y[6] | Z[6] (144 bytes), always normalised to Z = 1
 *                               (Montgomery one) or (0, 1, 0) for the identity -- valid both
 *                               as dusk/zkcrypto homogeneous and as ark Jacobian coordinates.
 *
 * Ownership: the caller owns every host buffer; the library never keeps a host pointer past
 * return.  pm_ctx / pm_bases are library-owned handles freed by pm_shutdown / pm_g1_bases_free.
 * Errors: every call returns PM_OK (0) or a negative pm_status; nothing unwinds across the
 * ABI.  pm_last_error(ctx) gives a human-readable string for the last failure on that ctx.
 * Threading: calls on one ctx are serialised by an internal mutex; use one ctx per thread
 * for concurrency (a pm_prover_key carries its proof workspace: one proof at a time per key).  One ctx drives one GPU (one process per GPU under torch.distributed).
 * There is no CPU fallback: without a usable gfx950 device pm_init fails with
 * PM_ERR_NO_DEVICE.
 */
#ifndef PLONK_MI355X_H
#define PLONK_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pm_ctx pm_ctx;
typedef struct pm_bases pm_bases;

typedef enum {
  PM_OK = 0,
  PM_ERR_BAD_ARG = -1,
  PM_ERR_DOMAIN_TOO_LARGE = -2, /* log_n >= 32 = Fr two-adicity (EvaluationDomain::new fails) */
  PM_ERR_OOM = -3,
  PM_ERR_HIP = -4,
  PM_ERR_NO_DEVICE = -5,
  PM_ERR_LENGTH = -6,           /* in_len > 2^log_n, or n > number of uploaded bases */
  PM_ERR_EXCHANGE = -7,         /* a multi-GPU exchange (callback or RCCL) failed or a peer gave up */
  PM_ERR_BUSY = -8              /* the prover key's workspace is in use by another proof */
} pm_status;

/* pm_fr_ntt flags */
#define PM_NTT_FORWARD 0u
#define PM_NTT_INVERSE 1u /* use group_gen_inv and scale by size_inv (ifft) */
#define PM_NTT_COSET 2u   /* fft: pre-scale a[i] *= 7^i ; with INVERSE: post-scale a[i] *= 7^-i */
#define PM_NTT_TRANSPOSED 4u /* pm_fr_ntt_fourstep_dev only: block-transposed order between a forward and an inverse transform */

/* pm_g1_msm scalar_form */
#define PM_SCALAR_MONTGOMERY 0u /* dusk `&[BlsScalar]` memory */
#define PM_SCALAR_CANONICAL 1u  /* ark `&[BigInteger256]` memory (plain integers < r) */

const char* pm_version(void);

/* Create a context on HIP device `device_id` (stream, twiddle caches, workspaces). */
int pm_init(int device_id, pm_ctx** out);
void pm_shutdown(pm_ctx* ctx);
const char* pm_last_error(const pm_ctx* ctx);
/* Block until everything queued on the context's stream has finished. */
int pm_sync(pm_ctx* ctx);
/* The context's own HIP stream (a hipStream_t; what hip_stream = NULL means in the *_dev entry points): for callers that
 * order their own work after the library's, or record timing events around it.  Owned by the context. */
void* pm_ctx_stream(pm_ctx* ctx);
/* Give back what the context caches between calls -- pass buffers, MSM / polynomial workspaces, staging, twiddle tables --
 * after waiting for the device; everything is rebuilt or regrown on demand (a workspace only ever grows otherwise: after
 * one 2^30-point transform a context holds 77 GB of pass buffers).  pm_bases and pm_prover_key objects are untouched.
 * freed_bytes (may be NULL): device memory returned to the driver.  May be called at any time from any thread; while a
 * pm_fr_ntt_fourstep_dev call of another thread is between two of its exchange steps on this context (it keeps table
 * pointers across them) the call changes nothing and returns PM_ERR_BUSY. */
int pm_trim(pm_ctx* ctx, size_t* freed_bytes);

/* ---- EvaluationDomain ------------------------------------------------------------------ */

/* EvaluationDomain::new(2^log_n): writes group_gen, group_gen_inv, size_inv (Montgomery Fr).
 * Pure host arithmetic, needs no context.  PM_ERR_DOMAIN_TOO_LARGE when log_n >= 32. */
int pm_domain_info(uint32_t log_n, uint64_t group_gen[4], uint64_t group_gen_inv[4],
                   uint64_t size_inv[4]);

/* EvaluationDomain's small helpers (dusk-plonk 0.8.2 fft::domain, ref:Cargo.toml:19) -- what a patch
 * at the quotient_poly::compute / linearisation_poly level calls besides the transforms.  The plain forms are pure host
 * arithmetic on host buffers and need no context; the _dev forms fill a device vector of 2^log_n canonical Fr (Montgomery
 * limbs) with the library's kernels (pm_fr_powers_dev, pm_fr_vec_op_dev, pm_fr_batch_inverse_dev) and return when it is
 * complete.  PM_ERR_DOMAIN_TOO_LARGE when log_n >= 32.
 *   evaluate_vanishing_polynomial(tau)             = tau^size - 1
 *   compute_vanishing_poly_over_coset(poly_degree)   out[i] = (g w^i)^poly_degree - 1, g = GENERATOR = 7, w = group_gen;
 *                                                    upstream asserts size > poly_degree: PM_ERR_BAD_ARG otherwise
 *   evaluate_all_lagrange_coefficients(tau)          out[i] = L_i(tau) = (tau^size - 1) / size * w^i / (tau - w^i);
 *                                                    tau = w^j in the domain: the indicator vector of j */
int pm_domain_evaluate_vanishing_polynomial(uint32_t log_n, const uint64_t tau[4], uint64_t out[4]);
int pm_domain_vanishing_poly_over_coset(uint32_t log_n, uint64_t poly_degree, uint64_t* out);
int pm_domain_vanishing_poly_over_coset_dev(pm_ctx* ctx, uint32_t log_n, uint64_t poly_degree, void* d_out, void* hip_stream);
int pm_domain_evaluate_all_lagrange_coefficients(uint32_t log_n, const uint64_t tau[4], uint64_t* out);
int pm_domain_evaluate_all_lagrange_coefficients_dev(pm_ctx* ctx, uint32_t log_n, const uint64_t tau[4], void* d_out,
                                                     void* hip_stream);

/* Build (and cache on the device) the twiddle tables of the 2^log_n domain.  Optional: the
 * first transform of a size does it implicitly. */
int pm_domain_prepare(pm_ctx* ctx, uint32_t log_n);

/* out[0 .. 2^log_n) = transform of in[0 .. in_len) zero-padded to 2^log_n (the reference's
 * `resize(size, zero)`).  Natural order in and out, canonical Montgomery limbs.
 * in == out is allowed (the *_in_place forms).  Host pointers. */
int pm_fr_ntt(pm_ctx* ctx, const uint64_t* in, size_t in_len, uint64_t* out, uint32_t log_n,
              uint32_t flags);

/* `batch` independent transforms of the same size (a prover round issues 4-6 of them).
 * Vector b starts at in + 4*b*in_stride / out + 4*b*out_stride (strides in Fr elements). */
int pm_fr_ntt_batch(pm_ctx* ctx, const uint64_t* in, size_t in_len, size_t in_stride,
                    uint64_t* out, size_t out_stride, uint32_t log_n, uint32_t batch,
                    uint32_t flags);

/* Same transform on DEVICE-resident data (addresses on ctx's GPU), asynchronous on
 * `hip_stream` (a hipStream_t; NULL = the context's own stream).  d_in == d_out allowed. */
int pm_fr_ntt_dev(pm_ctx* ctx, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
                  size_t out_stride, uint32_t log_n, uint32_t batch, uint32_t flags,
                  void* hip_stream);

/* One 2^log_n-point transform whose vector is block-distributed in natural order over `world` ranks (one
 * process and one context per GPU): rank r passes its N/world elements in d_inout
 * (DEVICE memory, transformed in place: the rank's block of x -> the same block of the result) and a scratch
 * buffer d_stage of 2 N/world elements (N/world when world == 1).  Four-step decomposition with three all-to-all
 * transposes; the sub-transforms are the library's own batched passes and the result equals pm_fr_ntt_dev's bit
 * for bit.  `exchange` performs the all-to-all of equal blocks (block p of d_send goes to rank p, block p of
 * d_recv comes from rank p; it is called with the library's stream idle and must return with the data in place;
 * non-zero = failure); NULL uses the context's RCCL communicator (pm_comm_init) on the context's stream.
 * world must be a power of two dividing both 2^(log_n / 2) and 2^(log_n - log_n / 2); 2 <= log_n <= 26.
 * Runs on the context's own stream.  Mirrors nothing upstream: dusk-plonk is single-device.
 * PM_NTT_TRANSPOSED saves the third all-to-all where the natural order is not needed (evaluations that are only
 * consumed pointwise, e.g. the quotient): a FORWARD transform then leaves X[k2 N1 + k1] at position k1 N2 + k2 of the
 * N1 x N2 row-major matrix (N1 = 2^(log_n / 2), rows block-distributed: rank r holds k1 in [r N1 / W, (r + 1) N1 / W)),
 * and an INVERSE transform takes exactly that layout and returns natural order. */
typedef int (*pm_alltoall_fn)(void* user, void* d_send, void* d_recv, size_t bytes_per_peer);
int pm_fr_ntt_fourstep_dev(pm_ctx* ctx, void* d_inout, void* d_stage, uint32_t log_n, uint32_t world,
                           uint32_t rank, uint32_t flags, pm_alltoall_fn exchange, void* user);
/* The same for `batch` vectors in ONE sequence of exchanges (a prover round transforms 4 - 20 polynomials of one size: the
 * all-to-all carries the batch's blocks together, so the number of exchange CALLS does not grow with the batch).
 * d_inout: batch x N/world elements, vector v's block at element v N/world; d_stage: 2 x batch x N/world elements
 * (1 x when world == 1).  d_halo (optional; forward + PM_NTT_TRANSPOSED only; NULL otherwise): batch x N2 elements, N2 =
 * 2^(log_n - log_n / 2): vector v's copy of the FIRST ROW OF THE NEXT RANK's part of the block-transposed result (row 0 for the
 * last rank), i.e. the values at index + 1 of this rank's last row -- what a consumer that reads X[k + 1] beside X[k] (the
 * quotient's z(w X) and next-row wires) needs from its neighbour; it travels inside the second all-to-all as one more column
 * per peer, no exchange of its own.  d_stage then holds 2 x batch x (N/world + N2) elements -- with world == 1:
 * batch x (N + N2): the halo row is packed there too.  The call cannot check the size of d_stage: the caller sizes it by this
 * rule (the library's own caller, csrc/prover_dist.hip.h, allocates 2 x 20 x (m + n2) elements once per key).  d_halo
 * doubles as the halo switch: a non-NULL d_halo on a call that is not forward + PM_NTT_TRANSPOSED is PM_ERR_BAD_ARG.
 * pm_fr_ntt_fourstep_dev is this call with batch = 1 and no halo. */
int pm_fr_ntt_fourstep_batch_dev(pm_ctx* ctx, void* d_inout, uint32_t batch, void* d_halo, void* d_stage, uint32_t log_n,
                                 uint32_t world, uint32_t rank, uint32_t flags, pm_alltoall_fn exchange, void* user);
/* Exchange counters of this context since the last reset: out[0] = all-to-all calls (one per transpose step with world > 1,
 * whichever transport carried it), out[1] = bytes this rank sent in them, out[2] = fixed-size message all-gathers of the
 * sharded / distributed prover, out[3] = transpose steps of the rank-split transforms whatever the world size (each is one
 * all-to-all call as soon as world > 1).  What a multi-GPU deployment pays per proof, countable on one GPU. */
int pm_comm_stats(pm_ctx* ctx, uint64_t out[4], int reset);

/* ---- KZG commit: G1 MSM ---------------------------------------------------------------- */

/* Upload n affine bases (CommitKey::powers_of_g) once; they stay resident in HBM.  A pm_bases is read-only after
 * pm_g1_bases_precompute and may be used by every context on its device at the same time (several proofs in flight over one
 * SRS); free it once, after the last call that uses it has returned. */
int pm_g1_bases_upload(pm_ctx* ctx, const uint64_t* xy, size_t n, pm_bases** out);
/* Optional, for a long-lived SRS: build the table of window multiples 2^(c w) * bases[i] in HBM
 * ((ceil(256/c) - 1) x 96 n extra bytes; about 8 MSMs of time, once).  Every later MSM on these
 * bases then needs one bucket set and no doublings: ~15 % faster.  window_bits 0 = default (log2 n, at most 20). */
int pm_g1_bases_precompute(pm_ctx* ctx, pm_bases* bases, uint32_t window_bits);
/* The same from affine points already in device memory (n x 96 bytes, the layout above). */
int pm_g1_bases_from_dev(pm_ctx* ctx, const void* d_xy, size_t n, pm_bases** out);
void pm_g1_bases_free(pm_ctx* ctx, pm_bases* bases);
size_t pm_g1_bases_len(const pm_bases* bases);

/* out = sum_{i<n} scalars[i] * bases[i]   (msm_variable_base(&bases[..n], scalars)).
 * Host scalars, n x 4 limbs.  n == 0 gives the identity. */
int pm_g1_msm(pm_ctx* ctx, const pm_bases* bases, size_t n, const uint64_t* scalars,
              uint32_t scalar_form, uint64_t out_xyz[18]);

/* Device-resident scalars; uses bases[offset .. offset+n) -- the shard primitive for
 * multi-GPU runs (each rank owns a slice of the points, then pm_g1_fold after all-gather).
 * Blocks until the 144-byte result is on the host. */
int pm_g1_msm_dev(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n,
                  const void* d_scalars, uint32_t scalar_form, uint64_t out_xyz[18],
                  void* hip_stream);

/* `batch` MSMs over the SAME bases in one pass (a prover round commits to 4-6 polynomials of one
 * size): scalar vector j starts at d_scalars + 32 * j * scalar_stride, result j at out_xyz + 18 j.
 * The sort and the additions scale with the batch, the latency-bound tail (bucket reduction, host
 * fold) is paid once.  batch <= 64. */
int pm_g1_msm_batch_dev(pm_ctx* ctx, const pm_bases* bases, size_t offset, size_t n, const void* d_scalars,
                        size_t scalar_stride, uint32_t batch, uint32_t scalar_form, uint64_t* out_xyz,
                        void* hip_stream);

/* out[i] = scalars[i] * base for device-resident scalars; affine results (n x 96 bytes, (0, 0) =
 * identity) in device memory.  SRS generation: with scalars = tau^i
 * (pm_fr_powers_dev) this is the powers_of_g of dusk_plonk's PublicParameters::setup, which upstream
 * computes on the CPU with a windowed single-base multiplication.  Blocks until done. */
int pm_g1_fixed_base_mul_dev(pm_ctx* ctx, const uint64_t base_xy[12], const void* d_scalars, size_t n,
                             uint32_t scalar_form, void* d_out_xy, void* hip_stream);

/* out = sum of k projective points (the group-law "all-reduce" after an all-gather). Host. */
int pm_g1_fold(const uint64_t* xyz_parts, size_t k, uint64_t out_xyz[18]);

/* Projective (homogeneous X/Z, Y/Z) -> affine; *is_identity = 1 and xy = 0 for infinity. */
int pm_g1_to_affine(const uint64_t xyz[18], uint64_t xy[12], int* is_identity);
/* k points ([k][18] -> [k][12], is_identity[k] optional) with one field inversion. */
int pm_g1_to_affine_batch(const uint64_t* xyz, size_t k, uint64_t* xy, int* is_identity);

/* ---- device-resident polynomial helpers (the callers either side of the hot path) ------- */
/* dusk_plonk::fft::{Polynomial, Evaluations} and
 * util::batch_inversion as used by the prover rounds between the NTT and MSM calls.  All
 * vectors are canonical Fr in DEVICE memory; `hip_stream` as in pm_fr_ntt_dev. */

/* Device memory for callers without their own HIP allocator (the Rust prover, the tests). */
int pm_dev_alloc(pm_ctx* ctx, size_t bytes, void** out);
int pm_dev_free(pm_ctx* ctx, void* p);
int pm_dev_upload(pm_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int pm_dev_download(pm_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* out[i] = a[i] (op) b[i]:  op 0 = add, 1 = sub, 2 = mul  (Evaluations / Polynomial `+ - *`).
 * b_len == 1 broadcasts one scalar (Polynomial * scalar).  In place allowed. */
int pm_fr_vec_op_dev(pm_ctx* ctx, int op, const void* d_a, const void* d_b, size_t b_len, void* d_out,
                     size_t n, void* hip_stream);
/* Polynomial::evaluate: out = sum_i coeffs[i] * point^i.  Blocks until `out` is on the host. */
int pm_fr_poly_evaluate_dev(pm_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t point[4],
                            uint64_t out[4], void* hip_stream);
/* The same for k <= PM_LINCOMB_MAX polynomials of n coefficients at ONE point (d_polys: host array of device
 * pointers, out: k x 4 limbs): one pass of kernels and a single host synchronisation -- the prover opens
 * 15 polynomials at z and 4 at z w. */
int pm_fr_poly_evaluate_many_dev(pm_ctx* ctx, uint32_t k, const void* const* d_polys, size_t n, const uint64_t point[4],
                                 uint64_t* out, void* hip_stream);
/* Polynomial::ruffini: quotient of coeffs(X) / (X - z), n-1 coefficients into d_out (the
 * remainder coeffs(z) is dropped, as upstream).  Not in place.  Asynchronous on the stream. */
int pm_fr_poly_ruffini_dev(pm_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t z[4], void* d_out,
                           void* hip_stream);
/* out[0] = 1, out[i] = in[0] * ... * in[i-1]: the grand-product accumulator z(X) of the permutation
 * argument (dusk_plonk::permutation).  In place allowed. */
int pm_fr_prefix_product_dev(pm_ctx* ctx, const void* d_in, size_t n, void* d_out, void* hip_stream);
/* util::batch_inversion: every non-zero element is replaced by its inverse, zeros stay zero. */
int pm_fr_batch_inverse_dev(pm_ctx* ctx, void* d_inout, size_t n, void* hip_stream);

/* ---- PLONK prover rounds (BASELINE.json configs[3]) ----------- */
/* The pointwise work of dusk_plonk::proof_system::{permutation, quotient_poly, linearisation_poly}
 * (dusk-plonk 0.8.2, ref:Cargo.toml:19) for a 4-wire arithmetic circuit
 *   q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c + PI = 0
 * with copy constraints over the cosets {1, k1, k2, k3} H, fused into one kernel per round so that
 * every polynomial stays in HBM between the NTT and MSM calls.  Vectors are canonical Fr in device
 * memory; challenges and constants are host Fr (Montgomery limbs). */

/* out[i] = scale * base^i, i < n: domain.elements(), the 4n-coset points g w^i, powers of a challenge. */
int pm_fr_powers_dev(pm_ctx* ctx, const uint64_t base[4], const uint64_t scale[4], size_t n, void* d_out,
                     void* hip_stream);

/* out = sum_j coeffs[j] * vecs[j], k <= PM_LINCOMB_MAX vectors of n elements (d_vecs: host array of
 * device pointers): the linearisation polynomial and the aggregated opening polynomial. */
#define PM_LINCOMB_MAX 16
int pm_fr_lincomb_dev(pm_ctx* ctx, uint32_t k, const void* const* d_vecs, const uint64_t* coeffs, size_t n,
                      void* d_out, void* hip_stream);

/* Permutation argument, per row i < n of the domain H (x_i = roots[i] = w^i):
 *   num[i] = prod_j (wires[j][i] + beta k_j x_i + gamma),   k_0 = 1
 *   den[i] = prod_j (wires[j][i] + beta sigmas[j][i] + gamma)
 * z = prefix_product(num / den) (pm_fr_batch_inverse_dev, pm_fr_vec_op_dev, pm_fr_prefix_product_dev). */
typedef struct pm_plonk_perm_args {
  const void* wires[4];   /* a, b, c, d evaluations on H */
  const void* sigmas[4];  /* sigma_1..4 evaluations on H */
  const void* roots;      /* w^i */
  uint64_t beta[4], gamma[4];
  uint64_t k[3][4];       /* coset representatives k1, k2, k3 (dusk: 7, 13, 17) */
} pm_plonk_perm_args;
int pm_plonk_perm_terms_dev(pm_ctx* ctx, const pm_plonk_perm_args* args, size_t n, void* d_num, void* d_den,
                            void* hip_stream);

/* Quotient numerator divided by Z_H, pointwise on the 4n coset (x_i = g w_4n^i, i < 4n):
 *   t[i] = zh_inv[i mod 4] * ( q_arith (q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c) + pi
 *          + q_range R + q_logic L + q_fixed_group_add F + q_variable_group_add V
 *          + alpha   ( z[i]   prod_j (w_j + beta k_j x_i + gamma)
 *                    - z[i+4] prod_j (w_j + beta sigma_j + gamma) )
 *          + alpha^2 ( z[i] - 1 ) l1[i] )
 * (index i+4 wraps: f(w X) on the 4n coset).  R, L, F, V are dusk-plonk 0.8's widget identities
 * (proof_system::widget::{range, logic, ecc::scalar_mul::fixed_base, ecc::curve_addition}) over the
 * row's wires and a_next, b_next, d_next, each times its separation challenge:
 *   R = s (delta(c - 4d) + k delta(b - 4c) + k^2 delta(a - 4b) + k^3 delta(d_next - 4a)),  k = s^2,
 *       delta(f) = f (f-1)(f-2)(f-3)
 *   L = s (delta(qa) + k delta(qb) + k^2 delta(qd) + k^3 (c - qa qb) + k^4 delta_xor_and(qa, qb, c, qd, q_c)),
 *       qa = a_next - 4a, qb = b_next - 4b, qd = d_next - 4d
 *   F = one round of JubJub fixed-base scalar multiplication (table point in q_l, q_r, q_c)
 *   V = JubJub addition (a, b) + (c, d) = (a_next, b_next), d_next = a d
 * coset_ifft of t gives the quotient polynomial.  A NULL q_arith means the constant 1, a NULL widget
 * selector means identically zero (that widget is not evaluated). */
typedef struct pm_plonk_quotient_args {
  const void* wires[4];
  const void* z;
  const void* q_m;
  const void* q_l;
  const void* q_r;
  const void* q_o;
  const void* q_4;
  const void* q_c;
  const void* pi;         /* public-input polynomial on the coset */
  const void* sigmas[4];
  const void* l1;         /* first Lagrange polynomial on the coset */
  const void* x;          /* the coset points g w_4n^i */
  uint64_t alpha[4], beta[4], gamma[4];
  uint64_t k[3][4];
  uint64_t zh_inv[4][4];  /* 1 / (x_i^n - 1), which only depends on i mod 4 */
  const void* q_arith;                /* NULL = 1 */
  const void* q_range;                /* NULL = 0, as the three below */
  const void* q_logic;
  const void* q_fixed_group_add;
  const void* q_variable_group_add;
  uint64_t range_sep[4], logic_sep[4], fixed_sep[4], var_sep[4];   /* the widgets' separation challenges */
} pm_plonk_quotient_args;
int pm_plonk_quotient_dev(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t n, void* d_out,
                          void* hip_stream);

/* ---- the whole prover behind one call ------------------------------------------------------------ */
/* Prover::{preprocess, prove_with_preprocessed} of dusk-plonk 0.8.2 (ref:Cargo.toml:19) sequenced inside the
 * library: 11 selector polynomials (the gate kinds the reference's gadgets emit: arithmetic, range, logic,
 * fixed-base scalar multiplication, variable-base curve addition -- ref:src/zk/gadgets.rs:34,37,40,88-91,211),
 * the 4-wire permutation, five rounds, 11 commitments, the 16 evaluations of dusk's Proof, a Merlin
 * transcript seeded with the verifier key.  Transcript (labels restated from the published 0.8 design,
 * unpinned): Transcript::new(label); q_m q_l q_r q_o q_c q_4 q_arith q_range q_logic q_variable_group_add
 * q_fixed_group_add left_sigma right_sigma out_sigma fourth_sigma; "dom-sep" = "circuit_size", "n";
 * w_l w_r w_o w_4; beta (re-absorbed), gamma; z; alpha and the four "... separation challenge"s; t_1..t_4;
 * z; the 17 "<name>_eval" scalars; "aggregate_witness" twice; w_z, w_z_w. */
#define PM_PLONK_SELECTORS 11   /* q_m q_l q_r q_o q_c q_4 q_arith q_range q_logic q_fixed_group_add q_variable_group_add */
#define PM_PLONK_VK_POINTS 15   /* the selector commitments in that order, then sigma_1..4 */
#define PM_PLONK_EVALS 17       /* a b c d a_next b_next d_next sigma_1 sigma_2 sigma_3 q_arith q_c q_l q_r z_next t r */
#define PM_PLONK_CHALLENGES 10  /* beta gamma alpha range_sep logic_sep fixed_sep var_sep z aw aw_shifted */
#define PM_PLONK_PROOF_BYTES 1040
/* prove flags.  DEFAULT (flags = 0): the public inputs (count, then position and value of every DECLARED input --
 * pass the positions of zero-valued inputs too) are absorbed into the transcript before round 1, so the statement is
 * bound to every challenge.  dusk-plonk 0.8.2 does not do that (its transcript never sees the public inputs):
 * PM_PLONK_UPSTREAM_TRANSCRIPT reproduces the restated upstream message sequence byte for byte and is what a drop-in
 * for dusk's Prover must pass ("Transcript modes").  The two modes agree up to and including the
 * four wire commitments and differ from `beta` on.  PM_PLONK_BIND_PUBLIC_INPUTS (r01/r02 spelling of the default) is
 * accepted and ignored; both flags together contradict each other: PM_ERR_BAD_ARG.
 * CHANGE r03: up to r02 flags = 0 meant the upstream sequence and binding was opt-in; a C caller that passed 0 and
 * needs the r02 bytes must now pass PM_PLONK_UPSTREAM_TRANSCRIPT. */
#define PM_PLONK_BIND_PUBLIC_INPUTS 1u
#define PM_PLONK_UPSTREAM_TRANSCRIPT 2u
typedef struct pm_prover_key pm_prover_key;
typedef struct pm_plonk_proof {
  uint64_t commitments[11][12];                  /* a b c d z t_1 t_2 t_3 t_4 w_z w_zw, affine, (0, 0) = identity */
  uint64_t evaluations[PM_PLONK_EVALS][4];       /* transcript order (above), Montgomery limbs; t is not part of dusk's Proof */
  uint64_t challenges[PM_PLONK_CHALLENGES][4];   /* recomputable from the transcript */
} pm_plonk_proof;
/* ProverKey: selectors[s] = n evaluations on H (host memory) in the order of PM_PLONK_SELECTORS, NULL =
 * identically zero; sigma_index[j n + i] = position (j' n + i') that follows wire j of gate i in its copy
 * cycle.  n a power of two >= 4.  Builds the coefficient forms, the 4n-coset forms the quotient needs
 * (trivial selectors -- q_arith = 1, a widget selector = 0 -- are recognised and skipped) and the per-proof
 * workspace (about 90 n x 32 bytes). */
int pm_plonk_preprocess(pm_ctx* ctx, const uint64_t* const selectors[PM_PLONK_SELECTORS], const int64_t* sigma_index,
                        size_t n, pm_prover_key** out);
void pm_plonk_key_free(pm_ctx* ctx, pm_prover_key* key);
/* Commits to the 15 polynomials of the key (the G1 part of dusk's VerifierKey) and seeds the key's base
 * transcript with them (Prover::preprocess).  Must run once before the first proof; commit_key: at least n
 * resident bases.  verifier_key_out: PM_PLONK_VK_POINTS affine points, may be NULL.  transcript_label NULL =
 * "plonk". */
int pm_plonk_key_commit(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key, const char* transcript_label,
                        uint64_t (*verifier_key_out)[12]);
int pm_plonk_verifier_key(const pm_prover_key* key, uint64_t (*out)[12]);
/* d_witness: device memory, [a | b | c | d] wire values, 4n Fr.  Public inputs: n_pi (position < n, value)
 * pairs in host memory -- the dense PI vector of dusk's construct_dense_pi_vec is built on the device.
 * commit_key: at least n resident bases (pm_g1_bases_upload / pm_g1_bases_from_dev, ideally with
 * pm_g1_bases_precompute).  One proof at a time per key (PM_ERR_BUSY otherwise). */
int pm_plonk_prove(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key, const void* d_witness,
                   const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                   pm_plonk_proof* out);
/* Proof::to_bytes: 11 x 48-byte compressed G1, then the 16 scalars of ProofEvaluations::to_bytes. */
int pm_plonk_proof_to_bytes(const pm_plonk_proof* proof, uint8_t out[PM_PLONK_PROOF_BYTES]);
/* ---- The prover with coefficient-range ownership end to end (SURVEY.md section 8e row 3 + 8f N5; configs[4]) ----------
 * pm_plonk_prove_sharded splits only the MSMs: every rank still holds every polynomial and repeats every transform.
 * Here rank r of `world` (a power of two, world^2 <= n) owns rows / coefficients [r n / world, (r + 1) n / world) of
 * every vector and nothing else -- workspace / world, no replicated transform: the size-n transforms run as
 * pm_fr_ntt_fourstep_batch_dev over the ranks (all the polynomials of a round in one sequence of all-to-alls), the 4n-coset
 * work as four size-n sub-coset transforms whose results stay in the block-transposed order (the quotient is pointwise; its
 * one next-row access reads the halo row the transform delivers), and prefix product, openings and Ruffini division are
 * local passes plus one all-gather of per-rank scalars each: 12 all-to-all calls and 8 all-gathers per proof
 * (pm_comm_stats).  The proof and the verifier key are bit-identical to pm_plonk_prove's on every rank.
 *   allgather: gathers ONE fixed-size message per rank (PM_COMM_MSG_WORDS u64 words, host memory; word 0 = a count,
 *              0 = the abort marker of a rank that gave up) into gathered[world][PM_COMM_MSG_WORDS]; returns 0 on success.
 *              NULL = the context's RCCL communicator (pm_comm_init).  8 per proof (the first an agreement on the
 *              arguments: public inputs, flags and size must be the same on every rank), 2 + 2 per key.
 *   alltoall:  the callback of pm_fr_ntt_fourstep_dev (device buffers); NULL = the communicator.
 * Inputs are the rank's slices: selector_slices[s] = m = n / world rows of selector s (NULL = identically zero on this
 * rank), sigma_index_slices = [4][m] (wire j of row rank m + i is followed by position sigma_index, a GLOBAL index
 * j' n + i'), commit_key_slice = powers [rank m, (rank + 1) m) of the commit key, d_witness_slices = [4][m] wire values
 * in device memory.  Public inputs are passed whole (positions are global) on every rank.  A rank whose arguments are
 * bad meets its peers in the agreement all-gather with the abort marker: every rank returns an error, nobody blocks.  A
 * rank that fails later (a HIP error between two all-to-alls) returns its error and sends the marker to the next
 * all-gather, but an all-to-all has no marker and no timeout: treat such an error as fatal for the group.
 * pm_plonk_dist_key_bytes: device memory this rank holds for the key and its per-proof workspace. */
#define PM_COMM_MSG_WORDS 289   /* 1 + 18 x PM_COMM_MAX_POINTS */
typedef int (*pm_allgather_fn)(void* user, const uint64_t* msg, uint64_t* gathered);
typedef struct pm_dist {
  uint32_t world, rank;
  pm_allgather_fn allgather;
  pm_alltoall_fn alltoall;
  void* user;
} pm_dist;
typedef struct pm_dist_key pm_dist_key;
int pm_plonk_preprocess_dist(pm_ctx* ctx, const pm_dist* dist, const uint64_t* const selector_slices[PM_PLONK_SELECTORS],
                             const int64_t* sigma_index_slices, size_t n, pm_dist_key** out);
void pm_plonk_dist_key_free(pm_ctx* ctx, pm_dist_key* key);
size_t pm_plonk_dist_key_bytes(const pm_dist_key* key);
int pm_plonk_key_commit_dist(pm_ctx* ctx, const pm_dist* dist, pm_dist_key* key, const pm_bases* commit_key_slice,
                             const char* transcript_label, uint64_t (*verifier_key_out)[12]);
int pm_plonk_prove_dist(pm_ctx* ctx, const pm_dist* dist, pm_dist_key* key, const pm_bases* commit_key_slice,
                        const void* d_witness_slices, const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi,
                        uint32_t flags, pm_plonk_proof* out);

/* Every label string of the transcript, in message order, as "key=label" lines (a static string).  The labels are
 * restated from the published dusk-plonk 0.8 design and are PARITY-UNPINNED; they live in ONE table (csrc/prover.hip,
 * namespace tl) that both the native prover and the Python verifier side read -- the single place to edit when
 * upstream vectors become available. */
const char* pm_plonk_transcript_labels(void);
/* The same with the SRS split over the GPUs of a node (BASELINE.json configs[4]): every rank calls this with
 * the same witness and its own slice of the commit key -- bases for coefficients [first_coefficient,
 * first_coefficient + pm_g1_bases_len(slice)) -- and `exchange` turns this rank's k partial points (k x 18
 * limbs, in place) into the sums over all ranks: an all-gather of k x 144 bytes followed by pm_g1_fold per
 * point.  exchange = NULL uses the context's RCCL communicator (pm_comm_init).  A callback is also entered
 * with k = 0 when this rank's local work failed: it must tell the peers (so that none blocks) and return
 * non-zero; a non-zero return aborts with PM_ERR_EXCHANGE. */
typedef int (*pm_exchange_fn)(void* user, uint64_t* xyz, uint32_t k);
int pm_plonk_key_commit_sharded(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key_slice,
                                size_t first_coefficient, pm_exchange_fn exchange, void* user,
                                const char* transcript_label, uint64_t (*verifier_key_out)[12]);
int pm_plonk_prove_sharded(pm_ctx* ctx, pm_prover_key* key, const pm_bases* commit_key_slice, size_t first_coefficient,
                           const void* d_witness, const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi,
                           uint32_t flags, pm_exchange_fn exchange, void* user, pm_plonk_proof* out);

/* ---- the exchange step inside the library (SURVEY.md sections 8b, 8e) ------------------------------- */
/* One process per GPU, one pm_ctx per process.  Rank 0 makes the id and hands its 128 bytes to the other ranks
 * out of band (MPI, a file, the host runtime's own broadcast); every rank then calls pm_comm_init, which is
 * collective (ncclCommInitRank over xGMI).  RCCL is bound at run time (dlopen of librccl.so.1).
 * pm_g1_allgather_fold: every rank passes its k <= PM_COMM_MAX_POINTS partial points (k x 18 limbs,
 * projective); on return each holds the sums over all ranks -- one ncclAllGather of a fixed 2312-byte message
 * per rank, then pm_g1_fold per point: an all-reduce under the group law (RCCL has no such reduction).  k = 0
 * is the abort marker of a rank whose local work failed: it still takes part, and the call returns
 * PM_ERR_EXCHANGE on every rank instead of leaving the peers blocked. */
#define PM_COMM_ID_BYTES 128
#define PM_COMM_MAX_POINTS 16
int pm_comm_unique_id(uint8_t id[PM_COMM_ID_BYTES]);
int pm_comm_init(pm_ctx* ctx, const uint8_t id[PM_COMM_ID_BYTES], int rank, int world);
int pm_comm_destroy(pm_ctx* ctx);
int pm_comm_info(const pm_ctx* ctx, int* rank, int* world);
int pm_g1_allgather_fold(pm_ctx* ctx, uint64_t* xyz, uint32_t k);
/* test hook: the fold step on `world` messages of 1 + 16 x 18 words ([count | points]) given by the caller */
int pm_test_fold_gathered(const uint64_t* msgs, int world, uint32_t k, uint64_t* out_xyz);
/* A dead peer INSIDE an exchange: pm_set_option(ctx, "comm_timeout_ms", T) with T > 0 (default 0 = off) puts every exchange
 * on the context's communicator -- the message all-gather and the all-to-all of the rank-split transform -- under a
 * deadline that covers both the enqueue and the wait for its completion (with the option on, an all-to-all is waited
 * for where it is issued).  When T ms pass without completion the context's watch thread calls ncclCommAbort, the
 * blocked call returns, the entry point returns PM_ERR_EXCHANGE (pm_last_error names the exchange and the option) and
 * the communicator is DEAD: every later exchange on it returns PM_ERR_EXCHANGE at once, until pm_comm_destroy +
 * pm_comm_init on every rank.  Choose T well above the slowest legitimate exchange (the peers of a 2^24-gate proof
 * arrive at their all-gathers tens of milliseconds apart; a first all-to-all also sets up connections).  The callback
 * transports have their own containment (a broken barrier).  Test hook, no GPU and no RCCL: the same deadline logic on a
 * stub table whose all-gather blocks until its communicator is aborted (peer_answers = 0) or returns at once (1):
 * first_rc / second_rc = two exchanges in a row, elapsed_ms = the first one's duration, aborts = ncclCommAbort calls. */
int pm_test_comm_deadline(long timeout_ms, int peer_answers, int* first_rc, long* elapsed_ms, int* second_rc, int* aborts,
                          char* err_out, size_t err_cap);

/* Keccak-f[1600] on a 200-byte state (host; the permutation under the Merlin / STROBE-128 transcript
 * the prover derives its challenges from -- merlin is a dependency of dusk-plonk, ref:Cargo.toml:19). */
void pm_keccak_f1600(uint8_t state[200]);

/* ---- introspection / tuning (not needed by the prover) --------------------------------- */

/* Number of kernel launches and the Stockham radices the library will use for 2^log_n. */
int pm_ntt_plan(uint32_t log_n, uint32_t radix_log2[4], uint32_t* n_passes);
/* Override tunables: "msm_window_bits", "msm_chunk", "msm_lb", "msm_max_pairs", "msm_pipeline" (1 = a batched MSM as
 * pipelined pieces on two streams; off: it measured slower), "ntt_tile_log", "ntt_radix", "ntt_max_radix", "ntt_xcd",
 * "ntt_direct_tw", "ntt_pipeline" (0 = no copy / compute overlap in pm_fr_ntt_batch), "poly_lookback" (prefix product in one
 * pass: 0 never, 1 while all tiles are resident -- the default --, 2 always).  PM_ERR_BAD_ARG if unknown. */
int pm_set_option(pm_ctx* ctx, const char* key, long value);
/* Opt-in per-kernel timing with hipEvents recorded on the launch stream (bench.py's roofline
 * leg).  pm_profile_read writes lines "<kernel> <launches> <total_ms>\n" into buf. */
int pm_profile_enable(pm_ctx* ctx, int on);
/* Restrict the timers to one kernel name (as printed by pm_profile_read), NULL = all: an event pair
 * costs ~5 us of launch stream time, so a timed region that needs one kernel's duration asks for that one. */
int pm_profile_select(pm_ctx* ctx, const char* kernel_name);
int pm_profile_read(pm_ctx* ctx, char* buf, size_t cap);
/* Elementwise field kernels used by the parity tests: op 0 = Fr mul, 1 = Fr add, 2 = Fr sub,
 * 3 = Fp mul, 4 = Fp add, 5 = Fp sub, 6 = Fr inverse of a, 7 = Fp inverse of a (b ignored, may be NULL; 0 -> 0).  Host pointers,
 * n elements. */
int pm_test_field_op(pm_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out,
                     size_t n);
/* Pure host, no context: the host-side field arithmetic behind the MSM fold and the prover's challenge scalars
 * (csrc/host_field.h).  op 0 = Fr product, 1 = Fp product, 2 = Fr inverse (binary extended Euclid), 3 = Fp inverse, 4 / 5 =
 * the same inverses by exponentiation; Montgomery form in and out, n elements; b is ignored by the inversions. */
int pm_test_host_field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* Pure host, no device: the library's sizing pass for one MSM piece of this shape -- out[4] = {its return code, device
 * workspace bytes, pinned host bytes, (digit, point) pairs at most}. */
int pm_test_msm_sizing(size_t n, uint32_t batch, long window_bits, uint32_t table_window_bits, uint32_t num_cus,
                       uint64_t out[4]);
/* Pure host, no context: the pass plan of a transform of 2^log_n points (tunables 0 = a fresh context's defaults) --
 * out[20] = {passes, 4 x {log2 radix, log2 columns per tile, threads per workgroup, LDS bytes}, mask of passes that have
 * a kernel, log2 group size of the blocked intermediate layout, 0}. */
int pm_test_ntt_plan(uint32_t log_n, uint32_t batch, long tile_log, long max_radix, long radix, uint32_t num_cus,
                     uint32_t out[20]);
/* Pure host, no context: the bucket-fill layout (csrc/msm_sort.hip.h) an MSM of this shape would run with -- out[16] =
 * {window bits, windows, bucket sets, bucket bits, partition bits, local bits, partitions per set, bins, partitions,
 * scalars per scatter tile, tiles, LDS bytes scatter, LDS bytes local sort, finer low partitions, their extra bits, 0} --
 * and, when the three arrays are given (2^bucket-bits / partitions-per-set entries), the partition of every bucket and
 * every partition's first bucket and log2 width.  table_window_bits 0 = bases without a window table. */
int pm_test_msm_geometry(size_t n, long window_bits, uint32_t table_window_bits, uint32_t batch, uint32_t out[16],
                         uint32_t* part_of_bucket, uint32_t* first_bucket, uint32_t* width_bits);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_MI355X_H */
