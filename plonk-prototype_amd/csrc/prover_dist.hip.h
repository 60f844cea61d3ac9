// The prover with COEFFICIENT-RANGE OWNERSHIP END TO END (SURVEY.md section 8e row 3 + section 8f N5; BASELINE.json configs[4]):
// included at the end of prover.hip (it shares the transcript, the widget scalars and the label table with the
// single-GPU prover -- both must produce the same bytes).
//
// pm_plonk_prove_sharded replicates every transform, the quotient and the openings on all ranks and only splits the MSMs;
// at 2^24 gates that is ~70 GB of workspace per rank and an Amdahl floor of ~16 % of a proof.  Here rank r of W owns
//     rows    [r m, (r + 1) m)  of every evaluation vector over H          (m = n / W)
//     coefficients [r m, (r + 1) m) of every polynomial -- the range its slice of the commit key commits to
//     points  k in [r m, (r + 1) m) of each of the four sub-cosets  g w4^s H  (s = 0 .. 3) of the 4n coset
// and nothing else: workspace / W, no replicated transform.  What the single-GPU prover does with a size-4n coset
// transform of a degree < n polynomial is four size-n transforms here (the 4n coset g <w4> is the union of the sub-cosets
// g_s H, g_s = g w4^s: f(g_s w^k) = NTT_n(f_i g_s^i)[k]) over the ranks (the all-to-all of section 7.5).
// r05: the exchanges are cut by construction.  All polynomials of a round go through ONE batched transform
// (pm_fr_ntt_fourstep_batch_dev: the all-to-all carries the whole batch), everything on the coset stays in the BLOCK-
// TRANSPOSED order the forward transform leaves after its second all-to-all (PM_NTT_TRANSPOSED: four planes [s][m] per
// polynomial, position k1_local N2 + k2 <-> k = k2 N1 + k1) -- the quotient is pointwise and does not care -- and the one thing
// that is not pointwise, z(w X) and the next-row wires = index k + 1 = one row down, comes from a HALO ROW that the same
// second all-to-all delivers as one more column per peer.  The quotient's coefficients come back through ONE batched
// inverse transform that takes the transposed order (two all-to-alls) and a pointwise 4-point inverse DFT.  Per proof:
// 3 + 2 (wires and public inputs: to coefficients, onto the coset) + 3 + 2 (z) + 2 (quotient) = 12 all-to-all calls, where
// r04 made 3 per transform x 34 transforms = 102 (pm_comm_stats counts them).  Prefix product, openings and Ruffini division
// are local passes plus one fixed-size all-gather of per-rank scalars each.  Every exchange besides the all-to-all is
// the SAME 2312-byte message all-gather as the sharded prover's (count word 0 = abort marker): 8 per proof, the first an
// agreement on the arguments: a rank whose arguments are bad meets its peers there with the marker.  A rank that fails
// LATER, between two all-to-alls -- a HIP error, an allocation, a dead process -- returns its error (or nothing); its peers
// see the marker at their next all-gather but cannot see it inside an all-to-all.  Containment (r06): with option
// comm_timeout_ms set, the peers' pending exchange is ended by ncclCommAbort after the deadline, their call returns
// PM_ERR_EXCHANGE and their communicator is marked dead (comm.hip); without the option they block until the host's own
// watchdog ends them.
// Results are bit-identical to pm_plonk_prove (tests/test_gpu_dist_prover_n5.py).

static constexpr uint32_t DIST_MAX_VECS = 20;   // vectors of one batched transform (5 polynomials x 4 sub-cosets): sizes the stage buffer
struct pm_dist_key {
  size_t n = 0, m = 0, lo = 0, n2 = 0;   // n2: row length of the block-transposed order, 2^(log_n - log_n / 2)
  uint32_t log_n = 0, world = 1, rank = 0;
  HFr omega, k[3], zh_inv[4], gs[4];
  // device arrays (element counts in units of m = n / world); "coset" arrays are four planes [s][m], block-transposed order
  void *roots = nullptr /* m */, *gs_pow = nullptr /* 4m: g_s^i */, *gs_inv_pow = nullptr /* 4m */,
       *sel_coeffs = nullptr /* 11m */, *sigma_evals = nullptr /* 4m */, *sigma_coeffs = nullptr /* 4m */,
       *sigma_coset = nullptr /* 16m */, *lx_coset = nullptr /* 8m: L_1, then the coset points x */, *sel_coset_all = nullptr;
  void* sel_coset[NSEL] = {};   // into sel_coset_all
  bool sel_zero[NSEL] = {};
  bool arith_is_one = false;
  // per-proof workspace: coeffs [a b c d pi z] 6m | num m | den m | coset_a 20m + 20 n2 halo rows | coset_z 4m + 4 n2 | tq 4m |
  // t 4m | r m | agg 2m | wit 2m | stage 2 x 20 (m + n2) | tmp m | one scalar
  void *coeffs = nullptr, *num = nullptr, *den = nullptr, *coset_a = nullptr, *halo_a = nullptr, *coset_z = nullptr,
       *halo_z = nullptr, *t = nullptr, *tq = nullptr, *r = nullptr, *agg = nullptr, *wit = nullptr, *stage = nullptr,
       *tmp = nullptr, *scalar = nullptr;
  size_t device_bytes = 0;   // what this rank holds for the key and its workspace
  bool committed = false;
  u64 vk[NSEL + 4][12] = {};
  Transcript base{std::string("plonk")};
  std::atomic<bool> busy{false};
};

namespace {
struct Dist {
  pm_dist d;
  int expect = 0;          // message all-gathers the call makes on every rank
  int done = 0;
  bool aborted = false;
};
// One all-gather of the fixed-size message; msg[0] = count (0 = abort marker).  gathered: world x COMM_MSG_WORDS.
int dist_exchange(pm_ctx* ctx, Dist& D, const std::vector<u64>& msg, std::vector<u64>& gathered) {
  gathered.assign(pm::COMM_MSG_WORDS * (size_t)D.d.world, 0);
  int rc;
  if (D.d.allgather) {
    rc = D.d.allgather(D.d.user, msg.data(), gathered.data()) != 0 ? PM_ERR_EXCHANGE : PM_OK;
  } else {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->comm || ctx->comm_world != (int)D.d.world || ctx->comm_rank != (int)D.d.rank)
      rc = pm::set_err(ctx, PM_ERR_EXCHANGE, "no all-gather callback and no matching communicator (pm_comm_init)");
    else
      rc = pm::comm_allgather_msg(ctx, msg.data(), gathered.data());
  }
  ++D.done;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ++ctx->stat_allgather_calls;
  }
  if (rc == PM_OK && msg[0] == 0) rc = PM_ERR_EXCHANGE;
  for (uint32_t r = 0; r < D.d.world && rc == PM_OK; ++r)
    if (gathered[r * pm::COMM_MSG_WORDS] != msg[0]) rc = PM_ERR_EXCHANGE;   // a peer gave up, or the ranks are out of step
  if (rc != PM_OK) D.aborted = true;
  return rc;
}
int dist_leave(pm_ctx* ctx, Dist& D, int rc) {
  if (rc != PM_OK && ctx && !D.aborted && D.done < D.expect) {
    std::vector<u64> msg(pm::COMM_MSG_WORDS, 0), g;
    (void)dist_exchange(ctx, D, msg, g);
  }
  return rc;
}
// k <= 16 partial points -> their sums over the ranks
int dist_points(pm_ctx* ctx, Dist& D, u64* xyz, uint32_t k) {
  std::vector<u64> msg(pm::COMM_MSG_WORDS, 0), g;
  msg[0] = k;
  memcpy(msg.data() + 1, xyz, 144 * (size_t)k);
  PK_TRY(dist_exchange(ctx, D, msg, g));
  return pm::fold_gathered(g.data(), (int)D.d.world, k, xyz);
}
// k <= 72 scalars per rank -> all[rank][j]
int dist_scalars(pm_ctx* ctx, Dist& D, const HFr* vals, uint32_t k, std::vector<HFr>& all) {
  std::vector<u64> msg(pm::COMM_MSG_WORDS, 0), g;
  msg[0] = k;
  for (uint32_t j = 0; j < k; ++j) memcpy(msg.data() + 1 + 4 * j, vals[j].l, 32);
  PK_TRY(dist_exchange(ctx, D, msg, g));
  all.resize((size_t)D.d.world * k);
  for (uint32_t r = 0; r < D.d.world; ++r)
    for (uint32_t j = 0; j < k; ++j) memcpy(all[(size_t)r * k + j].l, g.data() + r * pm::COMM_MSG_WORDS + 1 + 4 * j, 32);
  return PM_OK;
}
HFr fpow64(const HFr& a, u64 e) { return fpow(a, e); }

// `batch` size-n transforms over the ranks in ONE sequence of exchanges, in place on contiguous m-element blocks
int dist_ntt(pm_ctx* ctx, const Dist& D, const pm_dist_key* pk, void* d_blocks, uint32_t batch, void* d_halo, uint32_t flags) {
  if (batch > DIST_MAX_VECS) return PM_ERR_BAD_ARG;
  return pm_fr_ntt_fourstep_batch_dev(ctx, d_blocks, batch, d_halo, pk->stage, pk->log_n, D.d.world, D.d.rank, flags, D.d.alltoall,
                                      D.d.user);
}
// coefficients [lo, lo + m) of `count` <= 5 polynomials of degree < n  ->  their values on the rank's points of the four
// sub-cosets: planes [4 j + s][m] of d_planes in block-transposed order (+ the halo rows [4 j + s][n2], when asked for).
// One expansion kernel (f_i g_s^i), one batched transform: two all-to-alls whatever the count.
int dist_to_coset(pm_ctx* ctx, const Dist& D, const pm_dist_key* pk, const void* const* d_coeffs, uint32_t count, void* d_planes,
                  void* d_halo) {
  PK_TRY(pm::coset_expand(ctx, d_coeffs, count, pk->gs_pow, pk->m, d_planes));
  return dist_ntt(ctx, D, pk, d_planes, 4 * count, d_halo, PM_NTT_TRANSPOSED);
}
int upload_scalar(pm_ctx* ctx, const pm_dist_key* pk, const HFr& v) {
  PK_TRY(pm_sync(ctx));   // the previous user of the one-scalar buffer has finished
  return pm_dev_upload(ctx, pk->scalar, v.l, 32);
}
int commit_batch_dist(pm_ctx* ctx, Dist& D, const pm_dist_key* pk, const pm_bases* ck, const void* d, size_t n_coeffs,
                      uint32_t batch, u64 (*out_xy)[12]) {
  u64 xyz[16 * 18];
  if (batch > 16) return PM_ERR_BAD_ARG;
  const size_t hi = std::min(pk->lo + pk->m, n_coeffs);
  const size_t cnt = hi > pk->lo ? hi - pk->lo : 0;
  int rc = PM_OK;
  if (cnt > 0) {
    rc = pm_g1_msm_batch_dev(ctx, ck, 0, cnt, d, pk->m, batch, PM_SCALAR_MONTGOMERY, xyz, nullptr);
  } else {
    memset(xyz, 0, sizeof xyz);
    for (uint32_t b = 0; b < batch; ++b) memcpy(xyz + 18 * b + 6, pm::host::FP().one, 48);   // (0, 1, 0)
  }
  if (rc != PM_OK) return rc;   // dist_leave sends the abort marker
  PK_TRY(dist_points(ctx, D, xyz, batch));
  return pm_g1_to_affine_batch(xyz, batch, &out_xy[0][0], nullptr);
}
struct DistBusy {
  pm_dist_key* pk;
  bool ok;
  explicit DistBusy(pm_dist_key* k) : pk(k), ok(!k->busy.exchange(true)) {}
  ~DistBusy() {
    if (ok) pk->busy.store(false);
  }
};
int dist_check(const pm_dist* d, size_t n) {
  if (!d || d->world == 0 || (d->world & (d->world - 1)) || d->rank >= d->world) return PM_ERR_BAD_ARG;
  if (n < 4 || (n & (n - 1))) return PM_ERR_LENGTH;
  uint32_t lg = 0;
  while (((size_t)1 << lg) < n) ++lg;
  if (lg > 26) return PM_ERR_DOMAIN_TOO_LARGE;
  if (((size_t)1 << (lg / 2)) % d->world) return PM_ERR_BAD_ARG;   // the ranks must divide both factors of the size-n transform
  return PM_OK;
}
}  // namespace

extern "C" void pm_plonk_dist_key_free(pm_ctx* ctx, pm_dist_key* pk) {
  if (!pk) return;
  if (ctx) (void)pm_sync(ctx);
  for (void* p : {pk->roots, pk->gs_pow, pk->gs_inv_pow, pk->sel_coeffs, pk->sigma_evals, pk->sigma_coeffs, pk->sigma_coset,
                  pk->lx_coset, pk->sel_coset_all, pk->coeffs, pk->num, pk->den, pk->coset_a, pk->halo_a, pk->coset_z, pk->halo_z,
                  pk->t, pk->tq, pk->r, pk->agg, pk->wit, pk->stage, pk->tmp, pk->scalar})
    if (p && ctx) (void)pm_dev_free(ctx, p);
  delete pk;
}
extern "C" size_t pm_plonk_dist_key_bytes(const pm_dist_key* key) { return key ? key->device_bytes : 0; }

static int preprocess_dist_body(pm_ctx* ctx, Dist& D, pm_dist_key* pk, const uint64_t* const selector_slices[PM_PLONK_SELECTORS],
                                const int64_t* sigma_index_slices) {
  const size_t n = pk->n, m = pk->m, lo = pk->lo;
  const uint32_t lg = pk->log_n;
  u64 w[4], wi[4], si[4], w4[4];
  PK_TRY(pm_domain_info(lg, w, wi, si));
  PK_TRY(pm_domain_info(lg + 2, w4, wi, wi));
  pk->omega = get(w);
  const HFr n_inv = get(si), omega4 = get(w4), one = fone(), g = fr_u64(7), zero = pm::host::zero<4>();
  pk->k[0] = fr_u64(7);
  pk->k[1] = fr_u64(13);
  pk->k[2] = fr_u64(17);
  {
    HFr p = g;
    for (int s = 0; s < 4; ++s) {
      pk->gs[s] = p;
      p = fmul(p, omega4);
    }
  }
  // What can fail on ONE rank only is done before the first exchange, so that the failure travels as the abort marker
  // (ADVICE r04): the sigma indices of the slice (range, no repeats inside the slice; three power sums of the indices
  // go round in the agreement below and must be those of a permutation of 0 .. 4n - 1) and the fixed-size allocations.
  u64 sig_mom[3] = {0, 0, 0};
  {
    std::vector<uint8_t> seen((4 * n + 7) / 8, 0);
    for (size_t p = 0; p < 4 * m; ++p) {
      const int64_t q = sigma_index_slices[p];
      if (q < 0 || (size_t)q >= 4 * n) return pm::set_err(ctx, PM_ERR_BAD_ARG, "sigma index outside the circuit");
      if (seen[(size_t)q >> 3] & (1u << (q & 7))) return pm::set_err(ctx, PM_ERR_BAD_ARG, "sigma index repeated: not a permutation");
      seen[(size_t)q >> 3] |= (uint8_t)(1u << (q & 7));
      const u64 v = (u64)q;
      sig_mom[0] += v;
      sig_mom[1] += v * v;
      sig_mom[2] += v * v * v;
    }
  }
  struct Alloc { void** p; size_t elems; };
  auto alloc_all = [&](const std::vector<Alloc>& list) -> int {
    for (const Alloc& a : list) {
      PK_TRY(pm_dev_alloc(ctx, a.elems * 32, a.p));
      pk->device_bytes += a.elems * 32;
    }
    return PM_OK;
  };
  const size_t n2 = pk->n2;
  PK_TRY(alloc_all({{&pk->roots, m},          {&pk->gs_pow, 4 * m},      {&pk->gs_inv_pow, 4 * m},  {&pk->sel_coeffs, (size_t)NSEL * m},
                    {&pk->sigma_evals, 4 * m}, {&pk->sigma_coeffs, 4 * m}, {&pk->sigma_coset, 16 * m}, {&pk->lx_coset, 8 * m},
                    {&pk->coeffs, 6 * m},      {&pk->num, m},             {&pk->den, m},             {&pk->coset_a, 20 * m},
                    {&pk->halo_a, 20 * n2},    {&pk->coset_z, 4 * m},     {&pk->halo_z, 4 * n2},     {&pk->t, 4 * m},
                    {&pk->tq, 4 * m},          {&pk->r, m},               {&pk->agg, 2 * m},         {&pk->wit, 2 * m},
                    {&pk->stage, 2 * (size_t)DIST_MAX_VECS * (m + n2)},   {&pk->tmp, m},             {&pk->scalar, 1}}));
  // trivial selector polynomials: a GLOBAL property (every rank must take the same branches): local flags, one exchange
  {
    HFr flags[2] = {zero, zero};   // words: [selector non-zero mask, q_arith differs from one, rank, n], [sigma index power sums]
    u64 nz = 0, not_one = selector_slices[Q_ARITH] ? 0 : 1;
    for (int s = 0; s < NSEL; ++s)
      if (selector_slices[s])
        for (size_t i = 0; i < 4 * m; ++i)
          if (selector_slices[s][i]) {
            nz |= (u64)1 << s;
            break;
          }
    if (selector_slices[Q_ARITH])
      for (size_t i = 0; i < m && !not_one; ++i) not_one = memcmp(selector_slices[Q_ARITH] + 4 * i, one.l, 32) != 0;
    flags[0].l[0] = nz;
    flags[0].l[1] = not_one;
    flags[0].l[2] = D.d.rank;
    flags[0].l[3] = (u64)n;
    for (int k = 0; k < 3; ++k) flags[1].l[k] = sig_mom[k];
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, flags, 2, all));
    u64 nz_all = 0, not_one_all = 0, seen = 0, mom[3] = {0, 0, 0};
    for (uint32_t r = 0; r < D.d.world; ++r) {
      nz_all |= all[2 * r].l[0];
      not_one_all |= all[2 * r].l[1];
      if (all[2 * r].l[2] < 64) seen |= (u64)1 << all[2 * r].l[2];
      if (all[2 * r].l[3] != (u64)n) return pm::set_err(ctx, PM_ERR_LENGTH, "the ranks disagree on the circuit size");
      for (int k = 0; k < 3; ++k) mom[k] += all[2 * r + 1].l[k];
    }
    if (D.d.world <= 64 && seen != (D.d.world == 64 ? ~(u64)0 : (((u64)1 << D.d.world) - 1)))
      return pm::set_err(ctx, PM_ERR_BAD_ARG, "the ranks of the group are not 0 .. world - 1, each once");
    // sum q^k over 0 .. 4n - 1 (mod 2^64): a configuration guard against slices that overlap or leave gaps between the
    // ranks (inside a slice repeats were excluded above), not a proof that the union is a permutation
    u64 want[3] = {0, 0, 0};
    {
      const unsigned __int128 N = (unsigned __int128)4 * n;   // 4n <= 2^28
      const unsigned __int128 s1 = N * (N - 1) / 2;
      want[0] = (u64)s1;
      want[1] = (u64)((N - 1) * N * (2 * N - 1) / 6);
      want[2] = (u64)(s1 * s1);
    }
    for (int k = 0; k < 3; ++k)
      if (mom[k] != want[k]) return pm::set_err(ctx, PM_ERR_BAD_ARG, "the ranks' sigma indices are not a permutation of the 4n wire positions");
    for (int s = 0; s < NSEL; ++s) pk->sel_zero[s] = !((nz_all >> s) & 1);
    pk->arith_is_one = !not_one_all;
  }
  {
    // the allocations that depend on the flags, then the SECOND exchange: every rank says it holds everything it needs
    // before anyone enters the first all-to-all (which carries no abort marker and has no timeout)
    uint32_t count = 0;
    bool need[NSEL];
    for (int s = 0; s < NSEL; ++s) {
      need[s] = s <= Q_4 || (s == Q_ARITH ? !pk->arith_is_one : !pk->sel_zero[s]);
      count += need[s] ? 1u : 0u;
    }
    PK_TRY(alloc_all({{&pk->sel_coset_all, (size_t)count * 4 * m}}));   // one array: consecutive selectors transform as one batch
    count = 0;
    for (int s = 0; s < NSEL; ++s)
      if (need[s]) pk->sel_coset[s] = at(pk->sel_coset_all, (size_t)count++ * 4 * m);
    HFr ok = zero;
    ok.l[0] = 0x6f6b;
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, &ok, 1, all));
  }
  // this rank's domain points and the sub-coset powers g_s^i (i global)
  PK_TRY(pm_fr_powers_dev(ctx, pk->omega.l, fpow64(pk->omega, lo).l, m, pk->roots, nullptr));
  for (int s = 0; s < 4; ++s) {
    const HFr gi = finv(pk->gs[s]);
    PK_TRY(pm_fr_powers_dev(ctx, pk->gs[s].l, fpow64(pk->gs[s], lo).l, m, at(pk->gs_pow, s * m), nullptr));
    PK_TRY(pm_fr_powers_dev(ctx, gi.l, fpow64(gi, lo).l, m, at(pk->gs_inv_pow, s * m), nullptr));
  }
  // selectors: rows -> coefficient slices (ONE batched inverse transform) -> the coset forms the quotient kernel reads, five
  // polynomials per batched forward transform.  An identically-zero selector with a coset array goes through like the
  // others: the transform of zeros is zeros (the array was left as allocated).
  for (int s = 0; s < NSEL; ++s) {
    void* dst = at(pk->sel_coeffs, s * m);
    if (pk->sel_zero[s] || !selector_slices[s]) PK_TRY(pm_fr_powers_dev(ctx, zero.l, zero.l, m, dst, nullptr));
    else PK_TRY(pm_dev_upload(ctx, dst, selector_slices[s], m * 32));
  }
  PK_TRY(dist_ntt(ctx, D, pk, pk->sel_coeffs, NSEL, nullptr, PM_NTT_INVERSE));
  {
    const void* src[5];
    void* first = nullptr;
    uint32_t cnt = 0;
    for (int s = 0; s <= NSEL; ++s) {
      if (s < NSEL && pk->sel_coset[s]) {
        if (!cnt) first = pk->sel_coset[s];
        src[cnt++] = at(pk->sel_coeffs, s * m);
      }
      if (cnt == 5 || (s == NSEL && cnt)) {
        PK_TRY(dist_to_coset(ctx, D, pk, src, cnt, first, nullptr));
        cnt = 0;
      }
    }
  }
  // sigma_j(w^i) = k_j' w^i' for this rank's rows: gathered on the device from two-level tables of w (2 sqrt(n) entries)
  {
    u64 kk[3][4];
    for (int j = 0; j < 3; ++j) put(kk[j], pk->k[j]);
    PK_TRY(pm::sigma_evals_from_index(ctx, sigma_index_slices, 4 * m, lg, pk->omega.l, kk, pk->sigma_evals));
  }
  PM_HIP(ctx, hipMemcpyAsync(pk->sigma_coeffs, pk->sigma_evals, 4 * m * 32, hipMemcpyDeviceToDevice, ctx->stream));
  PK_TRY(dist_ntt(ctx, D, pk, pk->sigma_coeffs, 4, nullptr, PM_NTT_INVERSE));
  {
    const void* src[4];
    for (int j = 0; j < 4; ++j) src[j] = at(pk->sigma_coeffs, j * m);
    PK_TRY(dist_to_coset(ctx, D, pk, src, 4, pk->sigma_coset, nullptr));
  }
  // L_1 = (1/n) sum X^i and the polynomial X itself on the coset: the second gives the coset points x = g_s w^k in exactly
  // the order every other coset array has
  PK_TRY(pm_fr_powers_dev(ctx, one.l, n_inv.l, m, pk->num, nullptr));
  PK_TRY(pm_fr_powers_dev(ctx, zero.l, zero.l, m, pk->den, nullptr));
  if (lo == 0) PK_TRY(pm_dev_upload(ctx, at(pk->den, 1), one.l, 32));
  {
    const void* src[2] = {pk->num, pk->den};
    PK_TRY(dist_to_coset(ctx, D, pk, src, 2, pk->lx_coset, nullptr));
  }
  PK_TRY(pm_sync(ctx));
  {
    const HFr gn = fpow(g, n), i4 = fpow(omega4, n);
    HFr p = one;
    for (int k = 0; k < 4; ++k) {
      pk->zh_inv[k] = finv(fsub(fmul(gn, p), one));
      p = fmul(p, i4);
    }
  }
  return PM_OK;
}

extern "C" int pm_plonk_preprocess_dist(pm_ctx* ctx, const pm_dist* dist, const uint64_t* const selector_slices[PM_PLONK_SELECTORS],
                                        const int64_t* sigma_index_slices, size_t n, pm_dist_key** out) {
  if (!ctx || !dist || !selector_slices || !sigma_index_slices || !out) return PM_ERR_BAD_ARG;
  *out = nullptr;
  Dist D;
  D.d = *dist;
  D.expect = 2;
  int rc = dist_check(dist, n);
  if (rc) return rc;   // a malformed group or size is the same on every rank: nothing to tell the peers
  pm_dist_key* pk = new pm_dist_key();
  pk->n = n;
  pk->world = dist->world;
  pk->rank = dist->rank;
  pk->m = n / dist->world;
  pk->lo = pk->m * dist->rank;
  while (((size_t)1 << pk->log_n) < n) ++pk->log_n;
  pk->n2 = (size_t)1 << (pk->log_n - pk->log_n / 2);
  rc = preprocess_dist_body(ctx, D, pk, selector_slices, sigma_index_slices);
  rc = dist_leave(ctx, D, rc);
  if (rc) {
    pm_plonk_dist_key_free(ctx, pk);
    return rc;
  }
  *out = pk;
  return PM_OK;
}

extern "C" int pm_plonk_key_commit_dist(pm_ctx* ctx, const pm_dist* dist, pm_dist_key* pk, const pm_bases* ck_slice,
                                        const char* transcript_label, uint64_t (*vk_out)[12]) {
  if (!ctx || !dist || !pk || !ck_slice) return PM_ERR_BAD_ARG;
  Dist D;
  D.d = *dist;
  D.expect = 2;
  auto body = [&]() -> int {
    if (dist->world != pk->world || dist->rank != pk->rank) return PM_ERR_BAD_ARG;
    if (pm_g1_bases_len(ck_slice) < pk->m) return PM_ERR_LENGTH;   // powers [rank m, (rank + 1) m) of the commit key
    PK_TRY(commit_batch_dist(ctx, D, pk, ck_slice, pk->sel_coeffs, pk->n, NSEL, &pk->vk[0]));
    PK_TRY(commit_batch_dist(ctx, D, pk, ck_slice, pk->sigma_coeffs, pk->n, 4, &pk->vk[NSEL]));
    Transcript ts(transcript_label ? transcript_label : tl::PROTOCOL);
    for (int i = 0; i < NSEL; ++i) ts.append_commitment(SEL_LABELS[i], pk->vk[SEL_SEED_ORDER[i]]);
    for (int j = 0; j < 4; ++j) ts.append_commitment(SIGMA_LABELS[j], pk->vk[NSEL + j]);
    ts.append(tl::DOM_SEP, (const uint8_t*)tl::DOM_SEP_VALUE, strlen(tl::DOM_SEP_VALUE));
    ts.append_u64(tl::CIRCUIT_SIZE, pk->n);
    pk->base = ts;
    pk->committed = true;
    if (vk_out) memcpy(vk_out, pk->vk, sizeof pk->vk);
    return PM_OK;
  };
  return dist_leave(ctx, D, body());
}

static int prove_dist_body(pm_ctx* ctx, Dist& D, pm_dist_key* pk, const pm_bases* ck, const void* d_witness,
                           const uint64_t* pi_positions, const uint64_t* pi_values, size_t n_pi, uint32_t flags,
                           pm_plonk_proof* out) {
  if (!pk || !ck || !d_witness || !out) return PM_ERR_BAD_ARG;
  if (D.d.world != pk->world || D.d.rank != pk->rank) return PM_ERR_BAD_ARG;
  if (n_pi && (!pi_positions || !pi_values)) return PM_ERR_BAD_ARG;
  if (flags & ~(PM_PLONK_BIND_PUBLIC_INPUTS | PM_PLONK_UPSTREAM_TRANSCRIPT)) return PM_ERR_BAD_ARG;
  if (flags == (PM_PLONK_BIND_PUBLIC_INPUTS | PM_PLONK_UPSTREAM_TRANSCRIPT)) return PM_ERR_BAD_ARG;
  if (!pk->committed) return PM_ERR_BAD_ARG;
  DistBusy guard(pk);
  if (!guard.ok) return PM_ERR_BUSY;
  const size_t n = pk->n, m = pk->m, lo = pk->lo;
  const uint32_t W = pk->world, rk = pk->rank;
  if (pm_g1_bases_len(ck) < m) return PM_ERR_LENGTH;
  for (size_t i = 0; i < n_pi; ++i)
    if (pi_positions[i] >= n) return PM_ERR_LENGTH;
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  {
    // agreement: every rank has checked its arguments by now -- a rank whose check failed returns above and dist_leave sends
    // the abort marker HERE, before its peers enter the first all-to-all (ncclSend / ncclRecv carry no marker and have no
    // timeout).  The message also pins the statement: every rank must have been given the same public inputs and flags.
    HFr sum = pm::host::zero<4>();
    u64 h = 0x9e3779b97f4a7c15ULL ^ flags ^ ((u64)n_pi << 8);
    for (size_t i = 0; i < n_pi; ++i) {
      h = (h ^ pi_positions[i]) * 0xbf58476d1ce4e5b9ULL;
      for (int l = 0; l < 4; ++l) h = (h ^ pi_values[4 * i + l]) * 0x94d049bb133111ebULL + (h >> 29);
    }
    sum.l[0] = h;
    sum.l[1] = n;
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, &sum, 1, all));
    for (uint32_t r = 0; r < W; ++r)
      if (all[r].l[0] != h || all[r].l[1] != (u64)n)
        return pm::set_err(ctx, PM_ERR_BAD_ARG, "the ranks were given different public inputs, flags or circuit sizes");
  }
  Transcript ts = pk->base;
  if (!(flags & PM_PLONK_UPSTREAM_TRANSCRIPT)) {
    ts.append_u64(tl::PI_LEN, n_pi);
    for (size_t i = 0; i < n_pi; ++i) {
      ts.append_u64(tl::PI_POS, pi_positions[i]);
      ts.append_scalar(tl::PI_VALUE, get(pi_values + 4 * i));
    }
  }
  const HFr one = fone();
  const size_t n2 = pk->n2;
  auto coset = [&](int j) { return at(pk->coset_a, 4 * m * j); };   // planes of a, b, c, d (j < 4) and the public inputs (4)
  // ---- round 1 --------------------------------------------------------------------------------
  // wires and public inputs: ONE batched inverse transform (natural order: the coefficient slices are what the commit key
  // slice commits to), then ONE batched forward transform onto the four sub-cosets (20 vectors, halo rows with it)
  void *pi_coeffs = at(pk->coeffs, 4 * m), *z_coeffs = at(pk->coeffs, 5 * m);
  PM_HIP(ctx, hipMemcpyAsync(pk->coeffs, d_witness, 4 * m * 32, hipMemcpyDeviceToDevice, st));
  PM_HIP(ctx, hipMemsetAsync(pi_coeffs, 0, m * 32, st));
  for (size_t i = 0; i < n_pi; ++i)   // a repeated position keeps its last value (stream order)
    if (pi_positions[i] >= lo && pi_positions[i] < lo + m)
      PM_HIP(ctx, hipMemcpyAsync(at(pi_coeffs, pi_positions[i] - lo), pi_values + 4 * i, 32, hipMemcpyHostToDevice, st));
  PM_HIP(ctx, hipStreamSynchronize(st));   // pi_values is the caller's memory
  PK_TRY(dist_ntt(ctx, D, pk, pk->coeffs, 5, nullptr, PM_NTT_INVERSE));
  {
    const void* src[5];
    for (int j = 0; j < 5; ++j) src[j] = at(pk->coeffs, j * m);
    PK_TRY(dist_to_coset(ctx, D, pk, src, 5, pk->coset_a, pk->halo_a));
  }
  PK_TRY(commit_batch_dist(ctx, D, pk, ck, pk->coeffs, n, 4, &out->commitments[0]));
  for (int j = 0; j < 4; ++j) ts.append_commitment(tl::WIRES[j], out->commitments[j]);
  // ---- round 2 --------------------------------------------------------------------------------
  const HFr beta = ts.challenge_scalar(tl::BETA);
  ts.append_scalar(tl::BETA, beta);
  const HFr gamma = ts.challenge_scalar(tl::GAMMA);
  pm_plonk_perm_args pa;
  memset(&pa, 0, sizeof pa);
  for (int j = 0; j < 4; ++j) {
    pa.wires[j] = at((void*)d_witness, j * m);
    pa.sigmas[j] = at(pk->sigma_evals, j * m);
  }
  pa.roots = pk->roots;
  put(pa.beta, beta);
  put(pa.gamma, gamma);
  for (int j = 0; j < 3; ++j) put(pa.k[j], pk->k[j]);
  PK_TRY(pm_plonk_perm_terms_dev(ctx, &pa, m, pk->num, pk->den, nullptr));
  PK_TRY(pm::fr_batch_inverse_mul(ctx, pk->den, pk->num, m, nullptr));    // den[i] <- num[i] / den[i]
  PK_TRY(pm_fr_prefix_product_dev(ctx, pk->den, m, pk->num, nullptr));   // num[i] = product of this rank's ratios [0 .. i)
  {
    // the product of the whole slice = num[m - 1] den[m - 1]; every rank's goes round, the ranks below give the carry
    u64 last[2][4];
    PK_TRY(pm_dev_download(ctx, last[0], at(pk->num, m - 1), 32));
    PK_TRY(pm_dev_download(ctx, last[1], at(pk->den, m - 1), 32));
    const HFr total = fmul(get(last[0]), get(last[1]));
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, &total, 1, all));
    HFr carry = one;
    for (uint32_t r = 0; r < rk; ++r) carry = fmul(carry, all[r]);
    PK_TRY(upload_scalar(ctx, pk, carry));
    PK_TRY(pm_fr_vec_op_dev(ctx, 2, pk->num, pk->scalar, 1, pk->num, m, nullptr));
  }
  PM_HIP(ctx, hipMemcpyAsync(z_coeffs, pk->num, m * 32, hipMemcpyDeviceToDevice, st));
  PK_TRY(dist_ntt(ctx, D, pk, z_coeffs, 1, nullptr, PM_NTT_INVERSE));
  {
    const void* src[1] = {z_coeffs};
    PK_TRY(dist_to_coset(ctx, D, pk, src, 1, pk->coset_z, pk->halo_z));
  }
  PK_TRY(commit_batch_dist(ctx, D, pk, ck, z_coeffs, n, 1, &out->commitments[4]));
  ts.append_commitment(tl::PERM, out->commitments[4]);
  // ---- round 3 --------------------------------------------------------------------------------
  const HFr alpha = ts.challenge_scalar(tl::ALPHA);
  const HFr range_sep = ts.challenge_scalar(tl::RANGE_SEP);
  const HFr logic_sep = ts.challenge_scalar(tl::LOGIC_SEP);
  const HFr fixed_sep = ts.challenge_scalar(tl::FIXED_SEP);
  const HFr var_sep = ts.challenge_scalar(tl::VAR_SEP);
  pm_plonk_quotient_args qa;
  memset(&qa, 0, sizeof qa);
  for (int j = 0; j < 4; ++j) {
    qa.wires[j] = coset(j);
    qa.sigmas[j] = at(pk->sigma_coset, 4 * m * j);
  }
  qa.z = pk->coset_z;
  qa.pi = coset(4);
  qa.q_m = pk->sel_coset[Q_M];
  qa.q_l = pk->sel_coset[Q_L];
  qa.q_r = pk->sel_coset[Q_R];
  qa.q_o = pk->sel_coset[Q_O];
  qa.q_c = pk->sel_coset[Q_C];
  qa.q_4 = pk->sel_coset[Q_4];
  qa.q_arith = pk->sel_coset[Q_ARITH];
  qa.q_range = pk->sel_coset[Q_RANGE];
  qa.q_logic = pk->sel_coset[Q_LOGIC];
  qa.q_fixed_group_add = pk->sel_coset[Q_FIXED];
  qa.q_variable_group_add = pk->sel_coset[Q_VAR];
  qa.l1 = pk->lx_coset;
  qa.x = at(pk->lx_coset, 4 * m);
  put(qa.alpha, alpha);
  put(qa.beta, beta);
  put(qa.gamma, gamma);
  put(qa.range_sep, range_sep);
  put(qa.logic_sep, logic_sep);
  put(qa.fixed_sep, fixed_sep);
  put(qa.var_sep, var_sep);
  for (int j = 0; j < 3; ++j) put(qa.k[j], pk->k[j]);
  for (int j = 0; j < 4; ++j) put(qa.zh_inv[j], pk->zh_inv[j]);
  {
    // the same kernel on the rank's 4 m points, planar layout: "one row further" is index + n2, and past the rank's last row
    // the halo rows of a, b, d and z that came with the transforms (the last rank's are rank 0's first rows, shifted by one)
    pm::QuotPlanar ql;
    memset(&ql, 0, sizeof ql);
    ql.n2 = (uint32_t)n2;
    ql.rot = rk + 1 == W ? 1u : 0u;
    for (int j = 0; j < 4; ++j) ql.halo_w[j] = at(pk->halo_a, 4 * n2 * j);
    ql.halo_z = pk->halo_z;
    PK_TRY(pm::plonk_quotient_layout(ctx, &qa, m, false, &ql, pk->tq, nullptr));
  }
  {
    // t's coefficients from its values on the four sub-cosets: D_s = iNTT_n(t on g_s H) -- ONE batched inverse transform that
    // takes the block-transposed order --, D'_s[i] = D_s[i] g_s^-i = sum_m' (c_{i + m' n} g^(n m')) i4^(s m'):  a 4-point DFT
    // over m' for every i, undone pointwise
    PK_TRY(dist_ntt(ctx, D, pk, pk->tq, 4, nullptr, PM_NTT_INVERSE | PM_NTT_TRANSPOSED));
    PK_TRY(pm_fr_vec_op_dev(ctx, 2, pk->tq, pk->gs_inv_pow, 4 * m, pk->tq, 4 * m, nullptr));
    const HFr i4 = fpow(fmul(pk->gs[1], finv(pk->gs[0])), n), i4_inv = finv(i4);   // w4^n: a primitive fourth root of unity
    const HFr gn_inv = finv(fpow(pk->gs[0], n)), quarter = finv(fr_u64(4));
    HFr gm = one;   // g^(-n m')
    for (int mp = 0; mp < 4; ++mp) {
      const void* v[4];
      u64 c[4][4];
      for (int s = 0; s < 4; ++s) {
        v[s] = at(pk->tq, s * m);
        put(c[s], fmul(fmul(quarter, gm), fpow64(i4_inv, (u64)(s * mp))));
      }
      PK_TRY(pm_fr_lincomb_dev(ctx, 4, v, &c[0][0], m, at(pk->t, mp * m), nullptr));
      gm = fmul(gm, gn_inv);
    }
  }
  PK_TRY(commit_batch_dist(ctx, D, pk, ck, pk->t, n, 4, &out->commitments[5]));
  for (int i = 0; i < 4; ++i) ts.append_commitment(tl::QUOTIENT[i], out->commitments[5 + i]);
  // ---- round 4 --------------------------------------------------------------------------------
  const HFr zc = ts.challenge_scalar(tl::Z_CHALLENGE), zw = fmul(zc, pk->omega);
  enum { E_A, E_B, E_C, E_D, E_AN, E_BN, E_DN, E_S1, E_S2, E_S3, E_QARITH, E_QC, E_QL, E_QR, E_ZN, E_T, E_R, NEV };
  HFr ev[NEV];
  const HFr zc_lo = fpow64(zc, lo), zw_lo = fpow64(zw, lo);
  // r(z) is the linear combination of the values at z of the polynomials r combines (prover.hip, round 4): the ones the proof
  // does not open ride along as a second group -- one host synchronisation and ONE exchange for all openings, none for r
  enum { X_QM, X_QO, X_Q4, X_Z, X_S4, X_RANGE, X_LOGIC, X_FIXED, X_VAR, NX };
  HFr xv[NX];
  {
    // openings: every rank evaluates its coefficient slice (sum_i c_{lo + i} z^i), scales by z^lo, and the sums go round
    const void* at_z[15];
    const void* at_x[NX];
    u64 out_z[15][4], out_x[NX][4], out_zw[4][4];
    for (int j = 0; j < 4; ++j) at_z[j] = at(pk->coeffs, j * m);
    for (int j = 0; j < 3; ++j) at_z[4 + j] = at(pk->sigma_coeffs, j * m);
    at_z[7] = at(pk->sel_coeffs, Q_ARITH * m);
    at_z[8] = at(pk->sel_coeffs, Q_C * m);
    at_z[9] = at(pk->sel_coeffs, Q_L * m);
    at_z[10] = at(pk->sel_coeffs, Q_R * m);
    for (int i = 0; i < 4; ++i) at_z[11 + i] = at(pk->t, i * m);
    at_x[X_QM] = at(pk->sel_coeffs, Q_M * m);
    at_x[X_QO] = at(pk->sel_coeffs, Q_O * m);
    at_x[X_Q4] = at(pk->sel_coeffs, Q_4 * m);
    at_x[X_Z] = z_coeffs;
    at_x[X_S4] = at(pk->sigma_coeffs, 3 * m);
    uint32_t nx = X_RANGE;
    const int wsel[4] = {Q_RANGE, Q_LOGIC, Q_FIXED, Q_VAR};
    int xslot[4] = {-1, -1, -1, -1};
    for (int w = 0; w < 4; ++w)
      if (!pk->sel_zero[wsel[w]]) {
        xslot[w] = (int)nx;
        at_x[nx++] = at(pk->sel_coeffs, wsel[w] * m);
      }
    const void* at_zw[4] = {at(pk->coeffs, 0), at(pk->coeffs, m), at(pk->coeffs, 3 * m), z_coeffs};
    const uint32_t gk[3] = {15, nx, 4};
    const void* const* gp[3] = {at_z, at_x, at_zw};
    const uint64_t* gpt[3] = {zc.l, zc.l, zw.l};
    uint64_t* gout[3] = {&out_z[0][0], &out_x[0][0], &out_zw[0][0]};
    PK_TRY(pm::poly_evaluate_groups(ctx, 3, gk, gp, gpt, gout, m));
    constexpr uint32_t NP = 15 + 4 + NX;
    HFr part[NP];
    for (int j = 0; j < 15; ++j) part[j] = fmul(get(out_z[j]), zc_lo);
    for (int j = 0; j < 4; ++j) part[15 + j] = fmul(get(out_zw[j]), zw_lo);
    for (uint32_t j = 0; j < NX; ++j) part[19 + j] = j < nx ? fmul(get(out_x[j]), zc_lo) : pm::host::zero<4>();
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, part, NP, all));
    HFr sum[NP];
    for (uint32_t j = 0; j < NP; ++j) {
      sum[j] = pm::host::zero<4>();
      for (uint32_t r = 0; r < W; ++r) sum[j] = fadd(sum[j], all[(size_t)r * NP + j]);
    }
    for (int j = 0; j < 4; ++j) ev[E_A + j] = sum[j];
    for (int j = 0; j < 3; ++j) ev[E_S1 + j] = sum[4 + j];
    ev[E_QARITH] = sum[7];
    ev[E_QC] = sum[8];
    ev[E_QL] = sum[9];
    ev[E_QR] = sum[10];
    ev[E_AN] = sum[15];
    ev[E_BN] = sum[16];
    ev[E_DN] = sum[17];
    ev[E_ZN] = sum[18];
    const HFr zn_ = fpow(zc, n);
    ev[E_T] = fadd(sum[11], fmul(zn_, fadd(sum[12], fmul(zn_, fadd(sum[13], fmul(zn_, sum[14]))))));
    for (int j = 0; j < X_RANGE; ++j) xv[j] = sum[19 + j];
    for (int w = 0; w < 4; ++w) xv[X_RANGE + w] = xslot[w] >= 0 ? sum[19 + xslot[w]] : pm::host::zero<4>();
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
    HFr r_z = pm::host::zero<4>();
    auto term = [&](const void* v, const HFr& c, const HFr& value_at_z) {
      lin_v[k] = v;
      put(lin_c[k], c);
      r_z = fadd(r_z, fmul(c, value_at_z));
      ++k;
    };
    term(at(pk->sel_coeffs, Q_M * m), fmul(qar, fmul(a_, b_)), xv[X_QM]);
    term(at(pk->sel_coeffs, Q_L * m), fmul(qar, a_), ev[E_QL]);
    term(at(pk->sel_coeffs, Q_R * m), fmul(qar, b_), ev[E_QR]);
    term(at(pk->sel_coeffs, Q_O * m), fmul(qar, c_), xv[X_QO]);
    term(at(pk->sel_coeffs, Q_4 * m), fmul(qar, d_), xv[X_Q4]);
    term(at(pk->sel_coeffs, Q_C * m), qar, ev[E_QC]);
    if (!pk->sel_zero[Q_RANGE]) term(at(pk->sel_coeffs, Q_RANGE * m), widget_range(range_sep, re), xv[X_RANGE]);
    if (!pk->sel_zero[Q_LOGIC]) term(at(pk->sel_coeffs, Q_LOGIC * m), widget_logic(logic_sep, re), xv[X_LOGIC]);
    if (!pk->sel_zero[Q_FIXED]) term(at(pk->sel_coeffs, Q_FIXED * m), widget_fixed(fixed_sep, re), xv[X_FIXED]);
    if (!pk->sel_zero[Q_VAR]) term(at(pk->sel_coeffs, Q_VAR * m), widget_var(var_sep, re), xv[X_VAR]);
    term(z_coeffs, fadd(fmul(alpha, ident), fmul(alpha2, l1_z)), xv[X_Z]);
    term(at(pk->sigma_coeffs, 3 * m), fneg(fmul(fmul(fmul(alpha, copy3), beta), z_next)), xv[X_S4]);
    PK_TRY(pm_fr_lincomb_dev(ctx, k, lin_v, &lin_c[0][0], m, pk->r, nullptr));
    ev[E_R] = r_z;
  }
  static_assert(NEV == 17, "tl::EVALS lists the evaluations in this enum's order");
  for (int i = 0; i < NEV; ++i) {
    ts.append_scalar(tl::EVALS[i], ev[i]);
    put(out->evaluations[i], ev[i]);
  }
  // ---- round 5 --------------------------------------------------------------------------------
  const HFr aw = ts.challenge_scalar(tl::AGGREGATE);
  const HFr aws = ts.challenge_scalar(tl::AGGREGATE);
  void *agg1 = pk->agg, *agg2 = at(pk->agg, m);
  {
    const void* agg_v[12];
    u64 agg_c[12][4];
    HFr ac[12];
    ac[0] = one;
    ac[1] = zn;
    ac[2] = fmul(zn, zn);
    ac[3] = fmul(ac[2], zn);
    HFr vp = one;
    for (int e = 0; e < 8; ++e) {
      vp = fmul(vp, aw);
      ac[4 + e] = vp;
    }
    for (int i = 0; i < 4; ++i) agg_v[i] = at(pk->t, i * m);
    agg_v[4] = pk->r;
    for (int j = 0; j < 4; ++j) agg_v[5 + j] = at(pk->coeffs, j * m);
    for (int j = 0; j < 3; ++j) agg_v[9 + j] = at(pk->sigma_coeffs, j * m);
    for (int i = 0; i < 12; ++i) put(agg_c[i], ac[i]);
    PK_TRY(pm_fr_lincomb_dev(ctx, 12, agg_v, &agg_c[0][0], m, agg1, nullptr));
    const void* sh_v[4] = {z_coeffs, at(pk->coeffs, 0), at(pk->coeffs, m), at(pk->coeffs, 3 * m)};
    u64 sh_c[4][4];
    vp = one;
    for (int e = 0; e < 4; ++e) {
      put(sh_c[e], vp);
      vp = fmul(vp, aws);
    }
    PK_TRY(pm_fr_lincomb_dev(ctx, 4, sh_v, &sh_c[0][0], m, agg2, nullptr));
  }
  {
    // Ruffini over the ranks: q_k = sum_{j > k} c_j z^(j - k - 1).  The part with j inside the slice is the slice's own
    // division; the part above it is  z^(hi - 1 - k) H,  H = sum over the ranks above of (z^m)^(r' - r - 1) P_r',
    // P_r' = that rank's slice evaluated at z (sum_i c_{lo' + i} z^i): one exchange of two scalars.
    HFr P[2];
    PK_TRY(pm_fr_poly_evaluate_dev(ctx, agg1, m, zc.l, P[0].l, nullptr));
    PK_TRY(pm_fr_poly_evaluate_dev(ctx, agg2, m, zw.l, P[1].l, nullptr));
    std::vector<HFr> all;
    PK_TRY(dist_scalars(ctx, D, P, 2, all));
    const HFr pts[2] = {zc, zw};
    const void* src[2] = {agg1, agg2};
    for (int q = 0; q < 2; ++q) {
      void* dst = at(pk->wit, q * m);
      PK_TRY(pm_fr_poly_ruffini_dev(ctx, src[q], m, pts[q].l, dst, nullptr));   // m - 1 coefficients
      PM_HIP(ctx, hipMemsetAsync(at(dst, m - 1), 0, 32, st));
      const HFr zm = fpow64(pts[q], m);
      HFr H = pm::host::zero<4>();
      for (uint32_t r = W; r-- > rk + 1;) H = fadd(fmul(H, zm), all[(size_t)r * 2 + q]);
      const HFr zi = finv(pts[q]);
      PK_TRY(pm_fr_powers_dev(ctx, zi.l, fmul(H, fpow64(pts[q], m - 1)).l, m, pk->tmp, nullptr));   // H z^(m - 1 - i)
      PK_TRY(pm_fr_vec_op_dev(ctx, 0, dst, pk->tmp, m, dst, m, nullptr));
    }
  }
  PK_TRY(commit_batch_dist(ctx, D, pk, ck, pk->wit, n - 1, 2, &out->commitments[9]));
  ts.append_commitment(tl::W_Z, out->commitments[9]);
  ts.append_commitment(tl::W_ZW, out->commitments[10]);
  const HFr chal[10] = {beta, gamma, alpha, range_sep, logic_sep, fixed_sep, var_sep, zc, aw, aws};
  for (int i = 0; i < 10; ++i) put(out->challenges[i], chal[i]);
  return PM_OK;
}

extern "C" int pm_plonk_prove_dist(pm_ctx* ctx, const pm_dist* dist, pm_dist_key* key, const pm_bases* ck_slice,
                                   const void* d_witness_slices, const uint64_t* pi_positions, const uint64_t* pi_values,
                                   size_t n_pi, uint32_t flags, pm_plonk_proof* out) {
  if (!ctx || !dist) return PM_ERR_BAD_ARG;
  Dist D;
  D.d = *dist;
  D.expect = 8;
  return dist_leave(ctx, D, prove_dist_body(ctx, D, key, ck_slice, d_witness_slices, pi_positions, pi_values, n_pi, flags, out));
}
