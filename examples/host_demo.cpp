// The C++ host mirror (include/plonk_mi355x.hpp) used the way the reference's Rust code uses
// dusk-plonk: EvaluationDomain, msm_variable_base, CommitKey, Polynomial -- no Python in the process.
//   g++ -std=c++17 -O2 examples/host_demo.cpp -Iinclude -Lplonk-prototype_amd/lib -lplonk_mi355x
//       -Wl,-rpath,$PWD/plonk-prototype_amd/lib -o host_demo && ./host_demo     (one command line)
// Exit code 0 and "host_demo OK" on success; without a gfx950 device it reports the Error and exits 1.
#include <cstdio>
#include <cstring>

#include "plonk_mi355x.hpp"

using namespace plonk_mi355x;

static uint64_t rng_state = 0x243F6A8885A308D3ULL;
static uint64_t next_u64() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static Fr random_fr() {
  Fr r{next_u64(), next_u64(), next_u64(), next_u64() & ((1ULL << 62) - 1)};   // < 2^254 < r
  return r;
}
static const G1Affine G1_GEN = {0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL,
                                0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL,
                                0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL,
                                0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL};
#define REQUIRE(cond)                                              \
  do {                                                             \
    if (!(cond)) {                                                 \
      std::fprintf(stderr, "host_demo: failed: %s\n", #cond);      \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main() {
  try {
    Context ctx(0);
    // ---- EvaluationDomain ---------------------------------------------------------------------
    EvaluationDomain dom(ctx, 1000);
    REQUIRE(dom.size == 1024 && dom.log_size_of_group == 10);
    std::vector<Fr> a(1000), b(1000);
    for (auto& x : a) x = random_fr();
    for (auto& x : b) x = random_fr();
    std::vector<Fr> padded = a;
    padded.resize(1024, Fr{0, 0, 0, 0});
    REQUIRE(dom.ifft(dom.fft(a)) == padded);
    REQUIRE(dom.coset_ifft(dom.coset_fft(a)) == padded);
    std::vector<Fr> inplace = a;
    dom.fft_in_place(inplace);
    REQUIRE(inplace == dom.fft(a));
    std::vector<Fr> el = dom.elements();
    REQUIRE(el[0] == EvaluationDomain::one() && el[1] == dom.group_gen);
    // the small domain helpers: Z_H vanishes on H, the Lagrange basis at w^5 is the indicator of 5, and the vanishing
    // polynomial of degree 256 on the coset takes four values over the 1024-point domain
    const Fr zero_fr{0, 0, 0, 0};
    REQUIRE(dom.evaluate_vanishing_polynomial(el[7]) == zero_fr && !(dom.evaluate_vanishing_polynomial(random_fr()) == zero_fr));
    std::vector<Fr> lag = dom.evaluate_all_lagrange_coefficients(el[5]);
    REQUIRE(lag[5] == EvaluationDomain::one() && lag[4] == zero_fr && lag[1023] == zero_fr);
    std::vector<Fr> vh = dom.compute_vanishing_poly_over_coset(256);
    REQUIRE(vh[0] == vh[4] && vh[1] == vh[1021] && !(vh[0] == vh[1]));
    auto many = dom.fft_many({a, b, a}, PM_NTT_COSET);
    REQUIRE(many.size() == 3 && many[0] == dom.coset_fft(a) && many[1] == dom.coset_fft(b) && many[2] == many[0]);
    bool threw = false;
    try { EvaluationDomain too_big(ctx, (size_t)1 << 33); } catch (const Error& e) { threw = e.code == PM_ERR_DOMAIN_TOO_LARGE; }
    REQUIRE(threw);                                        // InvalidEvalDomainSize
    threw = false;
    try { dom.fft(std::vector<Fr>(1025, Fr{1, 0, 0, 0})); } catch (const Error& e) { threw = e.code == PM_ERR_LENGTH; }
    REQUIRE(threw);
    // linearity, with the sum formed on the device: fft(a) + fft(b) == fft(a + b)
    DevicePolynomial da(ctx, a), db(ctx, b);
    REQUIRE(DevicePolynomial(ctx, dom.fft(a)).add(DevicePolynomial(ctx, dom.fft(b))).to_host() == dom.fft(da.add(db).to_host()));
    REQUIRE(da.ntt(10, 0).to_host() == dom.fft(a));        // the device-resident transform is the same transform

    // ---- msm_variable_base / CommitKey ----------------------------------------------------------
    const size_t m = 600;
    std::vector<G1Affine> pts(m, G1_GEN);
    std::vector<Fr> sc(a.begin(), a.begin() + m);
    CommitKey ck(ctx, pts, /*precompute=*/true);
    REQUIRE(ck.max_degree() == m - 1);
    G1Affine c1 = ck.commit(sc);
    REQUIRE(c1 == to_affine(msm_variable_base(ctx, pts, sc)));
    // all bases equal G: commit(s) = (sum s_i) G = commit of the one-term polynomial [p(1)]
    DevicePolynomial dsc(ctx, sc);
    Fr sum = dsc.evaluate(EvaluationDomain::one());
    REQUIRE(c1 == ck.commit({sum}));
    REQUIRE(dsc.commit(ck) == c1);
    bool ident = false;
    to_affine(msm_variable_base(ctx, {}, {}), &ident);
    REQUIRE(ident);                                        // the empty sum is the identity
    threw = false;
    try { ck.commit(std::vector<Fr>(m + 1, Fr{0, 0, 0, 0})); } catch (const Error& e) { threw = e.code == PM_ERR_LENGTH; }
    REQUIRE(threw);                                        // PolynomialDegreeTooLarge

    // ---- Polynomial: p(X) - p(z) = q(X) (X - z) at a random point, all on the device --------------
    Fr z = random_fr(), x = random_fr();
    DevicePolynomial q = da.ruffini(z);
    REQUIRE(q.len() == a.size() - 1);
    DevicePolynomial px(ctx, std::vector<Fr>{da.evaluate(x)}), pz(ctx, std::vector<Fr>{da.evaluate(z)});
    DevicePolynomial qx(ctx, std::vector<Fr>{q.evaluate(x)}), dx(ctx, std::vector<Fr>{x}), dz(ctx, std::vector<Fr>{z});
    REQUIRE(px.sub(pz).to_host() == qx.mul(dx.sub(dz)).to_host());

    // ---- ProverKey / prove: a 64-gate circuit  a - c = 0  (q_l = 1, q_o = -1), no copy constraints ----
    const size_t gn = 64;
    const Fr zero{0, 0, 0, 0}, one = EvaluationDomain::one();
    const Fr minus_one = DevicePolynomial(ctx, std::vector<Fr>{zero}).sub(DevicePolynomial(ctx, std::vector<Fr>{one})).to_host()[0];
    std::array<std::vector<Fr>, PM_PLONK_SELECTORS> sel;   // the widget selectors stay empty = identically zero
    for (int s2 = 0; s2 < 7; ++s2) sel[s2].assign(gn, zero);
    sel[1].assign(gn, one);                                 // q_l
    sel[3].assign(gn, minus_one);                           // q_o
    sel[6].assign(gn, one);                                 // q_arith
    std::vector<int64_t> sigma(4 * gn);
    for (size_t p2 = 0; p2 < 4 * gn; ++p2) sigma[p2] = (int64_t)p2;
    std::vector<Fr> wit(4 * gn);
    for (size_t i = 0; i < gn; ++i) {
      wit[i] = wit[2 * gn + i] = random_fr();               // a = c
      wit[gn + i] = random_fr();
      wit[3 * gn + i] = random_fr();
    }
    CommitKey ck64(ctx, std::vector<G1Affine>(pts.begin(), pts.begin() + gn));
    ProverKey pk(ctx, sel, sigma, ck64);
    DevicePolynomial dwit(ctx, wit);
    Proof proof = pk.prove(ck64, dwit);
    Proof proof2 = pk.prove(ck64, dwit);
    REQUIRE(proof.commitments == proof2.commitments && proof.evaluations == proof2.evaluations);
    EvaluationDomain dom64(ctx, gn);
    REQUIRE(proof.commitments[0] == ck64.commit(dom64.ifft(std::vector<Fr>(wit.begin(), wit.begin() + gn))));   // [a]
    REQUIRE(proof.commitments[0] == proof.commitments[2]);  // a = c as polynomials
    REQUIRE(proof.bytes == proof2.bytes);
    ProverKey pk_other(ctx, sel, sigma, ck64, "other");      // another transcript label: other challenges, same [a]
    Proof proof3 = pk_other.prove(ck64, dwit);
    REQUIRE(proof3.challenges[0] != proof.challenges[0] && proof3.commitments[0] == proof.commitments[0]);
    REQUIRE(pk.verifier_key()[1] == ck64.commit(dom64.ifft(sel[1])));                                            // [q_l]
    // a public input at gate 3 (PI enters the gate equation: a - c + PI = 0 no longer holds, but the
    // transcript must see it): other challenges than without
    REQUIRE(pk.prove(ck64, dwit, {PublicInput{3, one}}).challenges[0] != proof.challenges[0]);
    // the distributed prover (pm_plonk_*_dist) on ONE rank -- the sub-coset decomposition without any exchange; the all-gather
    // of one rank is a copy: the same proof and verifier key, byte for byte
    pm_dist d1{1, 0, [](void*, const uint64_t* msg, uint64_t* gathered) -> int {
                 std::memcpy(gathered, msg, PM_COMM_MSG_WORDS * sizeof(uint64_t));
                 return 0;
               }, nullptr, nullptr};
    DistProverKey dpk(ctx, d1, sel, sigma, gn, ck64);
    ctx.comm_stats(true);
    REQUIRE(dpk.prove(ck64, dwit).bytes == proof.bytes && dpk.verifier_key() == pk.verifier_key() && dpk.device_bytes() > 0);
    // what the same proof would put on the wire with more than one rank: 12 transposes = all-to-all calls, 8 message all-gathers
    const Context::CommStats cs = ctx.comm_stats();
    REQUIRE(cs.transpose_steps == 12 && cs.allgather_calls == 8 && cs.alltoall_calls == 0);
    std::printf("host_demo OK (EvaluationDomain, msm_variable_base, CommitKey, Polynomial, ProverKey, DistProverKey over %s)\n", pm_version());
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "plonk_mi355x::Error %d: %s\n", e.code, e.what());
    return 1;
  }
}
