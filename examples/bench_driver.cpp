// C++ bench driver over the C ABI alone ("the Python ctypes harness and the C++ bench driver"): no Python,
// no torch, nothing but include/plonk_mi355x.h -- what a Rust `-sys` crate sees.  It times the two halves of the headline metric
// on device-resident synthetic data and checks both results before printing:
//   * forward + inverse BLS12-381 Fr NTT at 2^log_n (pm_fr_ntt_dev, the context's own stream, wall clock around pm_sync):
//     butterflies/s = steps * 2 * (n / 2) * log_n / t; check: iNTT(NTT(a)) == a byte for byte;
//   * a 2^log_n-point G1 MSM over a resident SRS stand-in P_i = g^i G generated ON the device (pm_fr_powers_dev +
//     pm_g1_fixed_base_mul_dev + pm_g1_bases_from_dev, window table precomputed): scalar-muls/s = n / t; check: the
//     discrete-log identity sum_i s_i P_i == (sum_i s_i g^i) G, the right side by one 1-point MSM.
// bench.py stays the driver-facing benchmark (one JSON line with roofline and CPU baseline); this file shows the same library
// reaching the same rates without the Python process around it.
//   g++ -std=c++17 -O2 examples/bench_driver.cpp -Iinclude -Lplonk-prototype_amd/lib -lplonk_mi355x
//       -Wl,-rpath,$PWD/plonk-prototype_amd/lib -o bench_driver && ./bench_driver [log_n = 20] [steps = 50] [msm_steps = 5]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "plonk_mi355x.h"

static const uint64_t R_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
// Fr one in Montgomery form (2^256 mod r) and dusk_bls12_381::G1Affine::generator(), Montgomery limbs
static const uint64_t FR_ONE[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const uint64_t G1_GEN[12] = {0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL,
                                    0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL, 0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL,
                                    0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL};

static uint64_t rng_state = 0x504C4F4E4BULL;
static uint64_t next_u64() {  // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// r = (a + b) mod R_MOD for a, b < R_MOD (Montgomery forms add like the values they stand for)
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
    const unsigned __int128 x = (unsigned __int128)t[i] - R_MOD[i] - (uint64_t)bw;
    d[i] = (uint64_t)x;
    bw = (x >> 64) & 1;
  }
  memcpy(r, (c || !bw) ? d : t, 32);
}

#define CHECK(call)                                                                            \
  do {                                                                                         \
    const int rc_ = (call);                                                                    \
    if (rc_ != PM_OK) {                                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? pm_last_error(ctx) : "no context"); \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

int main(int argc, char** argv) {
  const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 20;
  const int steps = argc > 2 ? atoi(argv[2]) : 50, msm_steps = argc > 3 ? atoi(argv[3]) : 5;
  if (log_n < 4 || log_n > 26 || steps < 1 || msm_steps < 1) return fprintf(stderr, "usage: bench_driver [log_n 4..26] [steps] [msm_steps]\n"), 2;
  const size_t n = (size_t)1 << log_n;
  pm_ctx* ctx = nullptr;
  CHECK(pm_init(0, &ctx));  // fails loudly without a gfx950 device: there is no CPU path

  // ---- NTT: forward + inverse, device resident
  std::vector<uint64_t> a(4 * n), back(4 * n);
  for (size_t i = 0; i < n; ++i) {
    for (int l = 0; l < 4; ++l) a[4 * i + l] = next_u64();
    a[4 * i + 3] &= ((uint64_t)1 << 62) - 1;  // < 2^254 < r: a valid BlsScalar
  }
  void *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;
  CHECK(pm_dev_alloc(ctx, n * 32, &d_a));
  CHECK(pm_dev_alloc(ctx, n * 32, &d_b));
  CHECK(pm_dev_alloc(ctx, n * 32, &d_c));
  CHECK(pm_dev_upload(ctx, d_a, a.data(), n * 32));
  auto step = [&]() -> int {
    int rc = pm_fr_ntt_dev(ctx, d_a, n, n, d_b, n, log_n, 1, 0, nullptr);
    return rc ? rc : pm_fr_ntt_dev(ctx, d_b, n, n, d_c, n, log_n, 1, PM_NTT_INVERSE, nullptr);
  };
  for (int i = 0; i < 5; ++i) CHECK(step());  // tables, code load, clocks
  CHECK(pm_sync(ctx));
  const auto t0 = clk::now();
  for (int i = 0; i < steps; ++i) CHECK(step());
  CHECK(pm_sync(ctx));
  const double t_ntt = secs(t0, clk::now()) / steps;
  CHECK(pm_dev_download(ctx, back.data(), d_c, n * 32));
  if (memcmp(back.data(), a.data(), n * 32)) return fprintf(stderr, "iNTT(NTT(a)) != a\n"), 1;
  CHECK(pm_dev_download(ctx, back.data(), d_b, n * 32));
  if (!memcmp(back.data(), a.data(), n * 32)) return fprintf(stderr, "NTT(a) == a?\n"), 1;

  // ---- MSM over P_i = g^i G (the SRS stand-in is made on the device; its discrete logs are known)
  uint64_t g[4];
  for (int l = 0; l < 4; ++l) g[l] = next_u64();
  g[3] &= ((uint64_t)1 << 62) - 1;
  void *d_k = nullptr, *d_pts = nullptr;
  CHECK(pm_dev_alloc(ctx, n * 32, &d_k));
  CHECK(pm_dev_alloc(ctx, n * 96, &d_pts));
  CHECK(pm_fr_powers_dev(ctx, g, FR_ONE, n, d_k, nullptr));
  CHECK(pm_g1_fixed_base_mul_dev(ctx, G1_GEN, d_k, n, PM_SCALAR_MONTGOMERY, d_pts, nullptr));
  pm_bases* srs = nullptr;
  CHECK(pm_g1_bases_from_dev(ctx, d_pts, n, &srs));
  CHECK(pm_dev_free(ctx, d_pts));
  const auto tp = clk::now();
  CHECK(pm_g1_bases_precompute(ctx, srs, 0));  // a resident commit key: window multiples once
  CHECK(pm_sync(ctx));
  const double t_table = secs(tp, clk::now());
  uint64_t res[18], res_xy[12];
  CHECK(pm_g1_msm_dev(ctx, srs, 0, n, d_a, PM_SCALAR_MONTGOMERY, res, nullptr));  // blocks until the point is on the host
  const auto t1 = clk::now();
  for (int i = 0; i < msm_steps; ++i) CHECK(pm_g1_msm_dev(ctx, srs, 0, n, d_a, PM_SCALAR_MONTGOMERY, res, nullptr));
  const double t_msm = secs(t1, clk::now()) / msm_steps;
  // check: sum_i s_i g^i by host arithmetic (the library's host product through its test hook, our own additions), then
  // one 1-point MSM of that scalar against G
  std::vector<uint64_t> k(4 * n), prod(4 * n);
  CHECK(pm_dev_download(ctx, k.data(), d_k, n * 32));
  CHECK(pm_test_host_field_op(0, a.data(), k.data(), prod.data(), n));
  uint64_t sum[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) fr_add(sum, sum, &prod[4 * i]);
  pm_bases* gen = nullptr;
  uint64_t want[18], want_xy[12];
  int inf_a = 0, inf_b = 0;
  CHECK(pm_g1_bases_upload(ctx, G1_GEN, 1, &gen));
  CHECK(pm_g1_msm(ctx, gen, 1, sum, PM_SCALAR_MONTGOMERY, want));
  CHECK(pm_g1_to_affine(res, res_xy, &inf_a));
  CHECK(pm_g1_to_affine(want, want_xy, &inf_b));
  if (inf_a || inf_b || memcmp(res_xy, want_xy, 96)) return fprintf(stderr, "MSM differs from the discrete-log identity\n"), 1;

  const double bfly = (double)n * log_n / t_ntt;  // 2 transforms x (n / 2) log n
  printf("{\"driver\": \"examples/bench_driver.cpp: C ABI only, no Python, no torch\", \"log_n\": %u, \"steps\": %d, "
         "\"ms_per_fwd_inv\": %.4f, \"ntt_butterflies_per_s\": %.4e, \"msm_steps\": %d, \"ms_per_msm\": %.3f, "
         "\"msm_scalar_muls_per_s\": %.4e, \"srs_window_table_ms\": %.1f, \"checks\": \"round trip byte-equal; MSM == (sum s_i g^i) G\"}\n",
         log_n, steps, t_ntt * 1e3, bfly, msm_steps, t_msm * 1e3, (double)n / t_msm, t_table * 1e3);
  pm_g1_bases_free(ctx, gen);
  pm_g1_bases_free(ctx, srs);
  for (void* p : {d_a, d_b, d_c, d_k}) CHECK(pm_dev_free(ctx, p));
  pm_shutdown(ctx);
  printf("bench_driver OK\n");
  return 0;
}
