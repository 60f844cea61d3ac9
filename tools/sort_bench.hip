// Stand-alone check + timing of the MSM bucket fill (csrc/msm_sort.hip.h): digits -> partitions -> local sort.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 tools/sort_bench.hip -o tools/sort_bench
//   ./tools/sort_bench [quick]
// Every case is compared with a CPU restatement of the digit rule (same multiset of (key, value) pairs, keys
// non-decreasing, every bucket contiguous); the large cases are checked by order + checksums only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define SORT_TIMING 1
#include "../plonk-prototype_amd/csrc/msm_sort.hip.h"

using namespace pm;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);    \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// CPU digits of a canonical 256-bit scalar (8 x u32), the rule of msm_variable_base with signed windows
static void cpu_pairs(const uint32_t* w, const MsmGeom& g, uint32_t j, uint32_t i, uint32_t offset, std::vector<uint64_t>& out) {
  uint32_t carry = 0;
  const uint32_t half = 1u << (g.c - 1);
  for (uint32_t k = 0; k < g.nwin; ++k) {
    const uint32_t lo = k * g.c, word = lo >> 5, sh = lo & 31;
    uint64_t two = (uint64_t)(word < 8 ? w[word] : 0u) | ((uint64_t)(word + 1 < 8 ? w[word + 1] : 0u) << 32);
    uint32_t d = (uint32_t)(two >> sh) & ((1u << g.c) - 1u);
    d += carry;
    uint32_t neg = 0;
    if (d > half) {
      d = (1u << g.c) - d;
      neg = 1;
      carry = 1;
    } else {
      carry = 0;
    }
    if (!d) continue;
    const uint32_t set = j * g.nsets + (g.nsets == 1 ? 0u : k);
    const uint32_t key = (set << g.bbits) | (d - 1);
    const uint32_t val = ((g.nsets == 1 ? k * g.row_stride : 0u) + offset + i) | (neg << 31);
    out.push_back(((uint64_t)key << 32) | val);
  }
}

struct Case {
  size_t n;
  uint32_t c;
  bool table;
  uint32_t batch;
  int dist;   // 0 uniform, 1 all equal, 2 witness-like, 3 all zero, 4 30 % equal, 5 long carry runs
  const char* name;
};

static bool run_case(const Case& cs, int reps, bool full_check) {
  const size_t n = cs.n;
  MsmGeom g = make_geom(n, cs.table ? 0 : cs.c, cs.table ? cs.c : 0, cs.table ? n : 0, cs.batch);
  if (g.bins > SORT_MAX_BINS) {
    printf("%-34s skipped: %u bins\n", cs.name, g.bins);
    return true;
  }
  const size_t m = n * cs.batch * g.nwin;
  // canonical scalars below r: top word below 0x73eda753
  std::vector<uint32_t> sc(8 * n * cs.batch);
  uint32_t eq[8];
  for (int q = 0; q < 8; ++q) eq[q] = (uint32_t)rnd();
  eq[7] %= 0x73eda753u;
  for (size_t i = 0; i < n * cs.batch; ++i) {
    uint32_t* w = &sc[8 * i];
    for (int q = 0; q < 8; ++q) w[q] = (uint32_t)rnd();
    w[7] %= 0x73eda753u;
    const double u = (double)(rnd() >> 11) / 9007199254740992.0;
    if (cs.dist == 1 || (cs.dist == 4 && u < 0.3)) memcpy(w, eq, 32);
    if (cs.dist == 3) memset(w, 0, 32);
    if (cs.dist == 5) {   // carries that run to the top: r - 1 - small, 2^254 - 1, a lone power of two
      static const uint32_t rm1[8] = {0x00000000u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
      if (u < 0.4) { memcpy(w, rm1, 32); w[2] -= (uint32_t)(rnd() & 0xff); }
      else if (u < 0.7) { for (int q = 0; q < 8; ++q) w[q] = 0xffffffffu; w[7] = 0x3fffffffu; }
      else { memset(w, 0, 32); const uint32_t b = (uint32_t)(rnd() % 254); w[b >> 5] = 1u << (b & 31); }
    }
    if (cs.dist == 2) {
      if (u < 0.90) { w[0] &= 0xffffu; for (int q = 1; q < 8; ++q) w[q] = 0; }
      else if (u < 0.95) memset(w, 0, 32);
      else if (u < 0.96) { memset(w, 0, 32); w[0] = 1; }
    }
  }
  void *d_sc, *d_ctl, *d_pairs, *d_keys, *d_vals, *d_canon, *d_rows;
  CK(hipMalloc(&d_canon, sc.size() * 4 + 32));
  CK(hipMalloc(&d_rows, (size_t)cs.batch * g.tiles * g.bins * 2 + 64));
  CK(hipMalloc(&d_sc, sc.size() * 4));
  CK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
  // ONE control block for every case, as in the library: it must come back to its idle state whatever the geometry
  static void* shared_ctl = nullptr;
  const uint32_t ctl_cap = 1u << 17;
  const size_t ctl_bytes = sort_ctl_words(ctl_cap) * 4;
  if (!shared_ctl) {
    CK(hipMalloc(&shared_ctl, ctl_bytes));
    CK(hipMemset(shared_ctl, 0, ctl_bytes));
  }
  d_ctl = shared_ctl;
  g.ctl_cap = ctl_cap;
  CK(hipMalloc(&d_pairs, (m + 1) * 8));
  CK(hipMalloc(&d_keys, (m + 1) * 4));
  CK(hipMalloc(&d_vals, (m + 1) * 4));
  CK(hipMemset(d_keys, 0xff, (m + 1) * 4));
  const uint32_t tiles_total = g.tiles * cs.batch;
  uint32_t tiles_per_wg = 1;
  while ((size_t)cs.batch * ((g.tiles + tiles_per_wg - 1) / tiles_per_wg) > 2048) ++tiles_per_wg;
  const uint32_t wgs_per_msm = (g.tiles + tiles_per_wg - 1) / tiles_per_wg;
  const size_t lds0 = (size_t)g.bins * 4, lds1 = sort_scatter_lds(g), lds2 = sort_local_lds(g);
  CK(hipFuncSetAttribute((const void*)msm_digits_scatter_kernel<SORT_THREADS1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  CK(hipFuncSetAttribute((const void*)msm_sort_local_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  {
    int o0 = 0, o1 = 0, o2 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o0, msm_digits_hist_kernel<SORT_THREADS0>, SORT_THREADS0, lds0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, msm_digits_scatter_kernel<SORT_THREADS1>, SORT_THREADS1, lds1));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, msm_sort_local_kernel, SORT_THREADS, lds2));
    printf("      occupancy (workgroups per CU, API): hist %d scatter %d local %d; LDS %zu / %zu / %zu B\n", o0, o1, o2, lds0, lds1, lds2);
  }
  hipEvent_t ev[4];
  for (auto& e : ev) CK(hipEventCreate(&e));
  float t0 = 0, t1 = 0, t2 = 0;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(ev[0], 0));
    hipLaunchKernelGGL(msm_digits_hist_kernel<SORT_THREADS0>, dim3(cs.batch * wgs_per_msm), dim3(SORT_THREADS0), lds0, 0, (const u32x4*)d_sc, n, n,
                       (u32)PM_SCALAR_CANONICAL, g, tiles_per_wg, wgs_per_msm, 131072u, 16u, (u32*)d_ctl, (u32x4*)d_canon,
                       (unsigned short*)d_rows);
    CK(hipEventRecord(ev[1], 0));
    hipLaunchKernelGGL(msm_digits_scatter_kernel<SORT_THREADS1>, dim3(std::min<uint32_t>(tiles_total, 256)), dim3(SORT_THREADS1), lds1, 0,
                       (const u32x4*)d_canon, (const unsigned short*)d_rows, n, g, 0u, (u32*)d_ctl, (u64*)d_pairs, tiles_total);
    CK(hipEventRecord(ev[2], 0));
    hipLaunchKernelGGL(msm_sort_local_kernel, dim3(g.np), dim3(SORT_THREADS), lds2, 0, g, (const u32*)d_ctl, (const u64*)d_pairs,
                       (u32*)d_keys, (u32*)d_vals);
    CK(hipEventRecord(ev[3], 0));
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (r == 0) continue;   // warm-up
    float a, b, c;
    CK(hipEventElapsedTime(&a, ev[0], ev[1]));
    CK(hipEventElapsedTime(&b, ev[1], ev[2]));
    CK(hipEventElapsedTime(&c, ev[2], ev[3]));
    t0 += a; t1 += b; t2 += c;
  }
  t0 /= reps; t1 /= reps; t2 /= reps;
  // results
  std::vector<uint32_t> ctl(sort_ctl_words(ctl_cap));
  CK(hipMemcpy(ctl.data(), d_ctl, ctl_bytes, hipMemcpyDeviceToHost));
  const uint32_t m_eff = ctl[CTL_M_EFF];
  bool ok = ctl[CTL_TICKET] == 0;
  for (uint32_t b = 0; b < ctl_cap; ++b) ok = ok && ctl[CTL_HEADER + b] == 0;
  if (!ok) printf("  control block not zeroed\n");
  std::vector<uint32_t> keys(m_eff), vals(m_eff);
  if (m_eff) {
    CK(hipMemcpy(keys.data(), d_keys, (size_t)m_eff * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(vals.data(), d_vals, (size_t)m_eff * 4, hipMemcpyDeviceToHost));
  }
  for (size_t e = 1; e < m_eff && ok; ++e)
    if (keys[e - 1] > keys[e]) { ok = false; printf("  keys out of order at %zu\n", e); }
  std::vector<uint64_t> exp;
  exp.reserve(m);
  for (uint32_t j = 0; j < cs.batch; ++j)
    for (size_t i = 0; i < n; ++i) cpu_pairs(&sc[8 * (j * n + i)], g, j, (uint32_t)i, 0, exp);
  if (exp.size() != m_eff) { ok = false; printf("  pair count %u, expected %zu\n", m_eff, exp.size()); }
  if (ok) {
    uint64_t s0 = 0, s1 = 0, x0 = 0, x1 = 0;
    for (size_t e = 0; e < m_eff; ++e) {
      const uint64_t pr = ((uint64_t)keys[e] << 32) | vals[e];
      s0 += pr * 0x9E3779B97F4A7C15ull; x0 ^= pr * 0xC2B2AE3D27D4EB4Full;
    }
    for (uint64_t pr : exp) { s1 += pr * 0x9E3779B97F4A7C15ull; x1 ^= pr * 0xC2B2AE3D27D4EB4Full; }
    if (s0 != s1 || x0 != x1) { ok = false; printf("  checksum mismatch\n"); }
    if (ok && full_check) {
      std::vector<uint64_t> got(m_eff);
      for (size_t e = 0; e < m_eff; ++e) got[e] = ((uint64_t)keys[e] << 32) | vals[e];
      std::sort(got.begin(), got.end());
      std::sort(exp.begin(), exp.end());
      if (got != exp) { ok = false; printf("  multiset mismatch\n"); }
    }
  }
  const double bytes = 2.0 * 32 * n * cs.batch + 8.0 * m_eff * 3;
  printf("%-34s n=%-9zu c=%-2u P=%-2u R=%-2u np=%-6u m_eff=%-10u hist %7.1f  scatter %7.1f  local %7.1f  sum %7.1f us  (%.2f TB/s)  %s\n",
         cs.name, n, g.c, g.pbits, g.rbits, g.np, m_eff, t0 * 1e3, t1 * 1e3, t2 * 1e3, (t0 + t1 + t2) * 1e3,
         bytes / ((t0 + t1 + t2) * 1e-3) / 1e12, ok ? "ok" : "FAILED");
  {
    unsigned long long dbg[3][16];
    CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(sort_dbg), sizeof dbg));
    printf("      one workgroup, us: scatter [load+reserve %.2f | scan %.2f | stage %.2f | deltas %.2f | copy-out %.2f]  local [count pass %.2f | a full tile: load %.2f | rank %.2f | scan %.2f | stage %.2f | copy-out %.2f]\n",
           (dbg[1][1] - dbg[1][0]) * 0.01, (dbg[1][2] - dbg[1][1]) * 0.01, (dbg[1][3] - dbg[1][2]) * 0.01, (dbg[1][4] - dbg[1][3]) * 0.01,
           (dbg[1][5] - dbg[1][4]) * 0.01, (dbg[2][7] - dbg[2][6]) * 0.01, (dbg[2][1] - dbg[2][0]) * 0.01, (dbg[2][2] - dbg[2][1]) * 0.01, (dbg[2][3] - dbg[2][2]) * 0.01,
           (dbg[2][4] - dbg[2][3]) * 0.01, (dbg[2][5] - dbg[2][4]) * 0.01);
    static unsigned long long span[3][4096][2];
    CK(hipMemcpyFromSymbol(span, HIP_SYMBOL(sort_span), sizeof span));
    printf("      hist workgroup, us: [load+digits %.2f | rows+atomics issue %.2f | drain %.2f]\n", (dbg[0][1] - dbg[0][0]) * 0.01,
           (dbg[0][2] - dbg[0][1]) * 0.01, (dbg[0][5] - dbg[0][2]) * 0.01);
    for (int kern = 0; kern <= 2; ++kern) {
      const uint32_t wgs = std::min<uint32_t>(4096, kern == 0 ? cs.batch * wgs_per_msm : kern == 1 ? tiles_total : g.np);
      unsigned long long t_min = ~0ull, t_max = 0;
      double busy = 0;
      for (uint32_t b = 0; b < wgs; ++b) {
        if (span[kern][b][1] <= span[kern][b][0]) continue;
        t_min = std::min(t_min, span[kern][b][0]);
        t_max = std::max(t_max, span[kern][b][1]);
        busy += (span[kern][b][1] - span[kern][b][0]) * 0.01;
      }
      // starts by quarter of the kernel's span
      int hist4[8] = {0};
      for (uint32_t b = 0; b < wgs; ++b) {
        if (span[kern][b][1] <= span[kern][b][0]) continue;
        int q = (int)((span[kern][b][0] - t_min) * 8 / (t_max - t_min + 1));
        hist4[q]++;
      }
      printf("      %s: first start -> last end %.2f us, mean workgroup %.2f us, sum %.1f us = %.2f per CU; starts per eighth:",
             kern == 0 ? "hist" : kern == 1 ? "scatter" : "local", (t_max - t_min) * 0.01, busy / wgs, busy, busy / 256);
      for (int q = 0; q < 8; ++q) printf(" %d", hist4[q]);
      printf("\n");
    }
  }
  fflush(stdout);
  for (auto& e : ev) CK(hipEventDestroy(e));
  CK(hipFree(d_canon)); CK(hipFree(d_rows)); CK(hipFree(d_sc)); CK(hipFree(d_pairs)); CK(hipFree(d_keys)); CK(hipFree(d_vals));
  return ok;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  std::vector<Case> cases = {
      {1, 5, false, 1, 0, "n=1"},
      {5, 5, false, 1, 0, "n=5"},
      {100, 8, false, 1, 0, "n=100 c=8"},
      {1000, 13, true, 1, 0, "n=1000 table c=13"},
      {3000, 4, false, 1, 0, "c=4, 64 windows"},
      {1000, 13, true, 1, 5, "n=1000 table c=13 carry runs"},
      {1000, 7, false, 1, 5, "n=1000 c=7 carry runs"},
      {4096, 8, false, 2, 0, "2^12 batch 2"},
      {100, 20, true, 1, 0, "n=100 table c=20"},
      {1 << 15, 13, true, 1, 0, "2^15 table c=13"},
      {1 << 17, 16, true, 1, 0, "2^17 table c=16 (shard)"},
      {1 << 17, 16, true, 1, 1, "2^17 all equal"},
      {1 << 17, 16, true, 1, 3, "2^17 all zero"},
      {1 << 18, 20, true, 1, 1, "2^18 c=20 all equal"},
      {1 << 18, 20, true, 1, 4, "2^18 c=20 30% equal"},
      {1 << 20, 16, false, 1, 0, "2^20 no table c=16"},
      {1 << 20, 20, false, 1, 0, "2^20 no table c=20"},
      {1 << 20, 20, true, 1, 0, "2^20 table c=20"},
      {1 << 20, 20, true, 1, 2, "2^20 witness-like"},
      {1 << 20, 20, true, 4, 0, "2^20 table c=20 batch 4"},
      {1 << 20, 22, true, 1, 0, "2^20 table c=22"},
  };
  if (!quick) {
    cases.push_back({1 << 22, 20, true, 1, 0, "2^22 table c=20"});
    cases.push_back({1 << 24, 20, true, 1, 0, "2^24 table c=20"});
    cases.push_back({1 << 24, 22, true, 1, 0, "2^24 table c=22"});
    cases.push_back({1 << 24, 24, true, 1, 0, "2^24 table c=24"});
  }
  bool all = true;
  const char* filter = argc > 2 ? argv[2] : nullptr;
  for (const Case& cs : cases) {
    if (filter && !strstr(cs.name, filter)) continue;
    const size_t m = cs.n * cs.batch * ((256 + cs.c - 1) / cs.c);
    all = run_case(cs, m > (1u << 26) ? 3 : 10, m <= (1u << 26)) && all;
  }
  printf(all ? "ALL OK\n" : "FAILURES\n");
  return all ? 0 : 1;
}
