// Stand-alone sweep of rocPRIM onesweep configurations for the MSM pair sort (u32 key + u32 value):
// hipcc -O3 --offload-arch=gfx950 -DPMRB=<radix bits> -DPMBLK=<block> -DPMIPT=<items per thread> sort_sweep.hip -o sort_sweep
// ./sort_sweep <key bits> [million pairs]   -- results: profiles/r02_sort_sweep.txt
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>
#ifndef PMRB
#define PMRB 10
#endif
#ifndef PMBLK
#define PMBLK 1024
#endif
#ifndef PMIPT
#define PMIPT 8
#endif
using ocfg = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<PMBLK, PMIPT>, rocprim::kernel_config<PMBLK, PMIPT>, PMRB,
                                                rocprim::block_radix_rank_algorithm::match>;
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, ocfg, 1024 * 1024>;
int main(int argc, char** argv) {
  const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 13) << 20;
  const int bits = argc > 1 ? atoi(argv[1]) : 20;
  std::vector<unsigned> hk(n), hv(n);
  unsigned s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hk[i] = (s >> 8) & ((1u << bits) - 1); hv[i] = (unsigned)i; }
  unsigned *k0, *k1, *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
  size_t tmp = 0;
  void* d_tmp = nullptr;
  for (int variant = 0; variant < 2; ++variant) {
    auto run = [&](void* t, size_t& sz) {
      return variant == 0 ? rocprim::radix_sort_pairs(t, sz, k0, k1, v0, v1, n, 0, bits, 0)
                          : rocprim::radix_sort_pairs<cfg>(t, sz, k0, k1, v0, v1, n, 0, bits, 0);
    };
    tmp = 0;
    if (run(nullptr, tmp) != hipSuccess) { printf("size query failed\n"); return 1; }
    hipMalloc(&d_tmp, tmp);
    for (int i = 0; i < 3; ++i) run(d_tmp, tmp);
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    const int reps = 20;
    for (int i = 0; i < reps; ++i) run(d_tmp, tmp);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
    std::vector<unsigned> ok(n);
    hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost);
    bool sorted = std::is_sorted(ok.begin(), ok.end());
    printf("%s bits=%d radix=%d block=%d ipt=%d: %.1f us  sorted=%d  tmp=%zu MB\n", variant ? "custom " : "default", bits, PMRB, PMBLK, PMIPT, us, (int)sorted, tmp >> 20);
    hipFree(d_tmp);
  }
  return 0;
}
