// r04: what one two-lane group addition (ec.hip.h: half_add) costs in ISOLATION -- registers only, no loads, no branches around it --
// with one, two and four waves per SIMD, against the ~18.2 K shader cycles per addition of msm_bucket_reduce_kernel's bucket loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 tools/half_add_bench.hip -o tools/half_add_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../plonk-prototype_amd/csrc/ec.hip.h"

using namespace pm;

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) bench_kernel(const u32* seed, u32 iters, u32 mode, u32* sink, unsigned long long* cyc) {
  extern __shared__ u32 pad[];   // the LDS request keeps a second workgroup off the CU
  const bool isB = threadIdx.x & 1;
  Half a, b;
  for (int i = 0; i < 14; ++i) {   // random field elements (not curve points: the formulas do not care, ZZ3 = 0 has probability ~0)
    a.c0.l[i] = seed[(threadIdx.x * 61 + i) & 1023] & 0xfffffffu;
    a.c1.l[i] = seed[(threadIdx.x * 67 + i + 100) & 1023] & 0xfffffffu;
    b.c0.l[i] = seed[(threadIdx.x * 71 + i + 200) & 1023] & 0xfffffffu;
    b.c1.l[i] = seed[(threadIdx.x * 73 + i + 300) & 1023] & 0xfffffffu;
  }
  a.inf = 0;
  b.inf = 0;
  Half run = a, sum = b;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {   // one chain: acc += b
#pragma unroll 1
    for (u32 i = 0; i < iters; ++i) run = half_add(run, b, isB);
  } else if (mode == 3) {   // as mode 1, the two waves of a SIMD (waves w and w + 4 of the workgroup) taking turns at the higher priority
#pragma unroll 1
    for (u32 i = 0; i < iters; ++i) {
      if ((i + (threadIdx.x >> 8)) & 1u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
      run = half_add(run, b, isB);
      sum = half_add(sum, run, isB);
    }
    __builtin_amdgcn_s_setprio(0);
  } else {           // the bucket loop's shape: running += b; sum += running
#pragma unroll 1
    for (u32 i = 0; i < iters; ++i) {
      run = half_add(run, b, isB);
      sum = half_add(sum, run, isB);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  u32 x = 0;
  for (int i = 0; i < 14; ++i) x ^= run.c0.l[i] ^ run.c1.l[i] ^ sum.c0.l[i] ^ sum.c1.l[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int WAVES>
static void run(const u32* d_seed, u32 iters, u32 mode, const char* what) {
  const int wgs = 256;
  u32* sink;
  unsigned long long* cyc;
  hipMalloc(&sink, wgs * 64 * WAVES * 4);
  hipMalloc(&cyc, wgs * WAVES * 8);
  hipFuncSetAttribute((const void*)bench_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench_kernel<WAVES>, dim3(wgs), dim3(64 * WAVES), 96 * 1024, 0, d_seed, iters, mode, sink, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * WAVES);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0, mx = 0;
  for (auto v : h) {
    s += (double)v;
    mx = mx > (double)v ? mx : (double)v;
  }
  const double adds = (mode == 0 ? 1.0 : 2.0) * iters;
  printf("%-34s waves/SIMD %d: mean %.0f cycles per addition per wave (slowest wave %.0f) -> SIMD spends %.0f cycles per addition\n", what, WAVES / 4,
         s / h.size() / adds, mx / adds, s / h.size() / adds / (WAVES / 4));
  hipFree(sink);
  hipFree(cyc);
}

int main() {
  std::vector<u32> seed(1024);
  unsigned long long st = 88172645463325252ull;
  for (auto& v : seed) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    v = (u32)st;
  }
  u32* d_seed;
  hipMalloc(&d_seed, 4096);
  hipMemcpy(d_seed, seed.data(), 4096, hipMemcpyHostToDevice);
  run<4>(d_seed, 64, 0, "acc += b");
  run<8>(d_seed, 64, 0, "acc += b");
  run<4>(d_seed, 32, 1, "running += b; sum += running");
  run<8>(d_seed, 32, 1, "running += b; sum += running");
  run<8>(d_seed, 32, 3, "... with alternating s_setprio");
  return 0;
}
