// Instruction-throughput microbenchmark for the integer paths the Fr/Fp Montgomery
// kernels depend on (gfx950).  Prints cycles per wave-instruction per SIMD assuming
// the clock reported by the runtime.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef unsigned long long u64; typedef uint32_t u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
#define ITER 512
#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

template<int KIND> __global__ void __launch_bounds__(256) k(u32* out, u32 seed) {
  u32 tid = threadIdx.x + blockIdx.x*blockDim.x;
  u32 a = tid*2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 c0 = a, c1 = b, c2 = a+b, c3 = a-b; u32 o0=0,o1=0,o2=0,o3=0;
  double d0 = (double)a, d1 = (double)b, d2 = 1.5, d3 = 2.5, dm = 1.0000001;
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) { // independent v_mad_u64_u32 (4 chains)
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == 1) { // dependent v_mad_u64_u32 (1 chain) x64
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0":"+v"(c0):"v"(a),"v"(b):"vcc");)
    } else if (KIND == 2) { // mac = mad + addc, 4 chains
      REP16(asm volatile(
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
        "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32_e32 %7, vcc, 0, %7, vcc"
        :"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3),"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == 3) { // v_mul_lo_u32 x4
      REP16(asm volatile("v_mul_lo_u32 %0, %4, %0\n\tv_mul_lo_u32 %1, %4, %1\n\tv_mul_lo_u32 %2, %4, %2\n\tv_mul_lo_u32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == 4) { // v_mul_hi_u32 x4
      REP16(asm volatile("v_mul_hi_u32 %0, %4, %0\n\tv_mul_hi_u32 %1, %4, %1\n\tv_mul_hi_u32 %2, %4, %2\n\tv_mul_hi_u32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == 5) { // v_add_co + v_addc chain x4
      REP16(asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n\tv_addc_co_u32_e32 %1, vcc, %4, %1, vcc\n\tv_addc_co_u32_e32 %2, vcc, %4, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %4, %3, vcc":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a):"vcc");)
    } else if (KIND == 6) { // v_fma_f64 x4 independent
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %4, %1\n\tv_fma_f64 %2, %2, %4, %2\n\tv_fma_f64 %3, %3, %4, %3":"+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3):"v"(dm));)
    } else if (KIND == 7) { // v_mad_u32_u24 x4
      REP16(asm volatile("v_mad_u32_u24 %0, %4, %0, %5\n\tv_mad_u32_u24 %1, %4, %1, %5\n\tv_mad_u32_u24 %2, %4, %2, %5\n\tv_mad_u32_u24 %3, %4, %3, %5":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b));)
    } else if (KIND == 8) { // v_mul_hi_u32_u24 x4
      REP16(asm volatile("v_mul_hi_u32_u24 %0, %4, %0\n\tv_mul_hi_u32_u24 %1, %4, %1\n\tv_mul_hi_u32_u24 %2, %4, %2\n\tv_mul_hi_u32_u24 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == 9) { // v_mov_b32 x4
      REP16(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == 10) { // v_mad_u64_u32 with SGPR multiplicand, 4 chains
      u32 sb = __builtin_amdgcn_readfirstlane(b);
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"s"(sb):"vcc");)
    } else if (KIND == 11) { // v_mul_f64
      REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4":"+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3):"v"(dm));)
    } else if (KIND == 12) { // v_add_f64
      REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4":"+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3):"v"(dm));)
    } else if (KIND == 13) { // v_lshl_add_u64 (64-bit add)
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(c0));)
    } else if (KIND == 14) { // v_add3_u32
      REP16(asm volatile("v_add3_u32 %0, %0, %4, %1\n\tv_add3_u32 %1, %1, %4, %2\n\tv_add3_u32 %2, %2, %4, %3\n\tv_add3_u32 %3, %3, %4, %0":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == 15) { // v_mad_i64_i32
      REP16(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %1, vcc, %4, %5, %1\n\tv_mad_i64_i32 %2, vcc, %4, %5, %2\n\tv_mad_i64_i32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    }
  }
  out[tid] = (u32)c0 ^ (u32)c1 ^ (u32)c2 ^ (u32)c3 ^ o0 ^ o1 ^ o2 ^ o3 ^ (u32)d0 ^ (u32)d1 ^ (u32)d2 ^ (u32)d3;
}

template<int KIND> int run(const char* name, int instr_per_rep, int waves_per_simd, double clock_hz, int cus, u32* d_out) {
  int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<KIND><<<blocks, 256>>>(d_out, 1); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0)); k<KIND><<<blocks, 256>>>(d_out, r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double winstr_per_simd = (double)ITER * 16 * instr_per_rep * waves_per_simd;
  double cyc = best * 1e-3 * clock_hz / winstr_per_simd;
  printf("%-34s waves/SIMD=%d  time=%8.3f ms  cycles/wave-instr/SIMD=%6.2f\n", name, waves_per_simd, best, cyc);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  double clk = p.clockRate * 1e3; int cus = p.multiProcessorCount;
  printf("device %s CUs=%d clock=%.0f MHz\n", p.name, cus, clk/1e6);
  u32* d; CK(hipMalloc(&d, 256 * cus * 8 * sizeof(u32) * 2));
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32 indep x4", 4, w, clk, cus, d);
    run<1>("v_mad_u64_u32 dependent", 4, w, clk, cus, d);
    run<2>("mac (mad_u64+addc) x4 [2 instr]", 8, w, clk, cus, d);
    run<10>("v_mad_u64_u32 sgpr operand", 4, w, clk, cus, d);
    run<15>("v_mad_i64_i32", 4, w, clk, cus, d);
    run<3>("v_mul_lo_u32", 4, w, clk, cus, d);
    run<4>("v_mul_hi_u32", 4, w, clk, cus, d);
    run<5>("v_add_co/addc chain", 4, w, clk, cus, d);
    run<13>("v_lshl_add_u64", 4, w, clk, cus, d);
    run<14>("v_add3_u32", 4, w, clk, cus, d);
    run<9>("v_mov_b32", 4, w, clk, cus, d);
    run<7>("v_mad_u32_u24", 4, w, clk, cus, d);
    run<8>("v_mul_hi_u32_u24", 4, w, clk, cus, d);
    run<6>("v_fma_f64", 4, w, clk, cus, d);
    run<11>("v_mul_f64", 4, w, clk, cus, d);
    run<12>("v_add_f64", 4, w, clk, cus, d);
  }
  return 0;
}
