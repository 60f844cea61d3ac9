// Instruction-issue microbenchmark for the integer paths the Fr/Fp Montgomery kernels depend on (gfx950).
//
// r05 rewrite.  The r01 version timed whole launches with hipEvents and converted with the
// NOMINAL 2.4 GHz clock; launches were 0.1-0.3 ms, so launch latency and the chip's real clock (2.0-2.2 GHz under
// integer load, r04 stamps) both sat inside the "cycles" it printed.  This version stamps every wave with
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the timed loop and reports
//     cycles per wave-instruction per SIMD = mean wave lifetime in shader cycles / (instructions per wave x waves per SIMD)
// plus the clock the run really had.  Waves per SIMD are PLACED: one workgroup of 256 x min(W, 4) threads per CU
// (x 2 workgroups for W = 8), a dynamic-LDS request keeps any further workgroup off the CU.
//
// Rows: every opcode of fe_mul<FrP>'s compiled body (156 v_mad_u64_u32 : 24 v_lshl_add_u64 : 16 v_lshrrev_b64 :
// 17 v_and_b32 : 12 v_mov_b32 : 9 v_sub_u32), their candidate 32-bit replacements, a mixed row in those
// proportions, and mad/cheap-op interleavings (does a 2-cycle op hide behind a 4-cycle one?).
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef unsigned long long u64; typedef uint32_t u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
#define ITER 256
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Kind {
  K_MAD_IND, K_MAD_DEP, K_MAD_SGPR, K_MADI, K_MUL_LO, K_MUL_HI, K_MUL_U24, K_MAD_U24,
  K_AND, K_AND_LIT, K_SUB, K_ADD, K_XOR, K_LSHR32, K_ASHR32, K_LSHR64, K_ASHR64, K_LSHL64, K_ALIGNBIT, K_BFE, K_AND_OR, K_LSHL_ADD32,
  K_LSHL_ADD64, K_ADD3, K_BFI, K_MOV, K_MOV_DPP, K_ADDCO, K_ADDC_CHAIN, K_CNDMASK, K_PERM,
  K_MIX_FRMUL, K_MAD_AND_1_1, K_MAD_AND_2_1, K_MAD_MOV_1_1, K_MAD_SHR64_1_1, K_MAD_ALIGN_1_1, K_AND_MOV_1_1, K_MAD_LSHLADD64_1_1,
  K_COUNT
};

// every asm block of a row is 4 instructions (rows that mix say so); REP16 -> 64 instructions per loop body
template <int KIND> __global__ void __launch_bounds__(1024) k(u32* out, u64* stamps, u32 seed) {
  extern __shared__ u32 lds_dummy[];
  u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  u32 a = tid * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 c0 = a, c1 = b, c2 = a + b, c3 = a - b; u32 o0 = a, o1 = b, o2 = a * 3, o3 = b * 5;
  u32 sb = __builtin_amdgcn_readfirstlane(b) | 1u;
  if (seed == 0xffffffffu) lds_dummy[threadIdx.x] = a;  // keep the LDS request alive
  __syncthreads();
  const u64 t0 = __builtin_readcyclecounter();
  const u64 r0 = wall_clock64();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == K_MAD_IND) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_DEP) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0":"+v"(c0):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_SGPR) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"s"(sb):"vcc");)
    } else if (KIND == K_MADI) {
      REP16(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %1, vcc, %4, %5, %1\n\tv_mad_i64_i32 %2, vcc, %4, %5, %2\n\tv_mad_i64_i32 %3, vcc, %4, %5, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MUL_LO) {
      REP16(asm volatile("v_mul_lo_u32 %0, %4, %0\n\tv_mul_lo_u32 %1, %4, %1\n\tv_mul_lo_u32 %2, %4, %2\n\tv_mul_lo_u32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_MUL_HI) {
      REP16(asm volatile("v_mul_hi_u32 %0, %4, %0\n\tv_mul_hi_u32 %1, %4, %1\n\tv_mul_hi_u32 %2, %4, %2\n\tv_mul_hi_u32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_MUL_U24) {
      REP16(asm volatile("v_mul_u32_u24_e32 %0, %4, %0\n\tv_mul_u32_u24_e32 %1, %4, %1\n\tv_mul_u32_u24_e32 %2, %4, %2\n\tv_mul_u32_u24_e32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_MAD_U24) {
      REP16(asm volatile("v_mad_u32_u24 %0, %4, %0, %5\n\tv_mad_u32_u24 %1, %4, %1, %5\n\tv_mad_u32_u24 %2, %4, %2, %5\n\tv_mad_u32_u24 %3, %4, %3, %5":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b));)
    } else if (KIND == K_AND) {
      REP16(asm volatile("v_and_b32_e32 %0, %4, %0\n\tv_and_b32_e32 %1, %4, %1\n\tv_and_b32_e32 %2, %4, %2\n\tv_and_b32_e32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_AND_LIT) {
      REP16(asm volatile("v_and_b32_e32 %0, 0x1fffffff, %0\n\tv_and_b32_e32 %1, 0x1fffffff, %1\n\tv_and_b32_e32 %2, 0x1fffffff, %2\n\tv_and_b32_e32 %3, 0x1fffffff, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_SUB) {
      REP16(asm volatile("v_sub_u32_e32 %0, %4, %0\n\tv_sub_u32_e32 %1, %4, %1\n\tv_sub_u32_e32 %2, %4, %2\n\tv_sub_u32_e32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_ADD) {
      REP16(asm volatile("v_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %2, %4, %2\n\tv_add_u32_e32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_XOR) {
      REP16(asm volatile("v_xor_b32_e32 %0, %4, %0\n\tv_xor_b32_e32 %1, %4, %1\n\tv_xor_b32_e32 %2, %4, %2\n\tv_xor_b32_e32 %3, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_LSHR32) {
      REP16(asm volatile("v_lshrrev_b32_e32 %0, 1, %0\n\tv_lshrrev_b32_e32 %1, 1, %1\n\tv_lshrrev_b32_e32 %2, 1, %2\n\tv_lshrrev_b32_e32 %3, 1, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_ASHR32) {
      REP16(asm volatile("v_ashrrev_i32_e32 %0, 1, %0\n\tv_ashrrev_i32_e32 %1, 1, %1\n\tv_ashrrev_i32_e32 %2, 1, %2\n\tv_ashrrev_i32_e32 %3, 1, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_LSHR64) {
      REP16(asm volatile("v_lshrrev_b64 %0, 29, %0\n\tv_lshrrev_b64 %1, 29, %1\n\tv_lshrrev_b64 %2, 29, %2\n\tv_lshrrev_b64 %3, 29, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3));)
    } else if (KIND == K_ASHR64) {
      REP16(asm volatile("v_ashrrev_i64 %0, 29, %0\n\tv_ashrrev_i64 %1, 29, %1\n\tv_ashrrev_i64 %2, 29, %2\n\tv_ashrrev_i64 %3, 29, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3));)
    } else if (KIND == K_LSHL64) {
      REP16(asm volatile("v_lshlrev_b64 %0, 3, %0\n\tv_lshlrev_b64 %1, 3, %1\n\tv_lshlrev_b64 %2, 3, %2\n\tv_lshlrev_b64 %3, 3, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3));)
    } else if (KIND == K_ALIGNBIT) {
      REP16(asm volatile("v_alignbit_b32 %0, %4, %0, 29\n\tv_alignbit_b32 %1, %4, %1, 29\n\tv_alignbit_b32 %2, %4, %2, 29\n\tv_alignbit_b32 %3, %4, %3, 29":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_BFE) {
      REP16(asm volatile("v_bfe_u32 %0, %0, 1, 29\n\tv_bfe_u32 %1, %1, 1, 29\n\tv_bfe_u32 %2, %2, 1, 29\n\tv_bfe_u32 %3, %3, 1, 29":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_AND_OR) {
      REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b));)
    } else if (KIND == K_LSHL_ADD32) {
      REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\tv_lshl_add_u32 %1, %1, 1, %4\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_lshl_add_u32 %3, %3, 1, %4":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_LSHL_ADD64) {
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(c0));)
    } else if (KIND == K_ADD3) {
      REP16(asm volatile("v_add3_u32 %0, %0, %4, %1\n\tv_add3_u32 %1, %1, %4, %2\n\tv_add3_u32 %2, %2, %4, %3\n\tv_add3_u32 %3, %3, %4, %0":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_BFI) {
      REP16(asm volatile("v_bfi_b32 %0, %0, %4, %5\n\tv_bfi_b32 %1, %1, %4, %5\n\tv_bfi_b32 %2, %2, %4, %5\n\tv_bfi_b32 %3, %3, %4, %5":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b));)
    } else if (KIND == K_MOV) {
      REP16(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_MOV_DPP) {
      REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3));)
    } else if (KIND == K_ADDCO) {
      REP16(asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n\tv_add_co_u32_e32 %1, vcc, %4, %1\n\tv_add_co_u32_e32 %2, vcc, %4, %2\n\tv_add_co_u32_e32 %3, vcc, %4, %3":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a):"vcc");)
    } else if (KIND == K_ADDC_CHAIN) {
      REP16(asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n\tv_addc_co_u32_e32 %1, vcc, %4, %1, vcc\n\tv_addc_co_u32_e32 %2, vcc, %4, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %4, %3, vcc":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a):"vcc");)
    } else if (KIND == K_CNDMASK) {
      REP16(asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n\tv_cndmask_b32_e32 %1, %4, %1, vcc\n\tv_cndmask_b32_e32 %2, %4, %2, vcc\n\tv_cndmask_b32_e32 %3, %4, %3, vcc":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a):"vcc");)
    } else if (KIND == K_PERM) {
      REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %4, %5\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %4, %5":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a),"v"(b));)
    } else if (KIND == K_MIX_FRMUL) {
      // 39 instructions = 26 mad : 4 lshl_add_u64 : 3 lshrrev_b64 : 3 and : 2 mov : 1 sub  (x6 = 156 : 24 : 18 : 18 : 12 : 6,
      // fe_mul<FrP> compiles to 156 : 24 : 16 : 17 : 12 : 9); dependencies as in the product: one chain per accumulator
      REP4(asm volatile(
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_lshl_add_u64 %2, %0, 0, %2\n\tv_sub_u32_e32 %5, 0, %6\n\tv_and_b32_e32 %5, 0x1fffffff, %5\n\tv_mov_b32 %7, 0\n\t"
        "v_lshl_add_u64 %2, %4, 0, %2\n\tv_lshrrev_b64 %2, 29, %2\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
        "v_lshl_add_u64 %2, %1, 0, %2\n\tv_and_b32_e32 %6, 0x1fffffff, %5\n\tv_lshrrev_b64 %2, 29, %2\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
        "v_lshl_add_u64 %2, %0, 0, %2\n\tv_and_b32_e32 %7, 0x1fffffff, %6\n\tv_mov_b32 %3, %7\n\tv_lshrrev_b64 %2, 29, %2"
        :"+v"(c0),"+v"(c1),"+v"(c2),"+v"(o3),"+v"(c3),"+v"(o0),"+v"(o1),"+v"(o2):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_AND_1_1) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_and_b32_e32 %2, %4, %2\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_and_b32_e32 %3, %4, %3":"+v"(c0),"+v"(c1),"+v"(o2),"+v"(o3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_AND_2_1) {  // 6 instructions per block: accounted for in main()
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_and_b32_e32 %2, %4, %2\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_and_b32_e32 %3, %4, %3":"+v"(c0),"+v"(c1),"+v"(o2),"+v"(o3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_MOV_1_1) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mov_b32 %2, %3\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mov_b32 %3, %2":"+v"(c0),"+v"(c1),"+v"(o2),"+v"(o3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_SHR64_1_1) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_lshrrev_b64 %2, 29, %2\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_lshrrev_b64 %3, 29, %3":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_MAD_ALIGN_1_1) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_alignbit_b32 %2, %4, %2, 29\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_alignbit_b32 %3, %4, %3, 29":"+v"(c0),"+v"(c1),"+v"(o2),"+v"(o3):"v"(a),"v"(b):"vcc");)
    } else if (KIND == K_AND_MOV_1_1) {
      REP16(asm volatile("v_and_b32_e32 %0, %4, %0\n\tv_mov_b32 %2, %3\n\tv_and_b32_e32 %1, %4, %1\n\tv_mov_b32 %3, %2":"+v"(o0),"+v"(o1),"+v"(o2),"+v"(o3):"v"(a));)
    } else if (KIND == K_MAD_LSHLADD64_1_1) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_lshl_add_u64 %3, %3, 0, %2":"+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3):"v"(a),"v"(b):"vcc");)
    }
  }
  const u64 t1 = __builtin_readcyclecounter();
  const u64 r1 = wall_clock64();
  out[tid] = (u32)c0 ^ (u32)c1 ^ (u32)c2 ^ (u32)c3 ^ o0 ^ o1 ^ o2 ^ o3 ^ (u32)(c0 >> 32) ^ (u32)(c1 >> 32) ^ (u32)(c2 >> 32) ^ (u32)(c3 >> 32);
  if ((threadIdx.x & 63) == 0) {
    const u32 w = tid >> 6;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

struct Row { const char* name; int kind; int instr_per_body; };  // instructions per loop body (per wave)

template <int KIND> int run(const char* name, int instr_per_body, int wps, int cus, u32* d_out, u64* d_st, std::vector<u64>& h_st) {
  const int wpb = wps < 4 ? wps : 4;            // waves per SIMD supplied by ONE workgroup
  const int bpc = wps / wpb;                    // workgroups per CU
  const int threads = 256 * wpb, blocks = cus * bpc;
  const size_t lds = (bpc == 1) ? 100 * 1024 : 70 * 1024;  // 160 KB per CU: exactly bpc workgroups fit
  static bool attr_set[K_COUNT] = {};
  if (!attr_set[KIND]) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); attr_set[KIND] = true; }
  double best_cyc = 1e30, best_clk = 0, best_max = 0;
  for (int r = 0; r < 4; ++r) {
    k<KIND><<<blocks, threads, lds>>>(d_out, d_st, (u32)r + 1);
    CK(hipDeviceSynchronize());
    const int nw = blocks * threads / 64;
    CK(hipMemcpy(h_st.data(), d_st, sizeof(u64) * 2 * nw, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0, mx = 0;
    for (int w = 0; w < nw; ++w) { sc += (double)h_st[2 * w]; sr += (double)h_st[2 * w + 1]; if ((double)h_st[2 * w] > mx) mx = (double)h_st[2 * w]; }
    const double mean_c = sc / nw, mean_r = sr / nw;            // shader cycles, 100 MHz ticks
    const double cyc = mean_c / ((double)ITER * instr_per_body * wps);
    if (r > 0 && cyc < best_cyc) { best_cyc = cyc; best_clk = mean_c / (mean_r * 10e-9) / 1e9; best_max = mx / ((double)ITER * instr_per_body * wps); }
  }
  printf("%-38s waves/SIMD=%d  cycles/wave-instr/SIMD=%6.2f (slowest wave %6.2f)  clock=%.2f GHz\n", name, wps, best_cyc, best_max, best_clk);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s CUs=%d nominal clock=%.0f MHz; cycles are s_memtime shader cycles per wave, clock = s_memtime / s_memrealtime\n", p.gcnArchName, cus, p.clockRate / 1e3);
  u32* d; CK(hipMalloc(&d, sizeof(u32) * 1024 * 2 * cus));
  u64* st; CK(hipMalloc(&st, sizeof(u64) * 2 * 32 * cus));
  std::vector<u64> h(2 * 32 * cus);
  for (int w : {1, 2, 4, 8}) {
#define ROW(K, NAME, IPB) if (run<K>(NAME, IPB, w, cus, d, st, h)) return 1;
    ROW(K_MAD_IND, "v_mad_u64_u32 indep x4", 64)
    ROW(K_MAD_DEP, "v_mad_u64_u32 dependent", 64)
    ROW(K_MAD_SGPR, "v_mad_u64_u32 sgpr operand", 64)
    ROW(K_MADI, "v_mad_i64_i32", 64)
    ROW(K_MUL_LO, "v_mul_lo_u32", 64)
    ROW(K_MUL_HI, "v_mul_hi_u32", 64)
    ROW(K_MUL_U24, "v_mul_u32_u24_e32", 64)
    ROW(K_MAD_U24, "v_mad_u32_u24", 64)
    ROW(K_AND, "v_and_b32_e32", 64)
    ROW(K_AND_LIT, "v_and_b32_e32 literal", 64)
    ROW(K_SUB, "v_sub_u32_e32", 64)
    ROW(K_ADD, "v_add_u32_e32", 64)
    ROW(K_XOR, "v_xor_b32_e32", 64)
    ROW(K_LSHR32, "v_lshrrev_b32_e32", 64)
    ROW(K_ASHR32, "v_ashrrev_i32_e32", 64)
    ROW(K_LSHR64, "v_lshrrev_b64", 64)
    ROW(K_ASHR64, "v_ashrrev_i64", 64)
    ROW(K_LSHL64, "v_lshlrev_b64", 64)
    ROW(K_ALIGNBIT, "v_alignbit_b32", 64)
    ROW(K_BFE, "v_bfe_u32", 64)
    ROW(K_AND_OR, "v_and_or_b32", 64)
    ROW(K_LSHL_ADD32, "v_lshl_add_u32", 64)
    ROW(K_LSHL_ADD64, "v_lshl_add_u64", 64)
    ROW(K_ADD3, "v_add3_u32", 64)
    ROW(K_BFI, "v_bfi_b32", 64)
    ROW(K_PERM, "v_perm_b32", 64)
    ROW(K_MOV, "v_mov_b32", 64)
    ROW(K_MOV_DPP, "v_mov_b32_dpp quad_perm", 64)
    ROW(K_ADDCO, "v_add_co_u32_e32 (writes vcc)", 64)
    ROW(K_ADDC_CHAIN, "v_add_co/v_addc_co chain", 64)
    ROW(K_CNDMASK, "v_cndmask_b32_e32 (reads vcc)", 64)
    ROW(K_MIX_FRMUL, "mix 26 mad:4 add64:3 shr64:3 and:2 mov:1 sub", 39 * 4)
    ROW(K_MAD_AND_1_1, "mad : and  1:1", 64)
    ROW(K_MAD_AND_2_1, "mad : and  2:1", 96)
    ROW(K_MAD_MOV_1_1, "mad : mov  1:1", 64)
    ROW(K_MAD_SHR64_1_1, "mad : lshrrev_b64  1:1", 64)
    ROW(K_MAD_ALIGN_1_1, "mad : alignbit  1:1", 64)
    ROW(K_MAD_LSHLADD64_1_1, "mad : lshl_add_u64  1:1", 64)
    ROW(K_AND_MOV_1_1, "and : mov  1:1", 64)
  }
  return 0;
}
