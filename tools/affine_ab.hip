// A/B for VERDICT r02 item 2(b): the FIRST tree level of the MSM's bucket accumulation as batched-affine additions
// against the XYZZ mixed addition the library uses -- on the access pattern of msm_accumulate_l1_kernel (pairs of
// random 96-byte rows of a 1.3 GB table of affine points), without any of the bucket bookkeeping, i.e. the most
// favourable setting for the affine form.
//
//   part 1  cost of one Fp inversion (binary GCD, csrc/field_inv.hip.h) in units of one Fp product, both as dependent
//           chains at two waves per SIMD
//   part 2  per pair (P, Q) of table rows:
//             XYZZ    acc = P; acc += Q                 (one madd: 8 M + 2 S with the merged reduction, r03)
//             affine  Montgomery's trick over k pairs per thread: prefix products of the x2 - x1 parked in global
//                     scratch (k x 14 limbs do not fit the register file), ONE inversion per thread, then
//                     lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1; the
//                     points are gathered a second time for the back-substitution (x only on the way up) and
//                     the sum is stored as a canonical affine point for the next level
//           both checked against each other on real curve points (the table repeats i G, i = 1 .. 4096).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 tools/affine_ab.hip -o tools/affine_ab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../plonk-prototype_amd/csrc/ec.hip.h"
#include "../plonk-prototype_amd/csrc/field_inv.hip.h"

using namespace pm;
#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

// ---- part 1: dependent chains
template <int OP>
__global__ void __launch_bounds__(128, 2) chain_kernel(u32x4* io, int reps) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp x = ld_fp_limbs(io + 4 * t);
  const Fp c = fe_pow2<FpP, 100>();
  for (int r = 0; r < reps; ++r) {
    if (OP == 0) x = fe_mul<FpP>(x, fe_add<FpP>(x, c));
    if (OP == 1) x = fe_add<FpP>(fe_inv_dev<FpP>(x), c);
    if (OP == 2) {   // Fermat: x^(p-2), square-and-multiply
      constexpr u32 E[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                             0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
      Fp inv = fe_one<FpP>(), base = x;
      for (int w = 0; w < 12; ++w)
        for (int bit = 0; bit < 32; ++bit) {
          if ((E[w] >> bit) & 1) inv = fe_mul<FpP>(inv, base);
          base = fe_sqr<FpP>(base);
        }
      x = fe_add<FpP>(inv, c);
    }
  }
  st_fp_limbs(io + 4 * t, fe_mul<FpP>(x, fe_one<FpP>()));
}

// ---- real curve points: row i = (i + 1) G, affine, device Montgomery form, canonical (the table format of msm.hip)
__global__ void make_points_kernel(u32x4* pts, u32 count, const Fp gx, const Fp gy) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Xyzz acc = xyzz_identity();
  const u32 k = i + 1;
  for (int bit = 31 - __clz(k); bit >= 0; --bit) {
    acc = xyzz_double(acc);
    if ((k >> bit) & 1) acc = xyzz_madd(acc, gx, gy);
  }
  const Fp zi = fe_inv_dev<FpP>(acc.zzz);                 // 1 / ZZZ
  const Fp t2 = fe_mul<FpP>(zi, acc.zz);                  // ZZ / ZZZ
  const Fp zzi = fe_sqr<FpP>(t2);                         // 1 / ZZ
  fe_store<FpP>(pts + 6 * i, fe_mul<FpP>(acc.x, zzi));
  fe_store<FpP>(pts + 6 * i + 3, fe_mul<FpP>(acc.y, zi));
}
__global__ void fill_table_kernel(const u32x4* pts, u32 count, u32x4* table, size_t rows) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const u32x4* s = pts + 6 * (size_t)((r * 2654435761ull) % count);
#pragma unroll
  for (int q = 0; q < 6; ++q) table[6 * r + q] = s[q];
}

__global__ void to_mont_kernel(Fp* v) { *v = fe_mul<FpP>(*v, fe_pow2<FpP, 2 * 392>()); }

// ---- part 2
__global__ void __launch_bounds__(128, 2) pairs_xyzz_kernel(const u32x4* table, const u32* idx, size_t npairs, u32 k, u32x4* sums,
                                                             u32* sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * k;
  if (lo >= npairs) return;
  const size_t hi = lo + k < npairs ? lo + k : npairs;
  u32 chk = 0;
  for (size_t i = lo; i < hi; ++i) {
    const u32x4* p = table + 6 * (size_t)idx[2 * i];
    const u32x4* q = table + 6 * (size_t)idx[2 * i + 1];
    Xyzz acc = xyzz_madd(xyzz_identity(), fe_load<FpP>(p), fe_load<FpP>(p + 3));
    acc = xyzz_madd(acc, fe_load<FpP>(q), fe_load<FpP>(q + 3));
    // what the accumulate keeps in registers is written here only so that the two variants can be compared
    if (sums) st_xyzz(sums, i, acc);
#pragma unroll
    for (int q = 0; q < 14; ++q) chk ^= acc.x.l[q] ^ acc.y.l[q] ^ acc.zz.l[q] ^ acc.zzz.l[q];
  }
  sink[t] = chk;   // keeps the additions alive when the sums are not stored
}
__global__ void __launch_bounds__(128, 2) pairs_affine_kernel(const u32x4* table, const u32* idx, size_t npairs, u32 k,
                                                               u32x4* prefix, u32x4* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * k;
  if (lo >= npairs) return;
  const size_t hi = lo + k < npairs ? lo + k : npairs;
  Fp acc = fe_one<FpP>();
  for (size_t i = lo; i < hi; ++i) {
    const Fp x1 = fe_load<FpP>(table + 6 * (size_t)idx[2 * i]), x2 = fe_load<FpP>(table + 6 * (size_t)idx[2 * i + 1]);
    u32 diff = 0;
#pragma unroll
    for (int q = 0; q < 14; ++q) diff |= x1.l[q] ^ x2.l[q];
    st_fp_limbs(prefix + 4 * i, acc);
    if (diff) acc = fe_mul<FpP>(acc, fe_norm<FpP>(fe_sub<FpP, 2, 1>(x2, x1)));   // equal x: P = +-Q, left to the slow path
  }
  Fp inv = fe_inv_dev<FpP>(acc);
  for (size_t i = hi; i-- > lo;) {
    const u32x4* p = table + 6 * (size_t)idx[2 * i];
    const u32x4* q = table + 6 * (size_t)idx[2 * i + 1];
    const Fp x1 = fe_load<FpP>(p), y1 = fe_load<FpP>(p + 3), x2 = fe_load<FpP>(q), y2 = fe_load<FpP>(q + 3);
    u32 diff = 0;
#pragma unroll
    for (int r = 0; r < 14; ++r) diff |= x1.l[r] ^ x2.l[r];
    if (!diff) {   // not handled here (doubling / identity): a zero record
      const u32x4 z = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < 6; ++r) out[6 * i + r] = z;
      continue;
    }
    const Fp dx = fe_norm<FpP>(fe_sub<FpP, 2, 1>(x2, x1));      // (1+, <3)
    const Fp dy = fe_norm<FpP>(fe_sub<FpP, 2, 1>(y2, y1));
    const Fp inv_i = fe_mul<FpP>(inv, ld_fp_limbs(prefix + 4 * i));
    inv = fe_mul<FpP>(inv, dx);
    const Fp lam = fe_mul<FpP>(dy, inv_i);
    const Fp l2 = fe_sqr<FpP>(lam);
    const Fp x3 = fe_norm<FpP>(fe_sub<FpP, 3, 1>(l2, fe_add<FpP>(x1, x2)));     // lambda^2 - x1 - x2 + 3 p   (1+, <5)
    const Fp y3 = fe_sub<FpP, 2, 1>(fe_mul<FpP>(lam, fe_norm<FpP>(fe_sub<FpP, 6, 1>(x1, x3))), y1);
    fe_store<FpP>(out + 6 * i, fe_mul<FpP>(x3, fe_one<FpP>()));
    fe_store<FpP>(out + 6 * i + 3, fe_mul<FpP>(y3, fe_one<FpP>()));
  }
}
// XYZZ sum -> canonical affine, to compare the two variants
__global__ void xyzz_to_affine_kernel(const u32x4* sums, size_t n, u32x4* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Xyzz p = ld_xyzz(sums, i);
  const Fp zi = fe_inv_dev<FpP>(p.zzz);
  const Fp t2 = fe_mul<FpP>(zi, p.zz);
  const Fp zzi = fe_sqr<FpP>(t2);
  fe_store<FpP>(out + 6 * i, fe_mul<FpP>(p.x, zzi));
  fe_store<FpP>(out + 6 * i + 3, fe_mul<FpP>(p.y, zi));
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char** argv) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // ---- part 1
  {
    const size_t threads = 256 * 4 * 2 * 64;
    u32x4* io;
    CK(hipMalloc(&io, threads * 64));
    std::vector<u32> init(threads * 16);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (i % 16 < 13) ? (u32)((i * 2654435761u) & 0xfffffffu) : 0u;
    const char* names[3] = {"Fp product (fe_mul)", "Fp inverse, binary GCD", "Fp inverse, x^(p-2)"};
    const int reps[3] = {2000, 40, 8};
    double per[3];
    for (int op = 0; op < 3; ++op) {
      CK(hipMemcpy(io, init.data(), init.size() * 4, hipMemcpyHostToDevice));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (op == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(threads / 128), dim3(128), 0, 0, io, reps[op]);
        if (op == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(threads / 128), dim3(128), 0, 0, io, reps[op]);
        if (op == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(threads / 128), dim3(128), 0, 0, io, reps[op]);
        CK(hipEventRecord(e1, 0));
        per[op] = time_ms(e0, e1) * 1e6 / reps[op];   // ns per op per wave-slot (all lanes in lock step)
      }
      printf("%-28s %9.1f ns per dependent op at two waves per SIMD  (%.1f products)\n", names[op], per[op], per[op] / per[0]);
    }
    CK(hipFree(io));
  }
  // ---- part 2
  const size_t rows = argc > 1 ? (size_t)atol(argv[1]) : (size_t)13 << 20;   // 2^20 points x 13 table rows
  const size_t npairs = rows / 2;
  const u32 real = 4096;
  u32x4 *pts, *table, *prefix, *out, *sums, *aff;
  u32 *idx, *sink;
  CK(hipMalloc(&sink, npairs * 4));
  CK(hipMalloc(&pts, (size_t)real * 96));
  CK(hipMalloc(&table, rows * 96));
  CK(hipMalloc(&idx, rows * 4));
  CK(hipMalloc(&prefix, npairs * 64));
  CK(hipMalloc(&out, npairs * 96));
  {
    // generator, device Montgomery form (x 2^392): canonical limbs -> fe_mul by 2^784
    const u64 GX[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                       0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
    const u64 GY[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                       0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
    // conversion on the device: a tiny kernel would do; here the host splits the integers into 28-bit limbs and the
    // make_points kernel receives them already multiplied by R' through one product with 2^784 inside a lambda kernel
    struct Conv {
      static Fp limbs(const u64* w) {
        Fp r;
        for (int i = 0; i < 14; ++i) {
          const int lo = 28 * i, j = lo / 64, sh = lo % 64;
          u64 v = w[j] >> sh;
          if (sh + 28 > 64 && j + 1 < 6) v |= w[j + 1] << (64 - sh);
          r.l[i] = (u32)(v & 0xfffffffu);
        }
        return r;
      }
    };
    Fp gx = Conv::limbs(GX), gy = Conv::limbs(GY);
    // to Montgomery form on the device: reuse chain-free helper kernel via fe_mul in make_points (gx, gy passed raw)
    // -> do it here with a one-thread kernel
    Fp* dg;
    CK(hipMalloc(&dg, 2 * sizeof(Fp)));
    CK(hipMemcpy(dg, &gx, sizeof(Fp), hipMemcpyHostToDevice));
    CK(hipMemcpy(dg + 1, &gy, sizeof(Fp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(to_mont_kernel, dim3(1), dim3(1), 0, 0, dg);
    hipLaunchKernelGGL(to_mont_kernel, dim3(1), dim3(1), 0, 0, dg + 1);
    CK(hipMemcpy(&gx, dg, sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&gy, dg + 1, sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipFree(dg));
    hipLaunchKernelGGL(make_points_kernel, dim3(real / 64), dim3(64), 0, 0, pts, real, gx, gy);
    hipLaunchKernelGGL(fill_table_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, pts, real, table, rows);
    CK(hipDeviceSynchronize());
  }
  {
    std::vector<u32> h(rows);
    u64 s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < rows; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      h[i] = (u32)((s >> 33) % rows);
    }
    CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
  }
  // correctness on the first 2^16 pairs
  {
    const size_t nchk = 1 << 16;
    CK(hipMalloc(&sums, nchk * 256));
    CK(hipMalloc(&aff, nchk * 96));
    hipLaunchKernelGGL(pairs_xyzz_kernel, dim3((unsigned)(nchk / 16 / 128)), dim3(128), 0, 0, table, idx, nchk, 16u, sums, sink);
    hipLaunchKernelGGL(xyzz_to_affine_kernel, dim3((unsigned)(nchk / 128)), dim3(128), 0, 0, sums, nchk, aff);
    hipLaunchKernelGGL(pairs_affine_kernel, dim3((unsigned)(nchk / 16 / 128)), dim3(128), 0, 0, table, idx, nchk, 16u, prefix, out);
    CK(hipDeviceSynchronize());
    std::vector<u32> a(nchk * 24), b(nchk * 24);
    CK(hipMemcpy(a.data(), aff, nchk * 96, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, nchk * 96, hipMemcpyDeviceToHost));
    size_t same = 0, skipped = 0, bad = 0;
    for (size_t i = 0; i < nchk; ++i) {
      bool zero = true, eq = true;
      for (int q = 0; q < 24; ++q) {
        zero = zero && b[24 * i + q] == 0;
        eq = eq && a[24 * i + q] == b[24 * i + q];
      }
      if (zero) ++skipped; else if (eq) ++same; else ++bad;
    }
    printf("affine sums vs XYZZ sums on %zu pairs of real curve points: %zu equal, %zu left to the slow path (equal x), %zu DIFFERENT\n",
           nchk, same, skipped, bad);
    CK(hipFree(sums));
    CK(hipFree(aff));
    if (bad) return 1;
  }
  printf("%zu pairs of random rows of a %.2f GB table:\n", npairs, rows * 96.0 / 1e9);
  for (u32 k : {16u, 32u, 64u, 128u, 256u}) {
    const size_t threads = (npairs + k - 1) / k;
    float tx = 0, ta = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(pairs_xyzz_kernel, dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, 0, table, idx, npairs, k, (u32x4*)nullptr, sink);
      CK(hipEventRecord(e1, 0));
      tx = time_ms(e0, e1);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(pairs_affine_kernel, dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, 0, table, idx, npairs, k, prefix, out);
      CK(hipEventRecord(e1, 0));
      ta = time_ms(e0, e1);
    }
    printf("  k = %3u pairs per thread (%7zu threads): XYZZ madd %.3f ms (%.1f ns per pair per CU-lane)  batched affine %.3f ms  -> affine / XYZZ = %.2f\n",
           k, threads, tx, tx * 1e6 / npairs, ta, ta / tx);
  }
  CK(hipGetLastError());
  return 0;
}
