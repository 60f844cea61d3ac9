// Test infrastructure: drives each oracle entry point once so `make -C oracle sanitize`
// (ASan+UBSan, CPU) covers them.  Not part of the product path.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
void orc_init(void); void orc_fr_sample(u64, size_t, u64*); int orc_fr_ntt(u64*, unsigned, unsigned, int);
void orc_g1_bases_arith(const u64*, const u64*, size_t, u64*, int); int orc_g1_msm(const u64*, const u64*, size_t, unsigned, u64*, int);
void orc_fr_poly_ruffini(const u64*, size_t, const u64*, u64*); void orc_fr_prefix_product(const u64*, size_t, u64*); void orc_fr_batch_inverse(u64*, size_t);
void orc_fr_powers(const u64*, const u64*, size_t, u64*); void orc_fr_lincomb(unsigned, const u64* const*, const u64*, size_t, u64*, int);
void orc_fr_batch_inverse_trick(u64*, size_t, int);
void orc_plonk_perm_terms(const u64* const*, const u64* const*, const u64*, const u64*, const u64*, size_t, u64*, u64*, int);
void orc_plonk_quotient(const u64* const*, size_t, const u64*, const u64*, const u64*, u64*, int);
int main(void) {
  orc_init();
  size_t n = 1 << 10;
  u64 *a = malloc(32 * n), *b = malloc(32 * n);
  orc_fr_sample(1, n, a); memcpy(b, a, 32 * n);
  for (unsigned f = 0; f < 4; ++f) { orc_fr_ntt(a, 10, f, 1); orc_fr_ntt(a, 10, f, 4); }
  u64 k0[4] = {5,0,0,0}, d[4] = {7,0,0,0}, out[18];
  u64 *pts = malloc(96 * 300);
  orc_g1_bases_arith(k0, d, 300, pts, 2);
  orc_g1_msm(pts, b, 300, 0, out, 1); orc_g1_msm(pts, b, 300, 0, out, 4); orc_g1_msm(pts, b, 0, 0, out, 1);
  u64 *q = malloc(32 * n);
  orc_fr_poly_ruffini(b, n, b + 4, q); orc_fr_prefix_product(b, n, q); orc_fr_batch_inverse(q, n);
  {
    size_t m = 64;                                  /* prover-round functions on a 4m = 256-point coset */
    u64 *big = malloc(32 * 4 * m * 19), *o2 = malloc(32 * 4 * m), *o3 = malloc(32 * 4 * m);
    orc_fr_sample(9, 4 * m * 19, big);
    const u64 *ptrs[18];
    for (int j = 0; j < 18; ++j) ptrs[j] = big + (size_t)j * 16 * m;
    orc_plonk_quotient(ptrs, m, b, b + 4, b + 8, o2, 3);
    orc_plonk_perm_terms(ptrs, ptrs + 4, ptrs[8], b, b + 4, 4 * m, o2, o3, 3);
    orc_fr_lincomb(7, ptrs, b, 4 * m, o2, 2);
    orc_fr_powers(b, b + 4, 4 * m, o3);
    orc_fr_batch_inverse_trick(o3, 4 * m, 3);
    orc_fr_batch_inverse_trick(o3, 0, 3);
    free(big); free(o2); free(o3);
  }
  printf("asan driver done %llx\n", (unsigned long long)out[0]);
  free(a); free(b); free(pts); free(q);
  return 0;
}
