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
  printf("asan driver done %llx\n", (unsigned long long)out[0]);
  free(a); free(b); free(pts); free(q);
  return 0;
}
