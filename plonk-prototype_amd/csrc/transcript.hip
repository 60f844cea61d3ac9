// Host-only: the Keccak-f[1600] permutation under the prover's Merlin/STROBE-128 transcript
// (plonk-prototype_amd/transcript.py).  merlin 2.x is a dependency of
// dusk-plonk 0.8.2 (ref:Cargo.toml:19), not in the reference tree; this is FIPS 202's permutation.
// A proof hashes ~1 KB, i.e. a dozen permutations: no device work.
#include <cstdint>
#include <cstring>

#include "../../include/plonk_mi355x.h"

namespace {
const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// rotation offsets r[x][y]
const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
inline uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
}  // namespace

extern "C" void pm_keccak_f1600(uint8_t state[200]) {
  uint64_t a[5][5], b[5][5], c[5], d[5];   // a[x][y], lane (x, y) at byte 8 (x + 5 y), little-endian
  for (int x = 0; x < 5; ++x)
    for (int y = 0; y < 5; ++y) memcpy(&a[x][y], state + 8 * (x + 5 * y), 8);
  for (int round = 0; round < 24; ++round) {
    for (int x = 0; x < 5; ++x) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) b[y][(2 * x + 3 * y) % 5] = rol(a[x][y] ^ d[x], ROT[x][y]);
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y]);
    a[0][0] ^= RC[round];
  }
  for (int x = 0; x < 5; ++x)
    for (int y = 0; y < 5; ++y) memcpy(state + 8 * (x + 5 * y), &a[x][y], 8);
}
