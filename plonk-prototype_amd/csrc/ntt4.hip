// Instantiations of the radix-4 pass kernels (own translation unit so that it compiles in
// parallel with ntt.hip) and their lookup / table builder for ntt.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "context.h"
#include "ntt_kernels4.hip.h"

namespace pm {

typedef void (*pass_fn)(const NttPassArgs, const NttConsts);
struct Pass4Entry {
  int S, LT, role;  // role: 0 single, 1 first, 2 middle, 3 last
  pass_fn fn;
};
// single-pass shapes (LT = 0) and the multi-pass tiles of 2^11 elements (2^12 for S = 10)
#define PM4_SINGLE(X) X(2, 0) X(3, 0) X(4, 0) X(5, 0) X(6, 0) X(7, 0) X(8, 0) X(9, 0) X(10, 0)
#define PM4_MULTI(X) X(5, 6) X(6, 5) X(7, 4) X(8, 3) X(9, 2) X(10, 1) X(10, 2) X(5, 5) X(6, 4) X(7, 3) X(8, 2) X(10, 0) X(9, 1)
static const Pass4Entry kPass4Table[] = {
#define X(S, LT) {S, LT, 0, ntt_pass4_kernel<S, LT, false, false, false>},
    PM4_SINGLE(X)
#undef X
#define X(S, LT)                                               \
  {S, LT, 1, ntt_pass4_kernel<S, LT, true, false, true>},      \
      {S, LT, 2, ntt_pass4_kernel<S, LT, false, true, true>},  \
      {S, LT, 3, ntt_pass4_kernel<S, LT, false, true, false>},
        PM4_MULTI(X)
#undef X
};

pass_fn find_pass4(int S, int LT, int role) {
  for (const Pass4Entry& e : kPass4Table)
    if (e.S == S && e.LT == LT && e.role == role) return e.fn;
  return nullptr;
}
size_t pass4_lds(int S, int LT) { return pass4_lds_bytes(S, LT); }
unsigned pass4_threads(int S, int LT) { return std::max(64u, (1u << (S + LT)) / 4); }

int build_step4_table(pm_ctx* ctx, void** out, const NttConsts& c, unsigned S, hipStream_t st) {
  const u32 total = (u32)step4_tw_total((int)S);
  PM_HIP(ctx, hipMalloc(out, (size_t)total * 48));
  hipLaunchKernelGGL(step4_tw_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (u32x4*)*out, c, S);
  PM_HIP(ctx, hipGetLastError());
  return PM_OK;
}

}  // namespace pm
