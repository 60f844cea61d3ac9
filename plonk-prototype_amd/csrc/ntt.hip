// Host side of the Fr NTT: domain tables, pass planning, launches, and the pm_fr_ntt* ABI.
// Mirrors dusk_plonk::fft::EvaluationDomain (dusk-plonk 0.8.2, ref:Cargo.toml:19):
//   new()      -> domain_constants()  (group_gen = ROOT_OF_UNITY^(2^(32-log_n)), size_inv, ...)
//   fft/ifft/coset_fft/coset_ifft -> ntt_run() with PM_NTT_INVERSE / PM_NTT_COSET
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <thread>

#include <algorithm>
#include <cstring>

#include "context.h"
#include "host_field.h"
#include "ntt_kernels.hip.h"

namespace pm {

using host::HFr;

// ------------------------------------------------------------------ kernel table
typedef void (*pass_fn)(const NttPassArgs, const NttConsts);
enum Role { ROLE_SINGLE = 0, ROLE_FIRST = 1, ROLE_MIDDLE = 2, ROLE_LAST = 3 };
pass_fn find_pass4(int S, int LT, int role);   // radix-4 kernels (ntt4.hip)
struct PassEntry {
  int S, LT, role;
  pass_fn fn;
};
// (S, LT): LT = 0 single-pass kernels; multi-pass tiles of 2^11 (LT = 11 - S) or 2^12 elements
#define PM_SINGLE(X) X(3, 0) X(4, 0) X(5, 0) X(6, 0) X(7, 0) X(8, 0) X(9, 0) X(10, 0)
#define PM_MULTI(X) \
  X(5, 6) X(6, 5) X(7, 4) X(8, 3) X(9, 2) X(10, 1) X(8, 4) X(9, 3) X(10, 2) X(5, 5) X(6, 4) X(7, 3) X(8, 2)
static const PassEntry kPassTable[] = {
#define X(S, LT) {S, LT, ROLE_SINGLE, ntt_pass_kernel<S, LT, false, false, false>},
    PM_SINGLE(X)
#undef X
#define X(S, LT)                                                      \
  {S, LT, ROLE_FIRST, ntt_pass_kernel<S, LT, true, false, true>},     \
      {S, LT, ROLE_MIDDLE, ntt_pass_kernel<S, LT, false, true, true>}, \
      {S, LT, ROLE_LAST, ntt_pass_kernel<S, LT, false, true, false>},
        PM_MULTI(X)
#undef X
};
static pass_fn find_pass(int S, int LT, int role) {
  for (const PassEntry& e : kPassTable)
    if (e.S == S && e.LT == LT && e.role == role) return e.fn;
  return nullptr;
}

// radix-4 kernels (ntt4.hip)
size_t pass4_lds(int S, int LT);
unsigned pass4_threads(int S, int LT);
int build_step4_table(pm_ctx* ctx, void** out, const NttConsts& c, unsigned S, hipStream_t st);

struct Plan {
  int npass = 0;
  int S[4] = {0, 0, 0, 0};
  int LT[4] = {0, 0, 0, 0};
};
static Plan make_plan(unsigned log_n, int tile_log, int max_radix = 10, int radix = 4, unsigned min_tiles = 256,
                      unsigned batch = 1) {
  Plan p;
  if (log_n < 3) return p;  // tiny kernel
  if (log_n <= 10) {  // one workgroup per transform; beyond 2^10 two passes over many CUs are faster
    p.npass = 1;     // (measured: a single 2^12 workgroup 52 us, two passes 35 us)
    p.S[0] = (int)log_n;
    return p;
  }
  p.npass = (int)((log_n + max_radix - 1) / max_radix);
  p.npass = std::min(p.npass, (int)log_n / 5);  // multi-pass kernels exist for radix >= 2^5
  p.npass = std::min(p.npass, 4);               // ... and a plan holds four passes (ntt_max_radix 6 or 7 above 2^24 / 2^28)
  int base = (int)log_n / p.npass, extra = (int)log_n % p.npass;
  for (int i = 0; i < p.npass; ++i) {
    p.S[i] = base + (i < extra ? 1 : 0);
    // tile_log 0 = auto (measured, profiles/r01_ntt_sweep.txt): 2^10-element tiles for S <= 8 (four
    // or more independent workgroups per CU, so barrier phases of one overlap arithmetic of
    // another), never fewer than 4 adjacent columns (64-byte runs per limb group)
    // r05: a radix-2^10 pass takes TWO columns per workgroup (512 threads, 73 KB of LDS: two workgroups per CU whose load / barrier /
    // store phases overlap) -- with the r05 product (95 - 106 VGPRs) that is 204 -> 195 us per forward + inverse 2^20 transform and
    // 206 -> 191 us in a batch of four (profiles/r05_ntt_tile_ab.txt; r02 measured no difference with the 128-VGPR kernels)
    int lt = tile_log ? tile_log - p.S[i] : (p.S[i] >= 10 && radix == 4 ? 1 : std::max(10 - p.S[i], 2));
    if (p.S[i] < 8 && lt > 11 - p.S[i]) lt = 11 - p.S[i];  // 2^12-element tiles exist for S >= 8 only
    if (!tile_log && radix == 4) {
      // fill the chip: narrower tiles while there are fewer tiles than CUs (measured, profiles/r02_ntt_tile_ab.txt:
      // 2^18: 61 -> 51 us, 2^19: 89 -> 72 us; no change from 2^20 up, where 4-column tiles already give 256)
      while (lt > (p.S[i] >= 10 ? 0 : 1) && ((((size_t)batch << log_n) >> (p.S[i] + lt)) < min_tiles) &&
             find_pass4(p.S[i], lt - 1, ROLE_LAST) != nullptr)
        --lt;
    }
    if (lt < 0) lt = 0;
    if (radix != 4 && lt < 1) lt = 1;                        // single-column tiles exist in the radix-4 family only
    if (radix != 4 && tile_log == 10 && p.S[i] > 8) lt = 11 - p.S[i];
    lt = std::min(lt, (int)log_n - p.S[i]);
    p.LT[i] = lt;
  }
  return p;
}

// ------------------------------------------------------------------ host constants
static HFr fr_pow2k(HFr x, unsigned k) {  // x^(2^k)
  for (unsigned i = 0; i < k; ++i) x = host::mul(x, x, host::FR());
  return x;
}
static HFr domain_gen(unsigned log_n) { return fr_pow2k(host::fr_root_of_unity(), 32 - log_n); }
// ABI Montgomery form (x * 2^256) -> device form (x * 2^261) as 9 x 29-bit limbs
static void to_limbs(u32* dst, HFr v) {
  for (int i = 0; i < 5; ++i) v = host::add(v, v, host::FR());
  for (int i = 0; i < 9; ++i) {
    const int lo = 29 * i, j = lo / 64, sh = lo % 64;
    u64 x = v.l[j] >> sh;
    if (sh + 29 > 64 && j + 1 < 4) x |= v.l[j + 1] << (64 - sh);
    dst[i] = (u32)(x & ((1u << 29) - 1));
  }
}

static int build_pow_table(pm_ctx* ctx, void** out, const HFr& base, const HFr& mult, u32 count,
                           u32 stride, hipStream_t st) {
  PM_HIP(ctx, hipMalloc(out, (size_t)count * 48));
  NttConsts c;
  memset(&c, 0, sizeof c);
  to_limbs(c.w8[0], base);
  to_limbs(c.scale, mult);
  to_limbs(c.one, host::one(host::FR()));
  hipLaunchKernelGGL(pow_table_kernel, dim3((count + 255) / 256), dim3(256), 0, st, (u32x4*)*out, c,
                     count, stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    (void)hipFree(*out);
    *out = nullptr;
    return set_err(ctx, PM_ERR_HIP, std::string("pow_table_kernel: ") + hipGetErrorString(e));
  }
  return PM_OK;
}

static int get_step_table(pm_ctx* ctx, int dir, unsigned S, void** out, hipStream_t st) {
  auto it = ctx->step_tw[dir].find(S);
  if (it != ctx->step_tw[dir].end()) {
    *out = it->second;
    return PM_OK;
  }
  HFr wR = domain_gen(S);
  if (dir) wR = host::inv(wR, host::FR());
  u32 total = (u32)step_tw_total((int)S);
  void* d = nullptr;
  PM_HIP(ctx, hipMalloc(&d, (size_t)total * 48));
  NttConsts c;
  memset(&c, 0, sizeof c);
  to_limbs(c.w8[0], wR);
  to_limbs(c.one, host::one(host::FR()));
  hipLaunchKernelGGL(step_tw_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (u32x4*)d, c, S);
  PM_HIP(ctx, hipGetLastError());
  ctx->step_tw[dir][S] = d;
  *out = d;
  return PM_OK;
}

static int get_step4_table(pm_ctx* ctx, int dir, unsigned S, void** out, hipStream_t st) {
  auto it = ctx->step4_tw[dir].find(S);
  if (it != ctx->step4_tw[dir].end()) {
    *out = it->second;
    return PM_OK;
  }
  HFr wR = domain_gen(S);
  if (dir) wR = host::inv(wR, host::FR());
  NttConsts c;
  memset(&c, 0, sizeof c);
  to_limbs(c.w8[0], wR);
  to_limbs(c.one, host::one(host::FR()));
  void* d = nullptr;
  int rc = build_step4_table(ctx, &d, c, S, st);
  if (rc) return rc;
  ctx->step4_tw[dir][S] = d;
  *out = d;
  return PM_OK;
}

static int get_domain_tables(pm_ctx* ctx, int dir, unsigned log_n, bool need_coset,
                             NttDomainTables** out, hipStream_t st) {
  NttDomainTables& t = ctx->domain[dir][log_n];
  const host::Field<4>& F = host::FR();
  t.lh = (log_n + 1) / 2;
  const u32 n_lo = 1u << t.lh, n_hi = 1u << (log_n - t.lh);
  // a pair of tables is published only when both were built: a failed second allocation must not
  // leave a half-initialised entry for the next call to launch with
  auto build_pair = [&](const HFr& base, void** lo_out, void** hi_out) -> int {
    void *lo = nullptr, *hi = nullptr;
    int rc = build_pow_table(ctx, &lo, base, host::one(F), n_lo, 1, st);
    if (!rc) rc = build_pow_table(ctx, &hi, base, host::one(F), n_hi, n_lo, st);
    if (rc) {
      if (lo) (void)hipFree(lo);
      if (hi) (void)hipFree(hi);
      return rc;
    }
    *lo_out = lo;
    *hi_out = hi;
    return PM_OK;
  };
  if (!t.tw_lo || !t.tw_hi) {
    HFr w = domain_gen(log_n);
    if (dir) w = host::inv(w, F);
    int rc = build_pair(w, &t.tw_lo, &t.tw_hi);
    if (rc) return rc;
  }
  if (need_coset && (!t.cs_lo || !t.cs_hi)) {
    HFr g = host::from_u64(host::FR_GENERATOR, F);
    if (dir) g = host::inv(g, F);
    int rc = build_pair(g, &t.cs_lo, &t.cs_hi);
    if (rc) return rc;
  }
  *out = &t;
  return PM_OK;
}

static void fill_consts(NttConsts& c, int dir, unsigned log_n) {
  const host::Field<4>& F = host::FR();
  HFr w8 = domain_gen(3);
  if (dir) w8 = host::inv(w8, F);
  HFr w = w8;
  for (int i = 0; i < 3; ++i) {
    to_limbs(c.w8[i], w);
    w = host::mul(w, w8, F);
  }
  to_limbs(c.one, host::one(F));
  if (dir)
    to_limbs(c.scale, host::inv(host::from_u64((u64)1 << log_n, F), F));
  else
    to_limbs(c.scale, host::one(F));
}

// Inter-pass twiddle table of pass i (N x 36 B, laid out like the wide data): built on first use for
// the plan in force, rebuilt when the tunables changed the plan.
static unsigned plan_wide_glog(const Plan& plan) {
  int min_lt = 3;
  for (int i = 0; i < plan.npass; ++i) min_lt = std::min(min_lt, plan.LT[i]);
  return (unsigned)std::max(min_lt, 2);
}
static int get_pass_tw(pm_ctx* ctx, NttDomainTables* dt, int i, bool last, int dir, unsigned log_n, unsigned log_ns,
                       int S, unsigned wide_glog, const NttConsts& kc, hipStream_t st, void** out) {
  const size_t n = (size_t)1 << log_n;
  const unsigned key = (((((log_ns << 8) | (unsigned)S) << 1) | (last ? 1u : 0u)) << 2) | wide_glog;
  if (dt->pass_tw[i] && dt->pass_tw_key[i] != key) {  // the plan changed (tunables): rebuild
    PM_HIP(ctx, hipDeviceSynchronize());
    PM_HIP(ctx, hipFree(dt->pass_tw[i]));
    dt->pass_tw[i] = nullptr;
  }
  if (!dt->pass_tw[i]) {
    void* t = nullptr;
    PM_HIP(ctx, hipMalloc(&t, n * 36));
    NttConsts c2 = kc;
    if (!(last && dir)) memcpy(c2.scale, kc.one, sizeof c2.scale);  // only the inverse's last pass carries n^-1
    hipLaunchKernelGGL(pass_tw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t, c2, log_n, log_ns,
                       (u32)S, (const u32x4*)dt->tw_hi, (const u32x4*)dt->tw_lo, dt->lh, wide_glog);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      (void)hipFree(t);
      return set_err(ctx, PM_ERR_HIP, std::string("pass_tw_kernel: ") + hipGetErrorString(e));
    }
    dt->pass_tw[i] = t;
    dt->pass_tw_key[i] = key;
  }
  *out = dt->pass_tw[i];
  return PM_OK;
}

// ------------------------------------------------------------------ execution
int ntt_run(pm_ctx* ctx, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
            size_t out_stride, unsigned log_n, unsigned batch, unsigned flags, hipStream_t st) {
  if (log_n >= host::FR_TWO_ADICITY)
    return set_err(ctx, PM_ERR_DOMAIN_TOO_LARGE, "log_n >= 32 (Fr two-adicity)");
  const size_t n = (size_t)1 << log_n;
  if (in_len > n) return set_err(ctx, PM_ERR_LENGTH, "in_len > 2^log_n");
  if (batch == 0) return PM_OK;
  if (batch > 1 && (in_stride < in_len || out_stride < n))
    return set_err(ctx, PM_ERR_BAD_ARG, "batch strides shorter than the vectors");
  if (flags & ~(PM_NTT_INVERSE | PM_NTT_COSET)) return set_err(ctx, PM_ERR_BAD_ARG, "unknown flags");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  OrderScope order_scope(ctx, ctx->ord_ntt, st);   // scratch vectors and lazily built tables are shared by all streams
  if (order_scope.rc) return order_scope.rc;
  const int dir = (flags & PM_NTT_INVERSE) ? 1 : 0;
  const bool coset = (flags & PM_NTT_COSET) != 0;

  if (log_n == 0) {  // size-1 domain: identity (w = 1, size_inv = 1, g^0 = 1); zero if in_len == 0
    for (unsigned b = 0; b < batch; ++b) {
      char* o = (char*)d_out + b * out_stride * 32;
      if (in_len == 0) {
        PM_HIP(ctx, hipMemsetAsync(o, 0, 32, st));
      } else if (o != (const char*)d_in + b * in_stride * 32) {
        PM_HIP(ctx, hipMemcpyAsync(o, (const char*)d_in + b * in_stride * 32, 32,
                                   hipMemcpyDeviceToDevice, st));
      }
    }
    return PM_OK;
  }

  NttDomainTables* dt = nullptr;
  int rc = get_domain_tables(ctx, dir, log_n, coset, &dt, st);
  if (rc) return rc;
  NttConsts kc;
  fill_consts(kc, dir, log_n);

  NttPassArgs a;
  memset(&a, 0, sizeof a);
  a.tw_hi = (const u32x4*)dt->tw_hi;
  a.tw_lo = (const u32x4*)dt->tw_lo;
  a.cs_hi = (const u32x4*)dt->cs_hi;
  a.cs_lo = (const u32x4*)dt->cs_lo;
  a.log_n = log_n;
  a.lh = dt->lh;

  const u32 pre = (coset && !dir) ? PASS_PRE_COSET : 0u;
  // n^-1 of an inverse transform: folded into the last pass's twiddle table when there are two
  // or more passes with direct tables; otherwise multiplied explicitly (PASS_POST_SCALE)
  const bool direct_tw = log_n <= 26 && ctx->opt_ntt_direct_tw;  // tables of N x 36 B per twiddled pass and direction

  Plan plan = make_plan(log_n, (int)ctx->opt_ntt_tile_log, (int)ctx->opt_ntt_max_radix, (int)ctx->opt_ntt_radix,
                        (unsigned)ctx->num_cus, batch);
  const bool scale_folded = dir && plan.npass > 1 && direct_tw;
  const u32 post = ((dir && !scale_folded) ? PASS_POST_SCALE : 0u) | ((dir && coset) ? PASS_POST_COSET : 0u);
  if (plan.npass == 0) {
    a.in = (const u32x4*)d_in;
    a.out = (u32x4*)d_out;
    a.batch_stride_in = in_stride;
    a.batch_stride_out = out_stride;
    a.in_len = (u32)in_len;
    a.flags = pre | post;
    hipLaunchKernelGGL(ntt_tiny_kernel, dim3(1, batch), dim3(64), 0, st, a, kc);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
  }

  // buffers: pass i reads src, writes (last ? out : tmp[i & 1]); a pass never runs in place
  const bool inplace = (d_in == d_out);
  const int ntmp = plan.npass - 1 + ((plan.npass == 1 && inplace) ? 1 : 0);
  for (int i = 0; i < std::min(ntmp, 2); ++i) {
    rc = ensure_buffer(ctx, ctx->ntt_tmp[i], (size_t)batch * n * 36);
    if (rc) return rc;
  }
  const void* src = d_in;
  unsigned log_ns = 0;
  a.wide_total = (unsigned long long)batch * n;
  a.wide_glog = plan_wide_glog(plan);
  for (int i = 0; i < plan.npass; ++i) {
    const bool last = (i == plan.npass - 1);
    const int S = plan.S[i], LT = plan.LT[i];
    // intermediate vectors are 9-limb "wide" planes in ntt_tmp[]; only a single-pass
    // in-place transform needs a canonical bounce buffer (ntt_tmp[0]) plus a copy
    void* dst = (last && !(plan.npass == 1 && inplace)) ? d_out : ctx->ntt_tmp[i & 1].ptr;
    const int role = plan.npass == 1 ? ROLE_SINGLE : (i == 0 ? ROLE_FIRST : (last ? ROLE_LAST : ROLE_MIDDLE));
    const bool r4 = ctx->opt_ntt_radix == 4 && find_pass4(S, LT, role) != nullptr;
    pass_fn fn = r4 ? find_pass4(S, LT, role) : find_pass(S, LT, role);
    if (!fn) return set_err(ctx, PM_ERR_BAD_ARG, "no kernel for this pass shape");
    void* stw = nullptr;
    rc = r4 ? get_step4_table(ctx, dir, (unsigned)S, &stw, st) : get_step_table(ctx, dir, (unsigned)S, &stw, st);
    if (rc) return rc;
    a.in = src;
    a.out = dst;
    a.step_tw = (const u32x4*)stw;
    a.batch_stride_in = in_stride;
    a.batch_stride_out = (plan.npass == 1 && inplace) ? n : out_stride;
    a.in_len = (u32)in_len;
    a.log_ns = log_ns;
    u32 post_i = last ? post : 0u;
    a.pass_tw = nullptr;
    u32 tw_flag = 0;
    if (i > 0 && direct_tw) {
      void* ptw = nullptr;
      rc = get_pass_tw(ctx, dt, i, last, dir, log_n, log_ns, S, a.wide_glog, kc, st, &ptw);
      if (rc) return rc;
      a.pass_tw = ptw;
      tw_flag = PASS_DIRECT_TW;
    }
    a.flags = (i == 0 ? pre : 0u) | post_i | tw_flag | (ctx->opt_ntt_xcd ? PASS_XCD_REMAP : 0u);
    const unsigned threads = r4 ? pass4_threads(S, LT) : std::max(64u, (1u << (S + LT)) / 8);
    const size_t lds = r4 ? pass4_lds(S, LT) : pass_lds_bytes(S, LT);
    if (lds > 64 * 1024)   // once per kernel and process, not per launch
      if (int lrc = raise_lds_limit(ctx, (const void*)fn, lds)) return lrc;
    const unsigned blocks = (unsigned)(n >> (S + LT));
    {
      static const char* kRoleName[4] = {"ntt_pass_single", "ntt_pass_first", "ntt_pass_middle", "ntt_pass_last"};
      ProfScope prof(ctx, st, kRoleName[role]);
      hipLaunchKernelGGL(fn, dim3(blocks, batch), dim3(threads), lds, st, a, kc);
    }
    PM_HIP(ctx, hipGetLastError());
    src = dst;
    log_ns += (unsigned)S;
  }
  if (plan.npass == 1 && inplace) {
    PM_HIP(ctx, hipMemcpy2DAsync(d_out, out_stride * 32, ctx->ntt_tmp[0].ptr, n * 32, n * 32, batch,
                                 hipMemcpyDeviceToDevice, st));
  }
  return PM_OK;
}

}  // namespace pm

// ------------------------------------------------------------------ C ABI
using namespace pm;

extern "C" int pm_ntt_plan(uint32_t log_n, uint32_t radix_log2[4], uint32_t* n_passes) {
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  Plan p = make_plan(log_n, 0);
  for (int i = 0; i < 4; ++i) radix_log2[i] = (uint32_t)p.S[i];
  *n_passes = (uint32_t)p.npass;
  return PM_OK;
}

extern "C" int pm_domain_info(uint32_t log_n, uint64_t group_gen[4], uint64_t group_gen_inv[4],
                              uint64_t size_inv[4]) {
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  const host::Field<4>& F = host::FR();
  HFr g = domain_gen(log_n);
  HFr gi = host::inv(g, F);
  HFr si = host::inv(host::from_u64((u64)1 << log_n, F), F);
  memcpy(group_gen, g.l, 32);
  memcpy(group_gen_inv, gi.l, 32);
  memcpy(size_inv, si.l, 32);
  return PM_OK;
}

// ---- the small helpers of dusk_plonk::fft::EvaluationDomain (dusk-plonk 0.8.2, ref:Cargo.toml:19):
// what a Rust patch at the quotient_poly::compute / linearisation level calls besides the transforms.
namespace {
HFr hfr_pow_u64(const HFr& a, host::u64 e) { return host::pow(a, &e, 1, host::FR()); }
// log of tau to the base w (a generator of the 2^log_n domain), or -1 when tau is not in the domain: Pohlig-Hellman
// over the 2-group, one bit per step
long long domain_log(const HFr& tau, unsigned log_n) {
  const host::Field<4>& F = host::FR();
  const HFr one = host::one(F), w = domain_gen(log_n), wi = host::inv(w, F);
  u64 idx = 0;
  HFr rest = tau;   // tau w^-idx: its order divides 2^(log_n - j) after step j
  HFr wpow = wi;    // w^-(2^j)
  for (unsigned j = 0; j < log_n; ++j) {
    HFr t = rest;
    for (unsigned s = 0; s + j + 1 < log_n; ++s) t = host::mul(t, t, F);   // rest^(2^(log_n - 1 - j)): 1 or -1
    if (!host::eq(t, one)) {
      idx |= (u64)1 << j;
      rest = host::mul(rest, wpow, F);
    }
    wpow = host::mul(wpow, wpow, F);
  }
  return host::eq(rest, one) ? (long long)idx : -1;
}
}  // namespace

// EvaluationDomain::evaluate_vanishing_polynomial(tau) = tau^size - 1
extern "C" int pm_domain_evaluate_vanishing_polynomial(uint32_t log_n, const uint64_t tau[4], uint64_t out[4]) {
  if (!tau || !out) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  const host::Field<4>& F = host::FR();
  HFr t;
  memcpy(t.l, tau, 32);
  const HFr r = host::sub(hfr_pow_u64(t, (u64)1 << log_n), host::one(F), F);
  memcpy(out, r.l, 32);
  return PM_OK;
}

// fft::domain::compute_vanishing_poly_over_coset(domain, poly_degree): Evaluations over the 2^log_n domain of
// X^poly_degree - 1 on the coset g H:  out[i] = g^d (w^d)^i - 1,  g = GENERATOR = 7.  Upstream asserts size > poly_degree.
extern "C" int pm_domain_vanishing_poly_over_coset(uint32_t log_n, uint64_t poly_degree, uint64_t* out) {
  if (!out) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  if (poly_degree >= ((u64)1 << log_n)) return PM_ERR_BAD_ARG;
  const host::Field<4>& F = host::FR();
  const HFr one = host::one(F), step = hfr_pow_u64(domain_gen(log_n), poly_degree);
  HFr cur = hfr_pow_u64(host::from_u64(host::FR_GENERATOR, F), poly_degree);
  for (size_t i = 0; i < ((size_t)1 << log_n); ++i) {
    const HFr v = host::sub(cur, one, F);
    memcpy(out + 4 * i, v.l, 32);
    cur = host::mul(cur, step, F);
  }
  return PM_OK;
}
extern "C" int pm_domain_vanishing_poly_over_coset_dev(pm_ctx* ctx, uint32_t log_n, uint64_t poly_degree, void* d_out,
                                                       void* hip_stream) {
  if (!ctx || !d_out) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  if (poly_degree >= ((u64)1 << log_n)) return PM_ERR_BAD_ARG;
  const host::Field<4>& F = host::FR();
  const HFr one = host::one(F), step = hfr_pow_u64(domain_gen(log_n), poly_degree);
  const HFr scale = hfr_pow_u64(host::from_u64(host::FR_GENERATOR, F), poly_degree);
  const size_t n = (size_t)1 << log_n;
  int rc = pm_fr_powers_dev(ctx, step.l, scale.l, n, d_out, hip_stream);
  // the scalar operand of the broadcast subtraction lives in the first element's place for one launch: a one-element
  // device vector of its own, freed after the stream has used it
  void* d_one = nullptr;
  if (!rc) rc = pm_dev_alloc(ctx, 32, &d_one);
  if (!rc) rc = pm_dev_upload(ctx, d_one, one.l, 32);
  if (!rc) rc = pm_fr_vec_op_dev(ctx, 1 /* sub */, d_out, d_one, 1, d_out, n, hip_stream);
  if (d_one) {
    if (hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : ctx->stream) != hipSuccess && !rc) rc = PM_ERR_HIP;
    (void)pm_dev_free(ctx, d_one);
  }
  return rc;
}

// EvaluationDomain::evaluate_all_lagrange_coefficients(tau): out[i] = L_i(tau) = (tau^n - 1) / n * w^i / (tau - w^i);
// for tau = w^j in the domain the indicator of j.
extern "C" int pm_domain_evaluate_all_lagrange_coefficients(uint32_t log_n, const uint64_t tau[4], uint64_t* out) {
  if (!tau || !out) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  const host::Field<4>& F = host::FR();
  const size_t n = (size_t)1 << log_n;
  HFr t;
  memcpy(t.l, tau, 32);
  const HFr one = host::one(F), w = domain_gen(log_n), tn = hfr_pow_u64(t, (u64)n);
  if (host::eq(tn, one)) {
    memset(out, 0, n * 32);
    const long long j = domain_log(t, log_n);
    if (j >= 0) memcpy(out + 4 * (size_t)j, one.l, 32);
    return PM_OK;
  }
  // util::batch_inversion (Montgomery's trick) on u[i] = tau - w^i; the running products go to `out`
  std::vector<HFr> u(n);
  HFr r = one, acc = one;
  for (size_t i = 0; i < n; ++i) {
    u[i] = host::sub(t, r, F);
    memcpy(out + 4 * i, acc.l, 32);          // product of u[0 .. i)
    acc = host::mul(acc, u[i], F);
    r = host::mul(r, w, F);
  }
  HFr inv_all = host::inv(acc, F);
  const HFr l0 = host::mul(host::sub(tn, one, F), host::inv(host::from_u64((u64)n, F), F), F);   // (tau^n - 1) / n
  const HFr wi = host::inv(w, F);
  HFr l = host::mul(l0, hfr_pow_u64(w, (u64)(n - 1)), F);   // l0 w^(n-1), walked downwards
  for (size_t i = n; i-- > 0;) {
    HFr prefix;
    memcpy(prefix.l, out + 4 * i, 32);
    const HFr ui_inv = host::mul(inv_all, prefix, F);
    inv_all = host::mul(inv_all, u[i], F);
    const HFr v = host::mul(l, ui_inv, F);
    memcpy(out + 4 * i, v.l, 32);
    l = host::mul(l, wi, F);
  }
  return PM_OK;
}
extern "C" int pm_domain_evaluate_all_lagrange_coefficients_dev(pm_ctx* ctx, uint32_t log_n, const uint64_t tau[4], void* d_out,
                                                                void* hip_stream) {
  if (!ctx || !tau || !d_out) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  const host::Field<4>& F = host::FR();
  const size_t n = (size_t)1 << log_n;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  HFr t;
  memcpy(t.l, tau, 32);
  const HFr one = host::one(F), w = domain_gen(log_n), tn = hfr_pow_u64(t, (u64)n);
  if (host::eq(tn, one)) {
    PM_HIP(ctx, hipSetDevice(ctx->device));
    PM_HIP(ctx, hipMemsetAsync(d_out, 0, n * 32, st));
    const long long j = domain_log(t, log_n);
    if (j >= 0) {
      PM_HIP(ctx, hipMemcpyAsync((char*)d_out + 32 * (size_t)j, one.l, 32, hipMemcpyHostToDevice, st));
      PM_HIP(ctx, hipStreamSynchronize(st));   // `one` is a stack value
    }
    return PM_OK;
  }
  // out = w^i;  out -= tau  (= -(tau - w^i));  out = 1 / out;  out *= w^i scaled by -(tau^n - 1) / n
  void *d_sc = nullptr, *d_roots = nullptr;
  int rc = pm_dev_alloc(ctx, 32, &d_sc);
  if (!rc) rc = pm_dev_alloc(ctx, n * 32, &d_roots);
  const HFr neg_l0 = host::mul(host::sub(one, tn, F), host::inv(host::from_u64((u64)n, F), F), F);
  if (!rc) rc = pm_fr_powers_dev(ctx, w.l, neg_l0.l, n, d_roots, hip_stream);   // -(tau^n - 1) / n * w^i
  if (!rc) rc = pm_fr_powers_dev(ctx, w.l, one.l, n, d_out, hip_stream);
  if (!rc) rc = pm_dev_upload(ctx, d_sc, t.l, 32);
  if (!rc) rc = pm_fr_vec_op_dev(ctx, 1 /* sub */, d_out, d_sc, 1, d_out, n, hip_stream);
  if (!rc) rc = pm_fr_batch_inverse_dev(ctx, d_out, n, hip_stream);
  if (!rc) rc = pm_fr_vec_op_dev(ctx, 2 /* mul */, d_out, d_roots, n, d_out, n, hip_stream);
  if (d_sc || d_roots) {
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = PM_ERR_HIP;
    if (d_sc) (void)pm_dev_free(ctx, d_sc);
    if (d_roots) (void)pm_dev_free(ctx, d_roots);
  }
  return rc;
}

// test hook (pure host): the pass plan ntt_run() follows for this size and these tunables (0 = the defaults of a fresh
// context).  out[20]: passes, then per pass (up to 4) {log2 radix S, log2 columns per tile LT, threads per workgroup,
// LDS bytes}, then a bit mask of the passes that have a kernel, then the blocked layout's log2 group size, 0.
extern "C" int pm_test_ntt_plan(uint32_t log_n, uint32_t batch, long tile_log, long max_radix, long radix, uint32_t num_cus,
                                uint32_t out[20]) {
  if (!out || !batch) return PM_ERR_BAD_ARG;
  if (log_n >= host::FR_TWO_ADICITY) return PM_ERR_DOMAIN_TOO_LARGE;
  const long rdx = radix ? radix : 4;
  Plan plan = make_plan(log_n, (int)tile_log, (int)(max_radix ? max_radix : 10), (int)rdx, num_cus ? num_cus : 256u, batch);
  memset(out, 0, 20 * sizeof(uint32_t));
  out[0] = (uint32_t)plan.npass;
  for (int i = 0; i < plan.npass; ++i) {
    const bool last = (i == plan.npass - 1);
    const int role = plan.npass == 1 ? ROLE_SINGLE : (i == 0 ? ROLE_FIRST : (last ? ROLE_LAST : ROLE_MIDDLE));
    const int S = plan.S[i], LT = plan.LT[i];
    const bool r4 = rdx == 4 && find_pass4(S, LT, role) != nullptr;
    out[1 + 4 * i] = (uint32_t)S;
    out[2 + 4 * i] = (uint32_t)LT;
    out[3 + 4 * i] = r4 ? pass4_threads(S, LT) : std::max(64u, (1u << (S + LT)) / 8);
    out[4 + 4 * i] = (uint32_t)(r4 ? pass4_lds(S, LT) : pass_lds_bytes(S, LT));
    if ((r4 ? find_pass4(S, LT, role) : find_pass(S, LT, role)) != nullptr) out[17] |= 1u << i;
  }
  out[18] = plan.npass ? plan_wide_glog(plan) : 0u;
  return PM_OK;
}

extern "C" int pm_domain_prepare(pm_ctx* ctx, uint32_t log_n) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (log_n >= host::FR_TWO_ADICITY)
    return set_err(ctx, PM_ERR_DOMAIN_TOO_LARGE, "log_n >= 32 (Fr two-adicity)");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  if (log_n == 0) return PM_OK;
  OrderScope order_scope(ctx, ctx->ord_ntt, ctx->stream);
  if (order_scope.rc) return order_scope.rc;
  // exactly the tables ntt_run() would build on the first transform of this size: domain and coset
  // tables, the step tables of the kernels the plan picks (radix-4 or radix-8 family), and the
  // inter-pass twiddle tables of both directions
  Plan plan = make_plan(log_n, (int)ctx->opt_ntt_tile_log, (int)ctx->opt_ntt_max_radix, (int)ctx->opt_ntt_radix,
                        (unsigned)ctx->num_cus, 1);
  const unsigned wide_glog = plan_wide_glog(plan);
  for (int dir = 0; dir < 2; ++dir) {
    NttDomainTables* dt;
    int rc = get_domain_tables(ctx, dir, log_n, true, &dt, ctx->stream);
    if (rc) return rc;
    NttConsts kc;
    fill_consts(kc, dir, log_n);
    unsigned log_ns = 0;
    for (int i = 0; i < plan.npass; ++i) {
      const bool last = (i == plan.npass - 1);
      const int role = plan.npass == 1 ? ROLE_SINGLE : (i == 0 ? ROLE_FIRST : (last ? ROLE_LAST : ROLE_MIDDLE));
      const bool r4 = ctx->opt_ntt_radix == 4 && find_pass4(plan.S[i], plan.LT[i], role) != nullptr;
      void* stw;
      rc = r4 ? get_step4_table(ctx, dir, (unsigned)plan.S[i], &stw, ctx->stream)
              : get_step_table(ctx, dir, (unsigned)plan.S[i], &stw, ctx->stream);
      if (rc) return rc;
      if (i > 0 && log_n <= 26 && ctx->opt_ntt_direct_tw) {
        void* ptw;
        rc = get_pass_tw(ctx, dt, i, last, dir, log_n, log_ns, plan.S[i], wide_glog, kc, ctx->stream, &ptw);
        if (rc) return rc;
      }
      log_ns += (unsigned)plan.S[i];
    }
  }
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PM_OK;
}

extern "C" int pm_fr_ntt_dev(pm_ctx* ctx, const void* d_in, size_t in_len, size_t in_stride,
                             void* d_out, size_t out_stride, uint32_t log_n, uint32_t batch,
                             uint32_t flags, void* hip_stream) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!d_out || (!d_in && in_len)) return set_err(ctx, PM_ERR_BAD_ARG, "null device pointer");
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  return ntt_run(ctx, d_in, in_len, in_stride, d_out, out_stride, log_n, batch, flags, st);
}

extern "C" int pm_fr_ntt_batch(pm_ctx* ctx, const uint64_t* in, size_t in_len, size_t in_stride,
                               uint64_t* out, size_t out_stride, uint32_t log_n, uint32_t batch,
                               uint32_t flags) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!out || (!in && in_len)) return set_err(ctx, PM_ERR_BAD_ARG, "null pointer");
  if (log_n >= host::FR_TWO_ADICITY)
    return set_err(ctx, PM_ERR_DOMAIN_TOO_LARGE, "log_n >= 32 (Fr two-adicity)");
  const size_t n = (size_t)1 << log_n;
  if (in_len > n) return set_err(ctx, PM_ERR_LENGTH, "in_len > 2^log_n");
  if (batch == 0) return PM_OK;
  if (batch > 1 && (in_stride < in_len || out_stride < n))
    return set_err(ctx, PM_ERR_BAD_ARG, "batch strides shorter than the vectors");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t in_elems = std::max<size_t>(in_len, 1);
  int rc = ensure_buffer(ctx, ctx->io_in, (size_t)batch * in_elems * 32);
  if (rc) return rc;
  rc = ensure_buffer(ctx, ctx->io_out, (size_t)batch * n * 32);
  if (rc) return rc;
  if (batch == 1 || ctx->opt_ntt_pipeline == 0) {
    if (in_len)
      PM_HIP(ctx, hipMemcpy2DAsync(ctx->io_in.ptr, in_elems * 32, in, in_stride * 32, in_len * 32,
                                   batch, hipMemcpyHostToDevice, st));
    rc = ntt_run(ctx, ctx->io_in.ptr, in_len, in_elems, ctx->io_out.ptr, n, log_n, batch, flags, st);
    if (rc) return rc;
    PM_HIP(ctx, hipMemcpy2DAsync(out, out_stride * 32, ctx->io_out.ptr, n * 32, n * 32, batch,
                                 hipMemcpyDeviceToHost, st));
    PM_HIP(ctx, hipStreamSynchronize(st));
    return PM_OK;
  }
  // Several polynomials from host memory (a prover round): the transfers are 10-30x the transform, so
  // they are what gets overlapped.  PCIe is full duplex: vector b+1 goes up on `copy_in` while vector b
  // is transformed on the compute stream and vector b-1 comes down on `copy_out`.  Copies from pageable
  // memory block the calling thread, hence the downloads run on a helper thread.
  if (!ctx->copy_in) PM_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
  if (!ctx->copy_out) PM_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev_in(batch), ev_done(batch);
  for (uint32_t b = 0; b < batch; ++b) {
    PM_HIP(ctx, hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming));
    PM_HIP(ctx, hipEventCreateWithFlags(&ev_done[b], hipEventDisableTiming));
  }
  std::mutex mq;
  std::condition_variable cv;
  uint32_t launched = 0;          // vectors whose transform has been enqueued (guarded by mq)
  bool abort_all = false;
  hipError_t down_err = hipSuccess;
  std::thread downloader([&] {
    (void)hipSetDevice(ctx->device);
    for (uint32_t b = 0; b < batch; ++b) {
      {
        std::unique_lock<std::mutex> lk2(mq);
        cv.wait(lk2, [&] { return launched > b || abort_all; });
        if (abort_all) return;
      }
      hipError_t e = hipStreamWaitEvent(ctx->copy_out, ev_done[b], 0);
      if (e == hipSuccess)
        e = hipMemcpyAsync(out + 4 * (size_t)b * out_stride, (const char*)ctx->io_out.ptr + (size_t)b * n * 32, n * 32,
                           hipMemcpyDeviceToHost, ctx->copy_out);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_out);
      if (e != hipSuccess) {
        down_err = e;
        return;
      }
    }
  });
  hipError_t up_err = hipSuccess;
  for (uint32_t b = 0; b < batch && !rc && up_err == hipSuccess; ++b) {
    char* d_in_b = (char*)ctx->io_in.ptr + (size_t)b * in_elems * 32;
    if (in_len) up_err = hipMemcpyAsync(d_in_b, in + 4 * (size_t)b * in_stride, in_len * 32, hipMemcpyHostToDevice, ctx->copy_in);
    if (up_err == hipSuccess) up_err = hipEventRecord(ev_in[b], ctx->copy_in);
    if (up_err == hipSuccess) up_err = hipStreamWaitEvent(st, ev_in[b], 0);
    if (up_err != hipSuccess) break;
    rc = ntt_run(ctx, d_in_b, in_len, in_elems, (char*)ctx->io_out.ptr + (size_t)b * n * 32, n, log_n, 1, flags, st);
    if (rc) break;
    up_err = hipEventRecord(ev_done[b], st);
    {
      std::lock_guard<std::mutex> lk2(mq);
      launched = b + 1;
    }
    cv.notify_one();
  }
  if (rc || up_err != hipSuccess) {
    std::lock_guard<std::mutex> lk2(mq);
    abort_all = true;
  }
  cv.notify_one();
  downloader.join();
  (void)hipStreamSynchronize(st);
  for (uint32_t b = 0; b < batch; ++b) {
    (void)hipEventDestroy(ev_in[b]);
    (void)hipEventDestroy(ev_done[b]);
  }
  if (rc) return rc;
  if (up_err != hipSuccess || down_err != hipSuccess)
    return set_err(ctx, PM_ERR_HIP, std::string("pipelined batch NTT: ") + hipGetErrorString(up_err != hipSuccess ? up_err : down_err));
  return PM_OK;
}

extern "C" int pm_fr_ntt(pm_ctx* ctx, const uint64_t* in, size_t in_len, uint64_t* out,
                         uint32_t log_n, uint32_t flags) {
  return pm_fr_ntt_batch(ctx, in, in_len, in_len, out, (size_t)1 << (log_n < 32 ? log_n : 0), log_n,
                         1, flags);
}

// ------------------------------------------------------------------ N5: one transform split over ranks
// SURVEY.md section 8e row 3 / 8f N5.  The vector of a 2^log_n-point transform is block-distributed in natural
// order over `world` ranks (one process and one pm_ctx per GPU): rank r holds x[r N/W, (r+1) N/W) and receives the
// same block of the result.  N = N1 N2 as an N1 x N2 row-major matrix (N1 = 2^(log_n / 2)):
//     transpose (all-to-all) -> N2/W batched transforms of size N1 -> twiddles w_N^(j k1)
//     -> transpose -> N1/W batched transforms of size N2 -> transpose (natural order)
// A transpose is: pack this rank's [R_loc][C] slice into per-peer blocks [peer][C/W][R_loc] (a 32 x 32 tile
// transpose through LDS; the twiddle / coset factors are folded into this pass), one all-to-all of N/W^2-element
// blocks (the context's RCCL communicator: grouped ncclSend / ncclRecv, one direct xGMI transfer per peer -- or a
// caller-supplied callback), and an unpack that interleaves what the peers sent into [C/W][R].  The sub-transforms
// are the library's own batched passes.  Results are identical to the single-GPU plan bit for bit.
namespace pm {

struct FsMul {            // factor applied to element (row a, column b) of the slice while it is packed / unpacked
  u32 mode;               // 0 none, 1 table^((row0 + a) * b)  (twiddle), 2 table^((row0 + a) * C + b)  (coset by index)
  const u32x4* hi;
  const u32x4* lo;
  u32 lh;
  u32 row0;
};
PM_DEV void fs_apply(u32x4& v0, u32x4& v1, const FsMul& m, u32 a, u32 b, u32 C) {
  if (m.mode == 0) return;
  const u32 e = m.mode == 1 ? (m.row0 + a) * b : (m.row0 + a) * C + b;
  u32x4 raw[2] = {v0, v1};
  Fr x = fe_load<FrP>(raw);
  x = fe_mul<FrP>(x, two_level(m.hi, m.lo, e, m.lh));
  fe_store<FrP>(raw, x);
  v0 = raw[0];
  v1 = raw[1];
}
// A transpose step works on `batch` vectors at once: the per-peer block of the all-to-all holds the batch's blocks one
// after the other, so ONE exchange carries all of them (message size x batch, call count / batch).
// send[((s * batch + v) * Cs + bl) * R_loc + a] = f(src_v[a * C + s * Cw + bl]),  s = b / Cw, bl = b % Cw;  Cs = Cw, or Cw + 1
// with `halo`: the block for rank s then ends with one more column, (s + 1) Cw mod C -- the first column of the NEXT
// rank's range (the last rank gets column 0) -- which the unpack turns into one extra row.
__global__ void __launch_bounds__(256) fs_pack_kernel(const u32x4* src, size_t src_stride, u32 batch, u32 R_loc, u32 C, u32 Cw,
                                                      u32 halo, u32x4* send, const FsMul m) {
  __shared__ u32x4 t0[32][33], t1[32][33];
  const u32 v = blockIdx.z;
  src += 2 * (size_t)v * src_stride;
  const u32 a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  const u32 tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;   // 32 x 8
  for (u32 r = ty; r < 32; r += 8) {
    const u32 a = a0 + r, b = b0 + tx;
    if (a < R_loc && b < C) {
      u32x4 v0 = src[2 * ((size_t)a * C + b)], v1 = src[2 * ((size_t)a * C + b) + 1];
      fs_apply(v0, v1, m, a, b, C);
      t0[r][tx] = v0;
      t1[r][tx] = v1;
    }
  }
  __syncthreads();
  const u32 Cs = Cw + halo, W = C / Cw;
  for (u32 r = ty; r < 32; r += 8) {
    const u32 b = b0 + r, a = a0 + tx;
    if (a < R_loc && b < C) {
      const u32 s = b / Cw, bl = b % Cw;
      const size_t o = (((size_t)s * batch + v) * Cs + bl) * R_loc + a;
      send[2 * o] = t0[tx][r];
      send[2 * o + 1] = t1[tx][r];
      if (halo && bl == 0) {   // also the extra column of the rank before s
        const u32 sp = (s + W - 1) % W;
        const size_t oh = (((size_t)sp * batch + v) * Cs + Cw) * R_loc + a;
        send[2 * oh] = t0[tx][r];
        send[2 * oh + 1] = t1[tx][r];
      }
    }
  }
}
// dst_v[bl * R + p * R_loc + a] = g(recv[((p * batch + v) * Cs + bl) * R_loc + a]),  R = W R_loc; g's index: row bl, column
// p R_loc + a.  With halo the extra row bl == Cw goes to halo_v[p * R_loc + a] instead.
__global__ void __launch_bounds__(256) fs_unpack_kernel(const u32x4* recv, u32 batch, u32 R_loc, u32 W, u32 Cw, u32 halo,
                                                        u32x4* dst, size_t dst_stride, u32x4* halo_dst, const FsMul m) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 v = blockIdx.y;
  const u32 R = R_loc * W, Cs = Cw + halo;
  if (t >= (size_t)Cs * R) return;
  const u32 bl = (u32)(t / R), rr = (u32)(t % R), p = rr / R_loc, a = rr % R_loc;
  const size_t i = (((size_t)p * batch + v) * Cs + bl) * R_loc + a;
  u32x4 v0 = recv[2 * i], v1 = recv[2 * i + 1];
  fs_apply(v0, v1, m, bl, rr, R);
  u32x4* o = bl < Cw ? dst + 2 * ((size_t)v * dst_stride + t) : halo_dst + 2 * ((size_t)v * R + rr);
  o[0] = v0;
  o[1] = v1;
}

}  // namespace pm

extern "C" int pm_fr_ntt_fourstep_batch_dev(pm_ctx* ctx, void* d_inout, uint32_t batch, void* d_halo, void* d_stage,
                                            uint32_t log_n, uint32_t world, uint32_t rank, uint32_t flags,
                                            pm_alltoall_fn exchange, void* user) {
  if (!ctx || !d_inout || !d_stage || batch == 0) return PM_ERR_BAD_ARG;
  if (world == 0 || (world & (world - 1)) || rank >= world) return PM_ERR_BAD_ARG;
  if (flags & ~(PM_NTT_INVERSE | PM_NTT_COSET | PM_NTT_TRANSPOSED)) return PM_ERR_BAD_ARG;
  if (log_n < 2 || log_n > 26) return PM_ERR_DOMAIN_TOO_LARGE;   // 32-bit exponents of the two-level tables
  const uint32_t l1 = log_n / 2, l2 = log_n - l1;
  const uint32_t n1 = 1u << l1, n2 = 1u << l2;
  if (n1 % world || n2 % world) return PM_ERR_BAD_ARG;            // the ranks must divide both factors
  const bool inverse = flags & PM_NTT_INVERSE, coset = flags & PM_NTT_COSET;
  const bool want_halo = d_halo != nullptr;
  if (want_halo && (inverse || !(flags & PM_NTT_TRANSPOSED))) return PM_ERR_BAD_ARG;   // a row of the block-transposed result
  const size_t blk = ((size_t)1 << log_n) / world;                // elements per rank and vector
  if (batch > 65535u) return PM_ERR_BAD_ARG;
  u32x4* x = (u32x4*)d_inout;
  u32x4* send = (u32x4*)d_stage;
  // stage: [send | recv], each batch x (blk, or blk + n2 with a halo row) elements; one rank needs no recv half
  const size_t stage_half = (size_t)batch * (blk + (want_halo ? n2 : 0));
  u32x4* recv = world == 1 ? send : send + 2 * stage_half;
  hipStream_t st;
  NttDomainTables* dt = nullptr;
  // the transform holds the NTT resource group from here to its last launch, with the context unlocked in between (the
  // exchange callback): the scope's closing event is recorded, under the lock, when the function returns
  struct HeldOrder {
    pm_ctx* ctx;
    OrderScope* s;
    ~HeldOrder() {
      if (!s) return;
      std::lock_guard<std::mutex> lk(ctx->mu);
      delete s;
      --ctx->calls_holding_tables;
    }
  } held{ctx, nullptr};
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    PM_HIP(ctx, hipSetDevice(ctx->device));
    st = ctx->stream;
    if (world > 1 && !exchange && (!ctx->comm || ctx->comm_world != (int)world || ctx->comm_rank != (int)rank))
      return set_err(ctx, PM_ERR_EXCHANGE, "no exchange callback and no matching communicator (pm_comm_init)");
    held.s = new OrderScope(ctx, ctx->ord_ntt, st);
    ++ctx->calls_holding_tables;   // dt and the tables behind it stay in use across the unlocked exchange steps: no pm_trim
    int rc = held.s->rc;
    if (!rc) rc = get_domain_tables(ctx, inverse ? 1 : 0, log_n, coset, &dt, st);
    if (rc) return rc;
  }
  const FsMul none{0, nullptr, nullptr, 0, 0};
  // one transpose of every vector of the batch: slice [R_loc][C] of a row-distributed [R][C] matrix -> slice [C / W][R]
  // of its transpose (+ one halo row).  The result is in *where: x -- or, with one rank and no factor on the way in, the
  // stage buffer itself (the unpack would be a plain copy: the next sub-transform reads it from there; `keep_in_x`
  // forces the copy)
  auto transpose = [&](const u32x4* from, u32 R_loc, u32 C, const FsMul& on_pack, const FsMul& on_unpack, bool keep_in_x,
                       bool halo, const u32x4** where) -> int {
    const u32 Cw = C / world, Cs = Cw + (halo ? 1u : 0u);
    const size_t peer_bytes = (size_t)batch * Cs * R_loc * 32;
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      ProfScope prof(ctx, st, "ntt_fourstep_transpose");
      hipLaunchKernelGGL(fs_pack_kernel, dim3((C + 31) / 32, (R_loc + 31) / 32, batch), dim3(256), 0, st, from, blk, batch, R_loc,
                         C, Cw, halo ? 1u : 0u, send, on_pack);
      PM_HIP(ctx, hipGetLastError());
      ++ctx->stat_transpose_steps;
      if (world > 1) {
        ++ctx->stat_alltoall_calls;
        ctx->stat_alltoall_bytes += peer_bytes * (world - 1);
      }
      if (world > 1 && !exchange) {
        int rc = comm_alltoall(ctx, send, recv, peer_bytes, st);
        if (rc) return rc;
      }
    }
    if (world == 1 && on_unpack.mode == 0 && !keep_in_x && !halo) {
      *where = send;
      return PM_OK;
    }
    if (world > 1 && exchange) {   // the callback sees finished data and returns when the peers' blocks are in place
      {
        std::lock_guard<std::mutex> lk(ctx->mu);
        PM_HIP(ctx, hipStreamSynchronize(st));
      }
      if (exchange(user, send, recv, peer_bytes) != 0) return set_err(ctx, PM_ERR_EXCHANGE, "the all-to-all callback failed");
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    ProfScope prof(ctx, st, "ntt_fourstep_transpose");
    const size_t total = (size_t)Cs * R_loc * world;
    hipLaunchKernelGGL(fs_unpack_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, st, (const u32x4*)recv, batch,
                       R_loc, world, Cw, halo ? 1u : 0u, x, blk, (u32x4*)d_halo, on_unpack);
    PM_HIP(ctx, hipGetLastError());
    *where = x;
    return PM_OK;
  };
  const uint32_t sub = inverse ? PM_NTT_INVERSE : 0;
  // PM_NTT_TRANSPOSED: a forward transform leaves its result as the [N1 / W][N2] matrix of step 4 -- X[k2 N1 + k1] at
  // position k1 N2 + k2, rank r holding rows k1 in [r N1 / W, (r + 1) N1 / W) -- and skips the third all-to-all; an
  // inverse transform TAKES that layout: with the two factors swapped it is exactly the state after step 1.  What sits
  // between the two in a prover (the pointwise quotient) does not care about the order.
  const bool transposed = flags & PM_NTT_TRANSPOSED;
  const uint32_t la = transposed && inverse ? l2 : l1, lb = log_n - la;   // rows x columns of the input matrix
  const uint32_t na = 1u << la, nb = 1u << lb;
  const u32x4* cur = x;
  int rc = PM_OK;
  if (!(transposed && inverse)) {
    // 1: [A/W][B] -> [B/W][A]  (coset_fft: x[n] *= g^n on the way out)
    FsMul pre = none;
    if (coset && !inverse) pre = FsMul{2, (const u32x4*)dt->cs_hi, (const u32x4*)dt->cs_lo, dt->lh, rank * (na / world)};
    rc = transpose(x, na / world, nb, pre, none, false, false, &cur);
    if (rc) return rc;
  }
  // `count` contiguous transforms of 2^lg points, in -> out (the batched passes put the batch on gridDim.y: at most 65535 per launch)
  auto sub_ntts = [&](const void* in, void* out, uint32_t lg, size_t count) -> int {
    const size_t len = (size_t)1 << lg;
    for (size_t done = 0; done < count;) {
      const uint32_t now = (uint32_t)std::min<size_t>(count - done, 32768);
      int r = pm_fr_ntt_dev(ctx, (const char*)in + done * len * 32, len, len, (char*)out + done * len * 32, len, lg, now, sub, nullptr);
      if (r) return r;
      done += now;
    }
    return PM_OK;
  };
  // 2: B/W transforms of size A over the columns, every vector of the batch (the blocks are contiguous)
  rc = sub_ntts(cur, x, la, (size_t)batch * (nb / world));
  if (rc) return rc;
  // 3: [B/W][A] -> [A/W][B], element (j, k1) times w^(j k1) on the way out; with a halo one more row: the next rank's first
  const FsMul tw{1, (const u32x4*)dt->tw_hi, (const u32x4*)dt->tw_lo, dt->lh, rank * (nb / world)};
  rc = transpose(x, nb / world, na, tw, none, false, want_halo, &cur);
  if (rc) return rc;
  // 4: A/W transforms of size B over the rows (and over the halo rows: one per vector, contiguous in d_halo)
  rc = sub_ntts(cur, x, lb, (size_t)batch * (na / world));
  if (rc) return rc;
  if (want_halo) {
    rc = sub_ntts(d_halo, d_halo, lb, batch);
    if (rc) return rc;
  }
  if (transposed && !inverse) return PM_OK;
  // 5: [A/W][B] (k1, k2) -> [B/W][A] (k2, k1) = natural order  (coset_ifft: X[k] *= g^-k on the way in)
  FsMul post = none;
  if (coset && inverse) post = FsMul{2, (const u32x4*)dt->cs_hi, (const u32x4*)dt->cs_lo, dt->lh, rank * (nb / world)};
  return transpose(x, na / world, nb, none, post, true, false, &cur);
}

extern "C" int pm_fr_ntt_fourstep_dev(pm_ctx* ctx, void* d_inout, void* d_stage, uint32_t log_n, uint32_t world,
                                      uint32_t rank, uint32_t flags, pm_alltoall_fn exchange, void* user) {
  return pm_fr_ntt_fourstep_batch_dev(ctx, d_inout, 1, nullptr, d_stage, log_n, world, rank, flags, exchange, user);
}

extern "C" int pm_comm_stats(pm_ctx* ctx, uint64_t out[4], int reset) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (out) {
    out[0] = ctx->stat_alltoall_calls;
    out[1] = ctx->stat_alltoall_bytes;
    out[2] = ctx->stat_allgather_calls;
    out[3] = ctx->stat_transpose_steps;
  }
  if (reset) ctx->stat_alltoall_calls = ctx->stat_alltoall_bytes = ctx->stat_allgather_calls = ctx->stat_transpose_steps = 0;
  return PM_OK;
}
