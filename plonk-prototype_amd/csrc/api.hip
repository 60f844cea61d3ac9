// Context management, error reporting and the elementwise field-op test hooks.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "context.h"
#include "field_inv.hip.h"
#include "fields.hip.h"
#include "host_field.h"

namespace pm {

int set_err(pm_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int raise_lds_limit(pm_ctx* ctx, const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> limit;   // (device, kernel) -> bytes already granted
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = limit[{ctx->device, fn}];
  if (bytes <= have) return PM_OK;
  PM_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return PM_OK;
}

int ensure_buffer(pm_ctx* ctx, DeviceBuffer& b, size_t bytes) {
  if (b.bytes >= bytes && b.ptr) return PM_OK;
  if (b.ptr) {
    PM_HIP(ctx, hipDeviceSynchronize());   // the old buffer may still be in use on any caller stream
    PM_HIP(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
  }
  PM_HIP(ctx, hipMalloc(&b.ptr, bytes));
  b.bytes = bytes;
  return PM_OK;
}

static int order_enter(pm_ctx* ctx, StreamOrder& o, hipStream_t st) {
  if (!o.ev) PM_HIP(ctx, hipEventCreateWithFlags(&o.ev, hipEventDisableTiming));
  if (o.used && o.last != st) {
    if (o.pending) {   // the predecessor ran on the context's own stream: record now (that stream is ours, always valid)
      PM_HIP(ctx, hipEventRecord(o.ev, ctx->stream));
      o.pending = false;
    }
    PM_HIP(ctx, hipStreamWaitEvent(st, o.ev, 0));
  }
  return PM_OK;
}
OrderScope::OrderScope(pm_ctx* c, StreamOrder& ord, hipStream_t s) : ctx(c), o(ord), st(s), rc(order_enter(c, ord, s)) {}
OrderScope::~OrderScope() {
  // whatever this call enqueued (also on an error path) is what a successor on another stream has to wait for.  A
  // caller's stream is never touched again after the call returns: its event is recorded here.
  if (st == ctx->stream) {
    o.pending = true;
  } else if (o.ev && hipEventRecord(o.ev, st) == hipSuccess) {
    o.pending = false;
  } else {
    (void)hipGetLastError();
    return;
  }
  o.last = st;
  o.used = true;
}

static hipEvent_t prof_event(pm_ctx* ctx) {
  if (!ctx->prof_pool.empty()) {
    hipEvent_t e = ctx->prof_pool.back();
    ctx->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);   // timing only: no system-scope fence per record
  return e;
}
ProfScope::ProfScope(pm_ctx* c, hipStream_t s, const char* name) : ctx(c), st(s) {
  if (!ctx->profile) return;
  if (!ctx->profile_only.empty() && ctx->profile_only != name) return;
  hipEvent_t a = prof_event(ctx);
  stop = prof_event(ctx);
  (void)hipEventRecord(a, st);
  ctx->prof_pending.push_back({a, stop, name});
}
ProfScope::~ProfScope() {
  if (stop) (void)hipEventRecord(stop, st);
}
int prof_collect(pm_ctx* ctx) {
  for (auto& p : ctx->prof_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      auto& st = ctx->prof_stats[p.name];
      st.total_ms += ms;
      st.count += 1;
    }
    ctx->prof_pool.push_back(p.start);
    ctx->prof_pool.push_back(p.stop);
  }
  ctx->prof_pending.clear();
  return PM_OK;
}

template <class P, int OP>
__global__ void field_op_kernel(const u32x4* a, const u32x4* b, u32x4* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int V = P::N / 4;
  Fe<P> x = fe_load<P>(a + V * i), y = fe_load<P>(b + V * i), r;
  // operands are canonical ABI-Montgomery values (R = 2^(32 NS)); the device product divides by
  // R' = 2^(W N), so rescale by 2^(W N - 32 NS) to return the ABI-form product
  if (OP == 0) r = fe_abi_to_dev<P>(fe_mul<P>(x, y));
  if (OP == 1) r = fe_add<P>(x, y);
  if (OP == 2) r = fe_mul<P>(fe_sub<P, 2, 1>(x, y), fe_one<P>());
  // 1 / x in the ABI form: the integer inverse of x R is x^-1 / R; times R^2 R' (and the product's 1 / R') = x^-1 R
  if (OP == 3) r = fe_mul<P>(fe_inv_int<P>(x), fe_pow2<P, 64 * P::NS + P::W * P::N>());
  fe_store<P>(out + V * i, r);
}

}  // namespace pm

using namespace pm;

extern "C" const char* pm_version(void) { return "plonk_mi355x 0.1 (gfx950)"; }

extern "C" int pm_init(int device_id, pm_ctx** out) {
  if (!out) return PM_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count)
    return PM_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return PM_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PM_ERR_NO_DEVICE;  // gfx950 code objects only
  if (hipSetDevice(device_id) != hipSuccess) return PM_ERR_HIP;
  pm_ctx* ctx = new pm_ctx();
  ctx->device = device_id;
  ctx->marks_on = getenv("PM_HOST_MARKS") != nullptr;
  ctx->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return PM_ERR_HIP;
  }
  *out = ctx;
  return PM_OK;
}

// Twiddle caches and workspaces: everything a context rebuilds or regrows on demand.  The device must be idle.
static void release_caches(pm_ctx* ctx, bool all) {
  for (int d = 0; d < 2; ++d) {
    for (auto& kv : ctx->step_tw[d]) (void)hipFree(kv.second);
    for (auto& kv : ctx->step4_tw[d]) (void)hipFree(kv.second);
    for (auto& kv : ctx->domain[d]) {
      (void)hipFree(kv.second.tw_hi);
      (void)hipFree(kv.second.tw_lo);
      (void)hipFree(kv.second.cs_hi);
      (void)hipFree(kv.second.cs_lo);
      for (void* t : kv.second.pass_tw) (void)hipFree(t);
    }
    ctx->step_tw[d].clear();
    ctx->step4_tw[d].clear();
    ctx->domain[d].clear();
  }
  for (DeviceBuffer* b : {&ctx->ntt_tmp[0], &ctx->ntt_tmp[1], &ctx->io_in, &ctx->io_out, &ctx->msm_ws, &ctx->msm_scalars,
                          &ctx->poly_ws, &ctx->poly_tab}) {
    if (b->ptr) (void)hipFree(b->ptr);
    b->ptr = nullptr;
    b->bytes = 0;
  }
  if (all && ctx->msm_ctl.ptr) (void)hipFree(ctx->msm_ctl.ptr);   // pm_trim keeps it: small, and zero-when-idle is its invariant
}

extern "C" int pm_trim(pm_ctx* ctx, size_t* freed_bytes) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->calls_holding_tables > 0)   // a four-step transform of another thread is between two of its exchange steps
    return set_err(ctx, PM_ERR_BUSY, "pm_trim while a pm_fr_ntt_fourstep_dev call is in flight on this context");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  PM_HIP(ctx, hipDeviceSynchronize());   // the buffers may be in use on any caller stream
  size_t free0 = 0, free1 = 0, total = 0;
  PM_HIP(ctx, hipMemGetInfo(&free0, &total));
  release_caches(ctx, false);
  PM_HIP(ctx, hipMemGetInfo(&free1, &total));
  if (freed_bytes) *freed_bytes = free1 > free0 ? free1 - free0 : 0;
  return PM_OK;
}

extern "C" void pm_shutdown(pm_ctx* ctx) {
  if (!ctx) return;
  (void)pm_comm_destroy(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  release_caches(ctx, true);
  if (ctx->msm_host_pinned) (void)hipHostFree(ctx->msm_host_pinned);
  if (ctx->poly_host_pinned) (void)hipHostFree(ctx->poly_host_pinned);
  if (ctx->msm_side) (void)hipStreamDestroy(ctx->msm_side);
  for (hipEvent_t e : ctx->msm_events) (void)hipEventDestroy(e);
  for (StreamOrder* o : {&ctx->ord_ntt, &ctx->ord_msm, &ctx->ord_poly})
    if (o->ev) (void)hipEventDestroy(o->ev);
  prof_collect(ctx);
  for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
  if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* pm_last_error(const pm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" void* pm_ctx_stream(pm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int pm_sync(pm_ctx* ctx) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  PM_HIP(ctx, hipSetDevice(ctx->device));
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PM_OK;
}

extern "C" int pm_profile_enable(pm_ctx* ctx, int on) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  prof_collect(ctx);
  ctx->profile = on != 0;
  if (on) {
    ctx->prof_stats.clear();
    // create the event pairs up front: hipEventCreate costs a few microseconds and must not sit
    // between two kernels of the region being timed
    while (ctx->prof_pool.size() < 4096) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) break;
      ctx->prof_pool.push_back(e);
    }
  }
  return PM_OK;
}

extern "C" int pm_profile_select(pm_ctx* ctx, const char* kernel_name) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->profile_only = kernel_name ? kernel_name : "";
  return PM_OK;
}

extern "C" int pm_profile_read(pm_ctx* ctx, char* buf, size_t cap) {
  if (!ctx || !buf || cap == 0) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  prof_collect(ctx);
  std::string out;
  for (auto& kv : ctx->prof_stats) {
    char line[256];
    snprintf(line, sizeof line, "%s %llu %.6f\n", kv.first.c_str(), kv.second.count, kv.second.total_ms);
    out += line;
  }
  if (out.size() + 1 > cap) return set_err(ctx, PM_ERR_BAD_ARG, "profile buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return PM_OK;
}

extern "C" int pm_set_option(pm_ctx* ctx, const char* key, long value) {
  if (!ctx || !key) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!strcmp(key, "msm_window_bits")) {
    if (value != 0 && (value < 4 || value > 20)) return set_err(ctx, PM_ERR_BAD_ARG, "msm_window_bits out of range");
    ctx->opt_msm_window_bits = value;
    return PM_OK;
  }
  if (!strcmp(key, "msm_chunk")) {
    if (value < 0 || value > 4096) return set_err(ctx, PM_ERR_BAD_ARG, "msm_chunk out of range");
    ctx->opt_msm_chunk = value;
    return PM_OK;
  }
  if (!strcmp(key, "msm_max_pairs")) {
    if (value < 0 || value > 0x7fffffffL) return set_err(ctx, PM_ERR_BAD_ARG, "msm_max_pairs out of range");
    ctx->opt_msm_max_pairs = value;
    return PM_OK;
  }
  if (!strcmp(key, "msm_lb")) {
    if (value < 0 || value > 1024 || (value & (value - 1))) return set_err(ctx, PM_ERR_BAD_ARG, "msm_lb must be a power of two");
    ctx->opt_msm_lb = value;
    return PM_OK;
  }
  if (!strcmp(key, "msm_pipeline")) {
    ctx->opt_msm_pipeline = value != 0;
    return PM_OK;
  }
  if (!strcmp(key, "poly_lookback")) {
    if (value < 0 || value > 2) return set_err(ctx, PM_ERR_BAD_ARG, "poly_lookback must be 0 (off), 1 (auto) or 2 (always)");
    ctx->opt_poly_lookback = value;
    return PM_OK;
  }
  if (!strcmp(key, "comm_timeout_ms")) {
    if (value < 0 || value > 86400000L) return set_err(ctx, PM_ERR_BAD_ARG, "comm_timeout_ms must be 0 (off) .. 86400000");
    ctx->opt_comm_timeout_ms = value;
    return PM_OK;
  }
  if (!strcmp(key, "binv_quads")) {
    if (value < 0 || value > 1024) return set_err(ctx, PM_ERR_BAD_ARG, "binv_quads must be 0 (auto) .. 1024");
    ctx->opt_binv_quads = value;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_xcd")) {
    ctx->opt_ntt_xcd = value != 0;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_pipeline")) {
    ctx->opt_ntt_pipeline = value != 0;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_direct_tw")) {
    ctx->opt_ntt_direct_tw = value != 0;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_radix")) {
    if (value != 4 && value != 8) return set_err(ctx, PM_ERR_BAD_ARG, "ntt_radix must be 4 or 8");
    ctx->opt_ntt_radix = value;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_max_radix")) {
    if (value < 6 || value > 10) return set_err(ctx, PM_ERR_BAD_ARG, "ntt_max_radix must be 6..10");
    ctx->opt_ntt_max_radix = value;
    return PM_OK;
  }
  if (!strcmp(key, "ntt_tile_log")) {
    if (value != 0 && (value < 10 || value > 12)) return set_err(ctx, PM_ERR_BAD_ARG, "ntt_tile_log must be 0 (auto), 10, 11 or 12");
    ctx->opt_ntt_tile_log = value;
    return PM_OK;
  }
  return set_err(ctx, PM_ERR_BAD_ARG, std::string("unknown option ") + key);
}

// Pure host, no context: the host-side field arithmetic behind the MSM fold and the prover's challenge scalars
// (host_field.h).  op 0 = Fr product, 1 = Fp product, 2 = Fr inverse (binary extended Euclid), 3 = Fp inverse, 4 / 5 = the
// same inverses by exponentiation (the independent check); Montgomery form in and out; b is ignored by the inversions.
extern "C" int pm_test_host_field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!a || !out || op < 0 || op > 5 || (op < 2 && !b)) return PM_ERR_BAD_ARG;
  using namespace pm::host;
  {
    // operands are field elements: canonical limbs (< m).  The host code relies on it everywhere; this is where it is asserted
    const bool fp = op & 1;
    const int N = fp ? 6 : 4;
    const uint64_t* m = fp ? FP().m : FR().m;
    auto canonical = [&](const uint64_t* x) {
      for (int l = N - 1; l >= 0; --l)
        if (x[l] != m[l]) return x[l] < m[l];
      return false;
    };
    for (size_t i = 0; i < n; ++i)
      if (!canonical(a + N * i) || (op < 2 && !canonical(b + N * i))) return PM_ERR_BAD_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    if (op == 0 || op == 2 || op == 4) {
      HFr x, y;
      memcpy(x.l, a + 4 * i, 32);
      if (op == 0) memcpy(y.l, b + 4 * i, 32);
      const HFr r = op == 0 ? mul(x, y, FR()) : (op == 2 ? inv(x, FR()) : inv_fermat(x, FR()));
      memcpy(out + 4 * i, r.l, 32);
    } else {
      HFp x, y;
      memcpy(x.l, a + 6 * i, 48);
      if (op == 1) memcpy(y.l, b + 6 * i, 48);
      const HFp r = op == 1 ? mul(x, y, FP()) : (op == 3 ? inv(x, FP()) : inv_fermat(x, FP()));
      memcpy(out + 6 * i, r.l, 48);
    }
  }
  return PM_OK;
}

extern "C" int pm_test_field_op(pm_ctx* ctx, int op, const uint64_t* a, const uint64_t* b,
                                uint64_t* out, size_t n) {
  if (!ctx || !a || !out) return PM_ERR_BAD_ARG;
  if (op < 0 || op > 7) return PM_ERR_BAD_ARG;
  if (!b) {
    if (op < 6) return PM_ERR_BAD_ARG;
    b = a;   // the inversions ignore b
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return PM_OK;
  PM_HIP(ctx, hipSetDevice(ctx->device));
  const size_t esz = (op < 3 || op == 6) ? 32 : 48;
  void *da = nullptr, *db = nullptr, *dc = nullptr;
  struct Free3 {   // the temporaries go away on every path, error returns included
    void **a, **b, **c;
    ~Free3() {
      if (*a) (void)hipFree(*a);
      if (*b) (void)hipFree(*b);
      if (*c) (void)hipFree(*c);
    }
  } free3{&da, &db, &dc};
  PM_HIP(ctx, hipMalloc(&da, n * esz));
  PM_HIP(ctx, hipMalloc(&db, n * esz));
  PM_HIP(ctx, hipMalloc(&dc, n * esz));
  PM_HIP(ctx, hipMemcpyAsync(da, a, n * esz, hipMemcpyHostToDevice, ctx->stream));
  PM_HIP(ctx, hipMemcpyAsync(db, b, n * esz, hipMemcpyHostToDevice, ctx->stream));
  dim3 g((unsigned)((n + 255) / 256)), blk(256);
  const u32x4 *pa = (const u32x4*)da, *pb = (const u32x4*)db;
  u32x4* pc = (u32x4*)dc;
  switch (op) {
    case 0: hipLaunchKernelGGL((field_op_kernel<FrP, 0>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 1: hipLaunchKernelGGL((field_op_kernel<FrP, 1>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 2: hipLaunchKernelGGL((field_op_kernel<FrP, 2>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 3: hipLaunchKernelGGL((field_op_kernel<FpP, 0>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 4: hipLaunchKernelGGL((field_op_kernel<FpP, 1>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 5: hipLaunchKernelGGL((field_op_kernel<FpP, 2>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 6: hipLaunchKernelGGL((field_op_kernel<FrP, 3>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
    case 7: hipLaunchKernelGGL((field_op_kernel<FpP, 3>), g, blk, 0, ctx->stream, pa, pb, pc, n); break;
  }
  PM_HIP(ctx, hipGetLastError());
  PM_HIP(ctx, hipMemcpyAsync(out, dc, n * esz, hipMemcpyDeviceToHost, ctx->stream));
  PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PM_OK;
}
