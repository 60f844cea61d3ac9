// The one exchange step of the hot path, inside the library (SURVEY.md section 8b "RCCL comm over the
// listed GPUs", section 8e): one process per GPU, every rank holds a pm_ctx, the MSM is sharded by points
// and the ranks exchange k partial G1 points -- ncclAllGather of a fixed 2312-byte message per rank over
// RCCL / xGMI (RCCL has no reduction under the group law), then the same fold on every rank.  Used by
// pm_plonk_prove_sharded / pm_plonk_key_commit_sharded when no exchange callback is given, and directly
// through pm_g1_allgather_fold.  RCCL is bound at run time (dlopen of librccl.so.1 -- the copy a host
// runtime such as PyTorch-ROCm already loaded, or the one in /opt/rocm/lib), so the library itself loads
// on machines without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the entry points are bound with dlsym, nothing links against librccl

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "context.h"

namespace pm {

// the handful of RCCL entry points used, with the prototypes of the installed rccl.h (the hand-written
// ones of r02 could only have failed on a multi-GPU node)
static_assert(sizeof(ncclUniqueId) == PM_COMM_ID_BYTES, "PM_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
typedef ncclUniqueId ncclUniqueIdRaw;
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather && GetErrorString; }
  bool p2p() const { return Send && Recv && GroupStart && GroupEnd; }
};
static Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  });
  return r;
}
static const ncclDataType_t kNcclUint8 = ncclUint8;
static const ncclDataType_t kNcclUint64 = ncclUint64;

// ---- a dead peer inside an exchange (r06; SURVEY.md section 5 "failure detection") ---------------------------------
// The all-gather's abort marker covers a rank that fails BEFORE an exchange: it still takes part and every rank gets
// PM_ERR_EXCHANGE.  A rank that dies, or fails between two all-to-alls, leaves its peers inside ncclAllGather / the grouped
// ncclSend / ncclRecv -- on the host while connections are set up, on the device afterwards -- and RCCL has no timeout.
// Option comm_timeout_ms (default 0 = off): every exchange on the context's communicator then runs under a deadline
// (the enqueue AND the wait for its completion: with the option on, the all-to-all is synchronised right away).  When
// the deadline passes, the context's watch thread calls ncclCommAbort on the communicator -- the documented way to end
// its in-flight operations from another thread -- the blocked call returns, the exchange reports PM_ERR_EXCHANGE, and the
// context marks its communicator dead: every later exchange fails at once with the same code until pm_comm_destroy +
// pm_comm_init.  The transports behind the callbacks (dist.LocalGroup / DistGroup) have their own: a broken barrier.
// Not testable over xGMI on a one-GPU box; the deadline logic is tested against a stub table (pm_test_comm_deadline).
class ExchangeWatch {
  std::mutex m;
  std::condition_variable cv;
  std::thread th;
  bool armed = false, fired = false, busy = false, quit = false;
  std::chrono::steady_clock::time_point deadline;
  std::function<void()> on_timeout;

  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return armed || quit; });
      if (quit) return;
      if (cv.wait_until(lk, deadline, [&] { return !armed || quit; })) continue;   // finished in time (or shutting down)
      fired = busy = true;                       // the deadline passed with the exchange still in flight
      const std::function<void()> f = on_timeout;
      lk.unlock();
      f();                                       // ncclCommAbort: the blocked call in the other thread returns
      lk.lock();
      busy = armed = false;
      cv.notify_all();
    }
  }

 public:
  ~ExchangeWatch() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  void arm(long ms, std::function<void()> f) {
    std::lock_guard<std::mutex> lk(m);
    if (!th.joinable()) th = std::thread([this] { run(); });
    armed = true;
    fired = false;
    deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    on_timeout = std::move(f);
    cv.notify_all();
  }
  bool disarm() {   // -> the deadline fired (and the abort has returned)
    std::unique_lock<std::mutex> lk(m);
    armed = false;
    cv.notify_all();
    cv.wait(lk, [&] { return !busy; });
    return fired;
  }
};
void exchange_watch_free(ExchangeWatch* w) { delete w; }

// What a guarded exchange needs of its owner (a pm_ctx, or the test hook's stand-in)
struct CommGuard {
  void** comm;
  bool* dead;
  long timeout_ms;
  ExchangeWatch** watch;
  std::string* err;
};
static const char* const kDeadMsg =
    "the communicator was aborted after an exchange timed out: pm_comm_destroy, then pm_comm_init again on every rank";
// body: the RCCL calls of one exchange and the wait for their completion -> a PM_* code
template <class F>
static int guarded_exchange(const CommGuard& g, const Rccl& r, const char* what, F&& body) {
  if (*g.dead) {
    *g.err = kDeadMsg;
    return PM_ERR_EXCHANGE;
  }
  if (g.timeout_ms <= 0 || !r.CommAbort) return body();
  if (!*g.watch) *g.watch = new ExchangeWatch();
  void* const comm = *g.comm;
  const decltype(r.CommAbort) abort_fn = r.CommAbort;
  (*g.watch)->arm(g.timeout_ms, [comm, abort_fn] { (void)abort_fn((ncclComm_t)comm); });
  const int rc = body();
  if ((*g.watch)->disarm()) {
    *g.dead = true;
    *g.comm = nullptr;            // ncclCommAbort has freed it
    *g.err = std::string(what) + ": no completion within " + std::to_string(g.timeout_ms) +
             " ms (comm_timeout_ms): a peer is gone; the communicator was aborted (ncclCommAbort) and is dead";
    return PM_ERR_EXCHANGE;
  }
  return rc;
}
static CommGuard guard_of(pm_ctx* ctx) {
  return CommGuard{&ctx->comm, &ctx->comm_dead, ctx->opt_comm_timeout_ms, &ctx->comm_watch, &ctx->err};
}

// message of one rank: [count | PM_COMM_MAX_POINTS x 18 limbs]; count = 0 is the abort marker of a rank
// whose local work failed (it still enters the collective, so no peer blocks)
static constexpr size_t MSG_WORDS = COMM_MSG_WORDS;
static_assert(COMM_MSG_WORDS == 1 + 18 * (size_t)PM_COMM_MAX_POINTS, "message = count + 16 points");

// Fold what the ranks sent: out[j] = sum over ranks of point j.  -> PM_OK, or PM_ERR_EXCHANGE when a rank
// sent the abort marker or the counts disagree.
int fold_gathered(const uint64_t* msgs, int world, uint32_t k_local, uint64_t* out_xyz) {
  for (int r = 0; r < world; ++r)
    if (msgs[r * MSG_WORDS] == 0 || msgs[r * MSG_WORDS] != k_local) return PM_ERR_EXCHANGE;
  std::vector<uint64_t> parts((size_t)world * 18);
  for (uint32_t j = 0; j < k_local; ++j) {
    for (int r = 0; r < world; ++r) memcpy(&parts[(size_t)r * 18], msgs + r * MSG_WORDS + 1 + 18 * (size_t)j, 144);
    int rc = pm_g1_fold(parts.data(), (size_t)world, out_xyz + 18 * (size_t)j);
    if (rc) return rc;
  }
  return PM_OK;
}

// All-to-all of equal blocks on the context's communicator and the given stream: block p of `send` goes to
// rank p, block p of `recv` comes from rank p (grouped ncclSend / ncclRecv: one direct xGMI transfer per peer).
// The caller holds no lock that the stream's earlier work needs.  Used by the rank-split NTT (ntt.hip).
int comm_alltoall(pm_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_peer, hipStream_t st) {
  if (ctx->comm_dead) return set_err(ctx, PM_ERR_EXCHANGE, kDeadMsg);
  if (!ctx->comm) return set_err(ctx, PM_ERR_EXCHANGE, "no communicator: pm_comm_init first");
  Rccl& r = rccl();
  if (!r.p2p()) return set_err(ctx, PM_ERR_EXCHANGE, "librccl has no ncclSend / ncclRecv");
  const int world = ctx->comm_world;
  const bool deadline = ctx->opt_comm_timeout_ms > 0;
  std::string detail;
  const int rc = guarded_exchange(guard_of(ctx), r, "all-to-all (grouped ncclSend / ncclRecv)", [&]() -> int {
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ncclResult_t nrc = r.GroupStart();
    for (int p = 0; p < world && nrc == ncclSuccess; ++p) {
      nrc = r.Send((const char*)d_send + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclUint8, p, comm, st);
      if (nrc == ncclSuccess) nrc = r.Recv((char*)d_recv + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclUint8, p, comm, st);
    }
    const ncclResult_t erc = r.GroupEnd();
    if (nrc == ncclSuccess) nrc = erc;
    if (nrc != ncclSuccess) {
      detail = std::string("ncclSend/ncclRecv: ") + r.GetErrorString(nrc);
      return PM_ERR_EXCHANGE;
    }
    // under a deadline the exchange is waited for here, where the watch can still end it
    if (deadline && hipStreamSynchronize(st) != hipSuccess) {
      (void)hipGetLastError();
      detail = "hipStreamSynchronize after the all-to-all failed";
      return PM_ERR_EXCHANGE;
    }
    return PM_OK;
  });
  if (rc && !ctx->comm_dead && !detail.empty()) return set_err(ctx, rc, detail);
  return rc;   // (a timeout: guarded_exchange has written ctx->err)
}

}  // namespace pm

using namespace pm;

extern "C" int pm_comm_unique_id(uint8_t id[PM_COMM_ID_BYTES]) {
  if (!id) return PM_ERR_BAD_ARG;
  Rccl& r = rccl();
  if (!r.ok()) return PM_ERR_EXCHANGE;
  ncclUniqueIdRaw raw;
  if (r.GetUniqueId(&raw) != ncclSuccess) return PM_ERR_EXCHANGE;
  memcpy(id, raw.internal, PM_COMM_ID_BYTES);
  return PM_OK;
}

extern "C" int pm_comm_init(pm_ctx* ctx, const uint8_t id[PM_COMM_ID_BYTES], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->comm || ctx->comm_dead) return set_err(ctx, PM_ERR_BAD_ARG, "the context already has a communicator (pm_comm_destroy first)");
  Rccl& r = rccl();
  if (!r.ok()) return set_err(ctx, PM_ERR_EXCHANGE, "librccl.so.1 could not be loaded");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueIdRaw raw;
  memcpy(raw.internal, id, PM_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t nrc = r.CommInitRank(&comm, world, raw, rank);
  if (nrc != ncclSuccess) return set_err(ctx, PM_ERR_EXCHANGE, std::string("ncclCommInitRank: ") + r.GetErrorString(nrc));
  hipError_t e = hipMalloc(&ctx->comm_send, MSG_WORDS * 8);
  if (e == hipSuccess) e = hipMalloc(&ctx->comm_recv, MSG_WORDS * 8 * (size_t)world);
  if (e == hipSuccess) e = hipHostMalloc(&ctx->comm_host, MSG_WORDS * 8 * ((size_t)world + 1), hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)r.CommDestroy(comm);
    if (ctx->comm_send) (void)hipFree(ctx->comm_send);
    if (ctx->comm_recv) (void)hipFree(ctx->comm_recv);
    ctx->comm_send = ctx->comm_recv = nullptr;
    return set_err(ctx, PM_ERR_OOM, "communicator buffers");
  }
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return PM_OK;
}

extern "C" int pm_comm_destroy(pm_ctx* ctx) {
  if (!ctx) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  exchange_watch_free(ctx->comm_watch);
  ctx->comm_watch = nullptr;
  if (!ctx->comm && !ctx->comm_dead) return PM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) (void)rccl().CommDestroy((ncclComm_t)ctx->comm);     // (an aborted communicator is already freed)
  (void)hipFree(ctx->comm_send);
  (void)hipFree(ctx->comm_recv);
  (void)hipHostFree(ctx->comm_host);
  ctx->comm = ctx->comm_send = ctx->comm_recv = ctx->comm_host = nullptr;
  ctx->comm_dead = false;
  ctx->comm_world = 1;
  ctx->comm_rank = 0;
  return PM_OK;
}

extern "C" int pm_comm_info(const pm_ctx* ctx, int* rank, int* world) {
  if (!ctx) return PM_ERR_BAD_ARG;
  if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
  if (world) *world = ctx->comm ? ctx->comm_world : 1;
  return PM_OK;
}

// All-gather of the fixed-size message on the context's communicator: gathered = world x COMM_MSG_WORDS words (host
// memory).  msg[0] = 0 is the abort marker; a rank that cannot stage its message still enters the collective with it
// (written by a memset), so that no peer waits in ncclAllGather for ever (it has no timeout).  The caller holds ctx->mu.
int pm::comm_allgather_msg(pm_ctx* ctx, const uint64_t* msg, uint64_t* gathered) {
  if (ctx->comm_dead) return set_err(ctx, PM_ERR_EXCHANGE, kDeadMsg);
  if (!ctx->comm) return set_err(ctx, PM_ERR_EXCHANGE, "no communicator: pm_comm_init first");
  PM_HIP(ctx, hipSetDevice(ctx->device));
  Rccl& r = rccl();
  const int world = ctx->comm_world;
  uint64_t* h_send = (uint64_t*)ctx->comm_host;
  uint64_t* h_recv = h_send + MSG_WORDS;
  memcpy(h_send, msg, MSG_WORDS * 8);
  hipStream_t st = ctx->stream;
  bool staged = hipMemcpyAsync(ctx->comm_send, h_send, MSG_WORDS * 8, hipMemcpyHostToDevice, st) == hipSuccess;
  if (!staged) {
    (void)hipGetLastError();
    (void)hipMemsetAsync(ctx->comm_send, 0, 8, st);
  }
  std::string detail;
  const int rc = guarded_exchange(guard_of(ctx), r, "all-gather (ncclAllGather)", [&]() -> int {
    const ncclResult_t nrc = r.AllGather(ctx->comm_send, ctx->comm_recv, MSG_WORDS, kNcclUint64, (ncclComm_t)ctx->comm, st);
    if (nrc != ncclSuccess) {
      detail = std::string("ncclAllGather: ") + r.GetErrorString(nrc);
      return PM_ERR_EXCHANGE;
    }
    hipError_t e = hipMemcpyAsync(h_recv, ctx->comm_recv, MSG_WORDS * 8 * (size_t)world, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      detail = std::string("waiting for the all-gather: ") + hipGetErrorString(e);
      return PM_ERR_HIP;
    }
    return PM_OK;
  });
  if (rc) return ctx->comm_dead || detail.empty() ? rc : set_err(ctx, rc, detail);
  if (!staged) return set_err(ctx, PM_ERR_EXCHANGE, "staging the exchange message failed: this rank aborted the exchange");
  memcpy(gathered, h_recv, MSG_WORDS * 8 * (size_t)world);
  return PM_OK;
}

// k = 0: this rank gave up (abort marker).  Returns PM_ERR_EXCHANGE on every rank if any rank did.
extern "C" int pm_g1_allgather_fold(pm_ctx* ctx, uint64_t* xyz, uint32_t k) {
  if (!ctx || (k && !xyz) || k > PM_COMM_MAX_POINTS) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->comm_dead) return set_err(ctx, PM_ERR_EXCHANGE, kDeadMsg);
  if (!ctx->comm) return set_err(ctx, PM_ERR_EXCHANGE, "no communicator: pm_comm_init first");
  std::vector<uint64_t> msg(MSG_WORDS, 0), gathered(MSG_WORDS * (size_t)ctx->comm_world);
  msg[0] = k;
  if (k) memcpy(msg.data() + 1, xyz, 144 * (size_t)k);
  const int xrc = comm_allgather_msg(ctx, msg.data(), gathered.data());
  if (xrc) return xrc;
  if (k == 0) return set_err(ctx, PM_ERR_EXCHANGE, "this rank aborted the exchange");
  const int rc = fold_gathered(gathered.data(), ctx->comm_world, k, xyz);
  if (rc == PM_ERR_EXCHANGE) return set_err(ctx, rc, "a peer rank aborted the exchange (or sent a different count)");
  return rc;
}

// Test hook: the fold of pm_g1_allgather_fold on messages given by the caller (world x 289 words), so the
// packing, the abort marker and the count check can be exercised without several GPUs.
extern "C" int pm_test_fold_gathered(const uint64_t* msgs, int world, uint32_t k, uint64_t* out_xyz) {
  if (!msgs || !out_xyz || world < 1 || k == 0 || k > PM_COMM_MAX_POINTS) return PM_ERR_BAD_ARG;
  return fold_gathered(msgs, world, k, out_xyz);
}

// Test hook (no GPU, no RCCL): the deadline logic of a guarded exchange against a STUB table whose ncclAllGather blocks until
// ncclCommAbort is called on its communicator (a peer that never arrives), or returns at once (peer_answers != 0).
//   first_rc     the guarded all-gather: PM_OK, or PM_ERR_EXCHANGE once the watch has aborted the communicator
//   elapsed_ms   how long it took (the deadline, not for ever)
//   second_rc    the next exchange on the same communicator: fails at once when the first one timed out
//   aborts       how often the stub's ncclCommAbort ran
namespace {
std::mutex stub_m;
std::condition_variable stub_cv;
bool stub_aborted = false;
int stub_abort_calls = 0, stub_gathers = 0;
ncclResult_t stub_allgather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) {
  std::unique_lock<std::mutex> lk(stub_m);
  ++stub_gathers;
  stub_cv.wait(lk, [] { return stub_aborted; });
  return ncclSuccess;                 // (what RCCL's enqueue returns: the failure shows in the wait)
}
ncclResult_t stub_allgather_ok(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) {
  std::lock_guard<std::mutex> lk(stub_m);
  ++stub_gathers;
  return ncclSuccess;
}
ncclResult_t stub_abort(ncclComm_t) {
  {
    std::lock_guard<std::mutex> lk(stub_m);
    stub_aborted = true;
    ++stub_abort_calls;
  }
  stub_cv.notify_all();
  return ncclSuccess;
}
}  // namespace
extern "C" int pm_test_comm_deadline(long timeout_ms, int peer_answers, int* first_rc, long* elapsed_ms, int* second_rc,
                                     int* aborts, char* err_out, size_t err_cap) {
  if (!first_rc || !elapsed_ms || !second_rc || !aborts) return PM_ERR_BAD_ARG;
  {
    std::lock_guard<std::mutex> lk(stub_m);
    stub_aborted = false;
    stub_abort_calls = stub_gathers = 0;
  }
  Rccl stub;
  stub.AllGather = peer_answers ? stub_allgather_ok : stub_allgather;
  stub.CommAbort = stub_abort;
  int fake = 0;
  void* comm = &fake;
  bool dead = false;
  ExchangeWatch* watch = nullptr;
  std::string err;
  const CommGuard g{&comm, &dead, timeout_ms, &watch, &err};
  auto exchange = [&]() {
    return guarded_exchange(g, stub, "all-gather (stub)", [&]() -> int {
      return stub.AllGather(nullptr, nullptr, 0, kNcclUint64, (ncclComm_t)comm, nullptr) == ncclSuccess ? PM_OK : PM_ERR_EXCHANGE;
    });
  };
  const auto t0 = std::chrono::steady_clock::now();
  *first_rc = exchange();
  *elapsed_ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  const std::string first_err = err;            // pm_last_error after the first exchange
  const int gathers_before = stub_gathers;
  *second_rc = exchange();
  if (dead && stub_gathers != gathers_before) *second_rc = -100;   // (not a PM_* code) a dead communicator must not be entered again
  *aborts = stub_abort_calls;
  if (err_out && err_cap) {
    strncpy(err_out, first_err.c_str(), err_cap - 1);
    err_out[err_cap - 1] = 0;
  }
  exchange_watch_free(watch);
  return PM_OK;
}
