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

#include <cstring>
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
  if (!ctx->comm) return set_err(ctx, PM_ERR_EXCHANGE, "no communicator: pm_comm_init first");
  Rccl& r = rccl();
  if (!r.p2p()) return set_err(ctx, PM_ERR_EXCHANGE, "librccl has no ncclSend / ncclRecv");
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  ncclResult_t nrc = r.GroupStart();
  for (int p = 0; p < ctx->comm_world && nrc == ncclSuccess; ++p) {
    nrc = r.Send((const char*)d_send + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclUint8, p, comm, st);
    if (nrc == ncclSuccess) nrc = r.Recv((char*)d_recv + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclUint8, p, comm, st);
  }
  const ncclResult_t erc = r.GroupEnd();
  if (nrc == ncclSuccess) nrc = erc;
  if (nrc != ncclSuccess) return set_err(ctx, PM_ERR_EXCHANGE, std::string("ncclSend/ncclRecv: ") + r.GetErrorString(nrc));
  return PM_OK;
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
  if (ctx->comm) return set_err(ctx, PM_ERR_BAD_ARG, "the context already has a communicator");
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
  if (!ctx->comm) return PM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
  (void)hipFree(ctx->comm_send);
  (void)hipFree(ctx->comm_recv);
  (void)hipHostFree(ctx->comm_host);
  ctx->comm = ctx->comm_send = ctx->comm_recv = ctx->comm_host = nullptr;
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
  const ncclResult_t nrc = r.AllGather(ctx->comm_send, ctx->comm_recv, MSG_WORDS, kNcclUint64, (ncclComm_t)ctx->comm, st);
  if (nrc != ncclSuccess) return set_err(ctx, PM_ERR_EXCHANGE, std::string("ncclAllGather: ") + r.GetErrorString(nrc));
  PM_HIP(ctx, hipMemcpyAsync(h_recv, ctx->comm_recv, MSG_WORDS * 8 * (size_t)world, hipMemcpyDeviceToHost, st));
  PM_HIP(ctx, hipStreamSynchronize(st));
  if (!staged) return set_err(ctx, PM_ERR_EXCHANGE, "staging the exchange message failed: this rank aborted the exchange");
  memcpy(gathered, h_recv, MSG_WORDS * 8 * (size_t)world);
  return PM_OK;
}

// k = 0: this rank gave up (abort marker).  Returns PM_ERR_EXCHANGE on every rank if any rank did.
extern "C" int pm_g1_allgather_fold(pm_ctx* ctx, uint64_t* xyz, uint32_t k) {
  if (!ctx || (k && !xyz) || k > PM_COMM_MAX_POINTS) return PM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
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
