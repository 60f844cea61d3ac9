// Library context: one GPU, one stream, cached domain tables, reusable workspaces.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/plonk_mi355x.h"

namespace pm {

class ExchangeWatch;                                    // comm.hip
void exchange_watch_free(ExchangeWatch* w);

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

// A group of shared resources (scratch buffers, lazily built tables) that calls on different HIP
// streams would otherwise race on.  Every call that touches the group holds an OrderScope: on entry,
// when the stream differs from the one that used the group last, the new stream first waits for the
// event the PREVIOUS call recorded on its own stream when it ended; on exit the call records that
// event for its successor.  The library never touches a stream after the call it was passed to
// returned (a caller may destroy it), and only the library's own work is ordered.  A context behaves
// like one in-order queue per group even when its *_dev entry points are driven from several
// streams; groups are independent: an NTT on one stream and an MSM on another still overlap.
// Calls on the context's OWN stream (what a caller gets by passing no stream; the library creates and destroys it)
// record lazily -- when a call on another stream next needs the event -- because an event record per call is +2 % on
// a forward + inverse 2^20 NTT (profiles/r03_order_scope_ab.txt).
struct StreamOrder {
  hipStream_t last = nullptr;
  bool used = false;      // an earlier call used the group
  bool pending = false;   // ... on the context's own stream, and its closing event is not recorded yet
  hipEvent_t ev = nullptr;
};

struct NttDomainTables {  // per (log_n, direction)
  void* tw_hi = nullptr;
  void* tw_lo = nullptr;
  void* cs_hi = nullptr;
  void* cs_lo = nullptr;
  void* pass_tw[4] = {nullptr, nullptr, nullptr, nullptr};  // direct twiddles of pass i (wide planes)
  unsigned pass_tw_key[4] = {0, 0, 0, 0};                   // (log_ns << 8 | S) << 1 | last: what it was built for
  unsigned lh = 0;
};

}  // namespace pm

struct pm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_in = nullptr, copy_out = nullptr;    // host-pointer batch calls: H2D / D2H beside the compute stream
  std::mutex mu;
  std::string err;
  // NTT caches
  std::map<unsigned, void*> step_tw[2];                 // [dir][S] -> device table
  std::map<unsigned, void*> step4_tw[2];                // same for the radix-4 kernels
  std::map<unsigned, pm::NttDomainTables> domain[2];    // [dir][log_n]
  pm::DeviceBuffer ntt_tmp[2];
  pm::DeviceBuffer io_in, io_out;                       // staging for host-pointer calls
  pm::StreamOrder ord_ntt, ord_msm, ord_poly;           // cross-stream ordering of the shared scratch + tables
  // host timeline (diagnostic, PM_HOST_MARKS=1 in the environment): (label, microseconds) pairs, printed by the prover.  Not
  // synchronised: meant for one proving thread per context (several threads in one context would interleave their marks)
  bool marks_on = false;
  std::vector<std::pair<const char*, double>> marks;
  unsigned long long stat_alltoall_calls = 0, stat_alltoall_bytes = 0, stat_allgather_calls = 0, stat_transpose_steps = 0;   // pm_comm_stats
  int calls_holding_tables = 0;                         // calls that keep table pointers across unlocked sections (four-step NTT): pm_trim refuses
  // MSM workspaces
  pm::DeviceBuffer msm_ws;
  pm::DeviceBuffer msm_ctl;                             // control block of the bucket fill (msm_sort.hip.h): zero when idle
  unsigned msm_ctl_cap = 0;                             // partitions it is laid out for
  hipStream_t msm_side = nullptr;                       // second stream of the piece pipeline of a batched MSM (msm.hip)
  std::vector<hipEvent_t> msm_events;                   // "piece i has left the accumulate"
  pm::DeviceBuffer msm_scalars;
  pm::DeviceBuffer poly_ws;                             // scratch of the polynomial helpers
  pm::DeviceBuffer poly_tab;                            // power tables of pm_fr_poly_ruffini_dev
  void* msm_host_pinned = nullptr;
  void* poly_host_pinned = nullptr;                     // 3 x PM_LINCOMB_MAX results of the evaluation batches (poly.hip)
  size_t msm_host_pinned_bytes = 0;
  // multi-GPU exchange (comm.hip): RCCL communicator of this rank, device and pinned staging buffers
  void* comm = nullptr;
  bool comm_dead = false;                               // an exchange timed out and the communicator was aborted (comm.hip)
  pm::ExchangeWatch* comm_watch = nullptr;              // the deadline thread of guarded exchanges (option comm_timeout_ms)
  int comm_rank = 0, comm_world = 1;
  void *comm_send = nullptr, *comm_recv = nullptr, *comm_host = nullptr;
  // opt-in per-kernel timing (hipEvents on the launch stream; read by bench.py)
  bool profile = false;
  std::string profile_only;      // when set, only scopes with exactly this name record events
  struct ProfPending {
    hipEvent_t start, stop;
    const char* name;
  };
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  struct ProfStat {
    double total_ms = 0;
    unsigned long long count = 0;
  };
  std::map<std::string, ProfStat> prof_stats;
  // tunables
  long opt_msm_window_bits = 0;  // 0 = auto
  long opt_ntt_tile_log = 0;     // 0 = auto
  long opt_ntt_radix = 4;        // in-tile butterfly radix: 4 (4 elements per thread) or 8
  long opt_ntt_xcd = 1;          // XCD-aware blockIdx -> tile mapping
  long opt_ntt_pipeline = 1;     // host-pointer batch calls: overlap H2D / transform / D2H per vector
  long opt_ntt_max_radix = 10;   // log2 of the largest pass radix (multi-pass plans)
  long opt_ntt_direct_tw = 1;    // inter-pass twiddles from per-pass N x 36 B tables (1) or from the two-level tables + one product (0)
  long opt_msm_chunk = 0;        // 0 = auto (entries per thread in the level-1 accumulate)
  long opt_msm_max_pairs = 0;    // 0 = 2^31 - 1; a batched MSM with more (digit, point) pairs runs in halves
  long opt_msm_lb = 0;           // 0 = auto (buckets per thread in the bucket reduce)
  long opt_comm_timeout_ms = 0;  // > 0: every exchange on the context's RCCL communicator runs under this deadline (comm.hip)
  long opt_binv_quads = 0;       // batch inversion: quads (4 elements) per thread and inversion; 0 = auto
  long opt_poly_lookback = 1;    // prefix product in one pass (decoupled look-back) instead of totals / scan / replay
  long opt_msm_pipeline = 0;     // 1: a batched MSM runs as up to four pieces on two streams (measured: loses, see msm.hip)
  int num_cus = 256;
};

struct pm_bases {
  void* d_xy = nullptr;  // n x 96 bytes, affine Montgomery, (0,0) = identity
  void* d_table = nullptr;   // optional: ceil(256/table_c) rows of n points, row w = 2^(table_c w) * bases
  unsigned table_c = 0;      // window bits the table was built for (0 = no table)
  size_t n = 0;
  int device = 0;
};

namespace pm {

int set_err(pm_ctx* ctx, int code, const std::string& msg);
// Scoped kernel timer: records an event pair around the launches issued while it lives.
struct ProfScope {
  pm_ctx* ctx;
  hipStream_t st;
  hipEvent_t stop = nullptr;
  ProfScope(pm_ctx* c, hipStream_t s, const char* name);
  ~ProfScope();
};
int prof_collect(pm_ctx* ctx);
int ensure_buffer(pm_ctx* ctx, DeviceBuffer& b, size_t bytes);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-function attribute of the PROCESS (per device), not of a
// context: the largest value any context asked for is remembered process-wide and only ever raised (ADVICE r03:
// with a per-context record, context B could lower what context A had set and A's next launch would fail).
int raise_lds_limit(pm_ctx* ctx, const void* fn, size_t bytes);
int comm_alltoall(pm_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_peer, hipStream_t st);   // comm.hip
// the fixed-size exchange message of the sharded / distributed prover: [count | 288 payload words]; count 0 = abort marker
static constexpr size_t COMM_MSG_WORDS = 1 + 18 * (size_t)PM_COMM_MAX_POINTS;
int comm_allgather_msg(pm_ctx* ctx, const uint64_t* msg, uint64_t* gathered);   // comm.hip; caller holds ctx->mu
int fold_gathered(const uint64_t* msgs, int world, uint32_t k_local, uint64_t* out_xyz);   // comm.hip
// quotient kernel over `rows` rows of the coset (4 rows points); halo: the arrays read at index + 4 (z, wires 0 1 3)
// carry four more points after the last row instead of wrapping around (plonk_rounds.hip)
int plonk_quotient_rows(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t rows, bool halo, void* d_out, void* hip_stream);
// planar layout of the distributed prover (plonk_rounds.hip, struct QuotLayout): four planes [s][rows] per array in the
// block-transposed order of the rank-split transform, halo rows [4][n2] for a, b, d (halo_w[0], [1], [3]) and z
struct QuotPlanar {
  uint32_t n2, rot;
  const void* halo_w[4];
  const void* halo_z;
};
int plonk_quotient_layout(pm_ctx* ctx, const pm_plonk_quotient_args* args, size_t rows, bool halo, const QuotPlanar* planar,
                          void* d_out, void* hip_stream);
inline void host_mark(pm_ctx* ctx, const char* label) {
  if (ctx && ctx->marks_on)
    ctx->marks.emplace_back(label, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
int fr_batch_inverse_mul(pm_ctx* ctx, void* d_inout, const void* d_mul, size_t n, void* hip_stream);   // poly.hip: mul_i / v_i in place
int poly_evaluate_groups(pm_ctx* ctx, uint32_t groups, const uint32_t* k, const void* const* const* polys,
                         const uint64_t* const* points, uint64_t* const* outs, size_t n);   // poly.hip
int coset_expand(pm_ctx* ctx, const void* const* d_src, uint32_t count, const void* d_gs_pow, size_t m, void* d_out);
int sigma_evals_from_index(pm_ctx* ctx, const int64_t* idx, size_t count, uint32_t log_n, const uint64_t omega[4],
                           const uint64_t k[3][4], void* d_out);
struct OrderScope {
  pm_ctx* ctx;
  StreamOrder& o;
  hipStream_t st;
  int rc;
  OrderScope(pm_ctx* c, StreamOrder& ord, hipStream_t s);
  ~OrderScope();
  OrderScope(const OrderScope&) = delete;
  OrderScope& operator=(const OrderScope&) = delete;
};

#define PM_HIP(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return pm::set_err(ctx, _e == hipErrorOutOfMemory ? PM_ERR_OOM : PM_ERR_HIP,          \
                         std::string(#call) + ": " + hipGetErrorString(_e));                \
  } while (0)

}  // namespace pm
