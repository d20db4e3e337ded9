// z-slab driver of the 3D Jacobi time step (include/fluidnet_hip.h, "z-slab decomposition"): host code only, written on
// top of the library's own C ABI -- fnx_advect_step / fnx_pre_projection / fnx_jacobi_pass[2] / fnx_post_projection with
// FnxGrid's slab view (z_offset, D_global) and compute windows (k_begin, k_end) -- plus two communicators: RCCL
// (ncclSend/ncclRecv between z-neighbours, loaded with dlopen) and an in-process one (several slabs of a domain driven by
// host threads of one process; the lock-step tests and single-process multi-device runs).
//
// The schedules are the ones fluidnet_cxx_amd/slab.py documents ("deep_first", the default; "edge_first"; "last_pass"): per step
//   1. ghost exchange of U, density (4 planes) posted; the planes whose advection reads no ghost plane are advected
//      meanwhile, the two 4-plane edge windows after it has landed
//   2. BC / buoyancy / wall stage + divergence on the owned planes; exchange of div (w-1 planes) posted
//   3. blocks of w Jacobi sweeps.  deep_first: the deep parts of all passes of a block (they read no ghost plane) while the
//      previous exchange is in flight, then the wait, then the edge parts (a chain of short launches that ends in the w owned
//      planes each neighbour needs), their exchange posted.  edge_first: the edge parts first, the interior parts behind the
//      posted exchange (div exchanged blocking)
//   4. exchange of p (1 plane), velocity update + wall BCs + BCs on the owned planes
// Every owned cell goes through the arithmetic of the single-domain step: same bits (tests/test_slab.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/fluidnet_hip.h"
#include "fnx_kernels.h"
#include "fnx_cnn.h"

namespace {

#define SLAB_HIP(expr)                                                                                     \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return fnx::set_error(FNX_EHIP, "HIP error: %s (%s)", hipGetErrorString(e_), #expr); \
  } while (0)
#define SLAB_OK(expr)            \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != FNX_OK) return rc_; \
  } while (0)

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------------
// RCCL communicator.  librccl is resolved at run time: libfluidnet_hip.so has no link-time dependency on it, and a
// process that already carries an RCCL (torch's) gets that one.
// ---------------------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*CommAbort)(NcclComm) = nullptr;       // optional
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0, kNcclMax = 2;      // ncclDataType_t / ncclRedOp_t values of nccl.h

int rccl_api(RcclApi** out) {
  static RcclApi api;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!api.lib) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return fnx::set_error(FNX_ECOMM, "librccl.so could not be loaded: %s", dlerror());
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(NcclUniqueId*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
    api.CommAbort = (int (*)(NcclComm))sym("ncclCommAbort");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclRecv");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclAllReduce");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send ||
        !api.Recv || !api.AllReduce) {
      api.lib = nullptr;
      return fnx::set_error(FNX_ECOMM, "librccl.so lacks an expected nccl* symbol");
    }
  }
  *out = &api;
  return FNX_OK;
}

struct RcclCtx { RcclApi* api; NcclComm comm; int rank, nranks; };
#define NCCL_OK(ctx, expr)                                                                                         \
  do {                                                                                                             \
    int r_ = (expr);                                                                                               \
    if (r_ != 0) return fnx::set_error(FNX_ECOMM, "RCCL error %d (%s) in %s", r_,                                  \
                                       (ctx)->api->GetErrorString ? (ctx)->api->GetErrorString(r_) : "?", #expr);  \
  } while (0)

int rccl_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) {
  RcclCtx* c = (RcclCtx*)vctx;
  hipStream_t s = (hipStream_t)stream;
  NCCL_OK(c, c->api->GroupStart());
  // a failed Send / Recv must not leave the group open (it would poison every later call of this thread): the first
  // error is kept, the group is closed, then the error is reported
  int first = 0; const char* where = "";
  auto rec = [&](int r, const char* w) { if (r != 0 && first == 0) { first = r; where = w; } };
  for (int i = 0; i < nsegs && first == 0; ++i) {
    const FnxSlabSeg& g = segs[i];
    if (c->rank > 0 && g.send_lo) rec(c->api->Send(g.send_lo, g.bytes, kNcclInt8, c->rank - 1, c->comm, s), "ncclSend");
    if (c->rank > 0 && g.recv_lo) rec(c->api->Recv(g.recv_lo, g.bytes, kNcclInt8, c->rank - 1, c->comm, s), "ncclRecv");
    if (c->rank < c->nranks - 1 && g.send_hi) rec(c->api->Send(g.send_hi, g.bytes, kNcclInt8, c->rank + 1, c->comm, s), "ncclSend");
    if (c->rank < c->nranks - 1 && g.recv_hi) rec(c->api->Recv(g.recv_hi, g.bytes, kNcclInt8, c->rank + 1, c->comm, s), "ncclRecv");
  }
  const int end = c->api->GroupEnd();
  if (first != 0) return fnx::set_error(FNX_ECOMM, "RCCL error %d (%s) in %s", first, c->api->GetErrorString ? c->api->GetErrorString(first) : "?", where);
  NCCL_OK(c, end);
  return FNX_OK;
}
void rccl_abort(void* vctx) {
  RcclCtx* c = (RcclCtx*)vctx;
  if (c->comm && c->api->CommAbort) { c->api->CommAbort(c->comm); c->comm = nullptr; }
}
int rccl_allreduce_max(void* vctx, float* x, int n, void* stream) {
  RcclCtx* c = (RcclCtx*)vctx;
  NCCL_OK(c, c->api->AllReduce(x, x, (size_t)n, kNcclFloat32, kNcclMax, c->comm, (hipStream_t)stream));
  return FNX_OK;
}
int rccl_allreduce_sum(void* vctx, float* x, int n, void* stream) {
  RcclCtx* c = (RcclCtx*)vctx;
  NCCL_OK(c, c->api->AllReduce(x, x, (size_t)n, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream));
  return FNX_OK;
}
void rccl_destroy(void* vctx) {
  RcclCtx* c = (RcclCtx*)vctx;
  if (c->comm) c->api->CommDestroy(c->comm);
  delete c;
}

// ---------------------------------------------------------------------------------------------------------------
// In-process communicator: rank r and r+1 meet in a mailbox per pair and exchange call.  The first to arrive leaves its
// pointers and an event recorded on its stream; the second enqueues both copies on ITS stream behind that event, records
// a completion event and wakes the first, whose stream then waits for it.  Every rank serves its lower pair before its
// upper pair, so the pairs are visited in ascending order everywhere and nobody waits in a cycle.
// ---------------------------------------------------------------------------------------------------------------
struct LoopPair {
  std::mutex mu;
  std::condition_variable cv;
  int phase = 0;                 // 0: empty, 1: first party posted, 2: copies enqueued (first party may pick up)
  std::vector<FnxSlabSeg> segs;  // the first party's segments
  bool first_is_lower = false;
  // one event per SIDE (0: the lower rank of the pair, 1: the upper one), created lazily by that side on ITS device and only
  // ever recorded on its own streams; the other side only waits on it (an event and the stream it is recorded on must belong
  // to the same device)
  hipEvent_t ev[2] = {nullptr, nullptr};
  int rc = FNX_OK;
};
struct LoopGroup {
  int nranks;
  std::vector<LoopPair> pairs;   // pair i: ranks i, i+1
  std::mutex mu; std::condition_variable cv;
  int red_count = 0, red_gen = 0; std::vector<float> red_val, red_out;   // (red_out: a finished round's result, safe from the next round's first arrival)
  int red_failed_gen = -1;       // the round that was abandoned (a rank timed out in it): every rank of that round fails together
  std::atomic<bool> aborted{false};   // a rank failed: every party waiting for a peer gives up (FNX_ECOMM) instead of hanging
  // a peer that never arrives (it failed before its exchange call) is an error, not a hang; fnx_slab_loopback_group_set_timeout
  std::atomic<int> timeout_ms{120 * 1000};
  explicit LoopGroup(int n) : nranks(n), pairs(n > 1 ? n - 1 : 0) {}
};
struct LoopCtx { LoopGroup* g; int rank; };

// waits on `cv` until pred() holds.  WAIT_ABORTED: the group was aborted (a rank failed; sticky until
// fnx_slab_loopback_group_reset).  WAIT_TIMEOUT: the peer did not show up in time -- this call fails, the group is NOT poisoned:
// a slow but healthy peer (first-call kernel load, a debugger) costs the caller one error it can retry after, not the group.
enum LoopWait { WAIT_OK = 0, WAIT_ABORTED, WAIT_TIMEOUT };
template <class Pred>
LoopWait loop_wait(LoopGroup* g, std::condition_variable& cv, std::unique_lock<std::mutex>& lk, Pred pred) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(g->timeout_ms.load());
  while (!pred()) {
    if (g->aborted.load()) return WAIT_ABORTED;
    if (cv.wait_for(lk, std::chrono::milliseconds(50)) == std::cv_status::timeout && std::chrono::steady_clock::now() > deadline)
      return pred() ? WAIT_OK : WAIT_TIMEOUT;
  }
  return WAIT_OK;
}
int loop_gone(LoopGroup* g, LoopWait why) {
  if (why == WAIT_TIMEOUT)
    return fnx::set_error(FNX_ECOMM, "loopback exchange: timed out after %d ms waiting for a peer rank (the group stays usable; "
                                     "fnx_slab_loopback_group_set_timeout changes the limit)", g->timeout_ms.load());
  return fnx::set_error(FNX_ECOMM, "loopback exchange: a peer rank failed (group aborted; fnx_slab_loopback_group_reset clears it)");
}

// one side of pair `pi`; lower = this rank is the lower one of the pair (it sends its *_hi pointers)
int loop_meet(LoopGroup* g, int pi, bool lower, const FnxSlabSeg* segs, int nsegs, hipStream_t s) {
  LoopPair& P = g->pairs[pi];
  const int me = lower ? 0 : 1, other = 1 - me;
  std::unique_lock<std::mutex> lk(P.mu);
  if (LoopWait w = loop_wait(g, P.cv, lk, [&] { return P.phase == 0 || (P.phase == 1 && P.first_is_lower != lower); })) return loop_gone(g, w);
  if (!P.ev[me]) SLAB_HIP(hipEventCreateWithFlags(&P.ev[me], hipEventDisableTiming));
  if (P.phase == 0) {                       // first to arrive
    P.segs.assign(segs, segs + nsegs);
    P.first_is_lower = lower;
    P.rc = FNX_OK;
    SLAB_HIP(hipEventRecord(P.ev[me], s));
    P.phase = 1;
    P.cv.notify_all();
    if (LoopWait w = loop_wait(g, P.cv, lk, [&] { return P.phase == 2; })) { P.phase = 0; P.cv.notify_all(); return loop_gone(g, w); }
    const int rc = P.rc;
    hipError_t e = rc == FNX_OK ? hipStreamWaitEvent(s, P.ev[other], 0) : hipSuccess;
    P.phase = 0;
    P.cv.notify_all();
    if (rc != FNX_OK) return rc;
    SLAB_HIP(e);
    return FNX_OK;
  }
  // second to arrive: both parties' buffers are known
  int rc = FNX_OK;
  if ((int)P.segs.size() != nsegs) rc = fnx::set_error(FNX_ECOMM, "loopback exchange: the two ranks of a pair posted %zu and %d segments", P.segs.size(), nsegs);
  if (rc == FNX_OK && hipStreamWaitEvent(s, P.ev[other], 0) != hipSuccess) rc = fnx::set_error(FNX_EHIP, "hipStreamWaitEvent failed");
  for (int i = 0; i < nsegs && rc == FNX_OK; ++i) {
    const FnxSlabSeg& lo = lower ? segs[i] : P.segs[i];     // the lower rank's segment: its hi side faces the pair
    const FnxSlabSeg& hi = lower ? P.segs[i] : segs[i];
    if (lo.bytes != hi.bytes) { rc = fnx::set_error(FNX_ECOMM, "loopback exchange: segment %d has %zu and %zu bytes", i, lo.bytes, hi.bytes); break; }
    if (hi.recv_lo && lo.send_hi && hipMemcpyAsync(hi.recv_lo, lo.send_hi, lo.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      rc = fnx::set_error(FNX_EHIP, "hipMemcpyAsync failed");
    if (lo.recv_hi && hi.send_lo && hipMemcpyAsync(lo.recv_hi, hi.send_lo, lo.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      rc = fnx::set_error(FNX_EHIP, "hipMemcpyAsync failed");
  }
  if (rc == FNX_OK && hipEventRecord(P.ev[me], s) != hipSuccess) rc = fnx::set_error(FNX_EHIP, "hipEventRecord failed");
  P.rc = rc;
  P.phase = 2;
  P.cv.notify_all();
  return rc;
}
int loop_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) {
  LoopCtx* c = (LoopCtx*)vctx;
  if (c->rank > 0) SLAB_OK(loop_meet(c->g, c->rank - 1, false, segs, nsegs, (hipStream_t)stream));
  if (c->rank < c->g->nranks - 1) SLAB_OK(loop_meet(c->g, c->rank, true, segs, nsegs, (hipStream_t)stream));
  return FNX_OK;
}
int loop_allreduce(void* vctx, float* x, int n, void* stream, bool sum) {
  LoopCtx* c = (LoopCtx*)vctx;
  LoopGroup* g = c->g;
  if (n < 1) return fnx::set_error(FNX_EINVAL, "loopback allreduce: n < 1");
  std::vector<float> hv((size_t)n);
  float* h = hv.data();
  SLAB_HIP(hipMemcpyAsync(h, x, n * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  SLAB_HIP(hipStreamSynchronize((hipStream_t)stream));
  {
    std::unique_lock<std::mutex> lk(g->mu);
    const int gen = g->red_gen;
    if (g->red_count == 0) g->red_val.assign(h, h + n);
    else for (int i = 0; i < n; ++i) g->red_val[i] = sum ? g->red_val[i] + h[i] : (h[i] > g->red_val[i] ? h[i] : g->red_val[i]);
    if (++g->red_count == g->nranks) { g->red_count = 0; g->red_out = g->red_val; ++g->red_gen; g->cv.notify_all(); }
    else if (LoopWait w = loop_wait(g, g->cv, lk, [&] { return g->red_gen != gen; })) {
      // This rank's value is already folded into red_val and cannot be taken out again: the unfinished round is abandoned as a
      // whole (a retry starts a fresh one; the ranks still waiting in this round fail with the same error), never resumed --
      // resuming would add a retrying rank twice.
      if (g->red_gen == gen) { g->red_count = 0; g->red_failed_gen = gen; ++g->red_gen; g->cv.notify_all(); }
      return loop_gone(g, w);
    }
    if (g->red_failed_gen == gen) return loop_gone(g, WAIT_TIMEOUT);
    for (int i = 0; i < n; ++i) h[i] = g->red_out[i];
  }
  SLAB_HIP(hipMemcpyAsync(x, h, n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
  SLAB_HIP(hipStreamSynchronize((hipStream_t)stream));
  return FNX_OK;
}
int loop_allreduce_max(void* vctx, float* x, int n, void* stream) { return loop_allreduce(vctx, x, n, stream, false); }
int loop_allreduce_sum(void* vctx, float* x, int n, void* stream) { return loop_allreduce(vctx, x, n, stream, true); }
void loop_destroy(void* vctx) { delete (LoopCtx*)vctx; }
void loop_abort(void* vctx) {
  LoopGroup* g = ((LoopCtx*)vctx)->g;
  g->aborted.store(true);
  for (LoopPair& p : g->pairs) p.cv.notify_all();
  g->cv.notify_all();
}

// ---------------------------------------------------------------------------------------------------------------
// Link-model communicator: ONE rank rehearses a middle slab on one GPU.  An exchange occupies its stream for
// latency + bytes-per-direction / bandwidth (a one-wave kernel that sleeps on the device's constant-rate clock) and then
// fills the ghost planes from the slab's own edge planes (the lower neighbour's top planes are taken to be this slab's top
// planes: a periodic stack of this slab -- the launch sequence, sizes and stream ordering are the real ones, the physics is
// not).  What it measures is how much of a transfer of an assumed link a schedule hides.
// ---------------------------------------------------------------------------------------------------------------
struct ModelCtx {
  double latency_us, gbps;
  // direct sends (direct_begin / direct_exchange): the producing kernel stores into these buffers; the exchange's launch moves them into
  // the ghost planes and lasts until latency + bytes / bandwidth have passed SINCE THE PRODUCING KERNEL STARTED (a clock
  // word the kernel itself writes when it starts): stores that leave as the march produces them overlap the kernel that issues them
  char* buf[2] = {nullptr, nullptr}; size_t cap = 0, stride = 0; unsigned long long* stamp = nullptr;
};
// One launch per exchange, like a real transport's: it moves the planes (ghost planes <- the slab's own edge planes) and does not end
// before latency + bytes per direction / bandwidth have passed on the device's constant-rate clock (wall_clock64: 100 MHz on gfx950).
// (Until round 4 the wait was a kernel of its own FOLLOWED by device-to-device copies: 15-25 us per exchange that no transport adds.)
constexpr int kModelSegs = 16;
struct ModelArgs { const char* src[2 * kModelSegs]; char* dst[2 * kModelSegs]; unsigned long long bytes[2 * kModelSegs]; int n; unsigned long long ticks;
                   const unsigned long long* since; unsigned long long tail_ticks; };
__global__ __launch_bounds__(256) void link_model_xfer_kernel(ModelArgs a) {
  // `since` (a direct send): the transfer is timed from the start of the PRODUCING launch -- its stores leave while it computes -- but
  // it cannot be through before that launch is (this kernel starts behind it in stream order) plus the link's latency: the ready
  // flag is raised after the last plane has been stored and nothing overlaps its flight (round 6; until then a producing launch longer
  // than latency + bytes / bandwidth made the exchange free)
  const unsigned long long now = wall_clock64();
  unsigned long long t0 = a.since ? *a.since : now;
  if (a.since && now + a.tail_ticks > t0 + a.ticks) { t0 = now; a.ticks = a.tail_ticks; }
  for (int i = 0; i < a.n; ++i) {
    const unsigned long long n16 = a.bytes[i] >> 4;
    if ((((unsigned long long)a.src[i] | (unsigned long long)a.dst[i] | a.bytes[i]) & 15ull) == 0) {
      const uint4* s = (const uint4*)a.src[i]; uint4* d = (uint4*)a.dst[i];
      for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < n16; q += (unsigned long long)gridDim.x * 256) d[q] = s[q];
    } else {
      for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < a.bytes[i]; q += (unsigned long long)gridDim.x * 256) a.dst[i][q] = a.src[i][q];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    while (wall_clock64() - t0 < a.ticks) __builtin_amdgcn_s_sleep(32);
}
int model_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) {
  ModelCtx* c = (ModelCtx*)vctx;
  hipStream_t s = (hipStream_t)stream;
  size_t bytes = 0;
  for (int i = 0; i < nsegs; ++i) bytes += segs[i].bytes;
  const double us = c->gbps > 0.0 ? c->latency_us + (double)bytes / (c->gbps * 1e3) : 0.0;
  for (int i0 = 0; i0 < nsegs; i0 += kModelSegs) {
    ModelArgs a{};
    for (int i = i0; i < nsegs && i < i0 + kModelSegs; ++i) {
      const FnxSlabSeg& g = segs[i];
      if (g.recv_lo && g.send_hi) { a.src[a.n] = (const char*)g.send_hi; a.dst[a.n] = (char*)g.recv_lo; a.bytes[a.n++] = g.bytes; }
      if (g.recv_hi && g.send_lo) { a.src[a.n] = (const char*)g.send_lo; a.dst[a.n] = (char*)g.recv_hi; a.bytes[a.n++] = g.bytes; }
    }
    a.ticks = i0 == 0 ? (unsigned long long)(us * 100.0) : 0ull;      // (the whole exchange's time rides on its first launch)
    link_model_xfer_kernel<<<64, 256, 0, s>>>(a);
  }
  SLAB_HIP(hipGetLastError());
  return FNX_OK;
}
int model_direct_begin(void* vctx, size_t bytes, int nsegs, void* dst[2][2], const unsigned* select[2], size_t* seg_stride, void** start_clock,
                       void* /*stream*/) {
  ModelCtx* c = (ModelCtx*)vctx;
  const size_t stride = (bytes + 255) & ~(size_t)255, need = stride * (size_t)nsegs;
  if (bytes == 0 || nsegs < 1 || nsegs > kModelSegs) return fnx::set_error(FNX_EINVAL, "link model direct send: bad segments");
  if (need > c->cap) {
    for (int d = 0; d < 2; ++d) { if (c->buf[d]) (void)hipFree(c->buf[d]); c->buf[d] = nullptr; }
    for (int d = 0; d < 2; ++d) SLAB_HIP(hipMalloc((void**)&c->buf[d], need));
    c->cap = need;
  }
  if (!c->stamp) SLAB_HIP(hipMalloc((void**)&c->stamp, sizeof(unsigned long long)));
  c->stride = stride;
  for (int d = 0; d < 2; ++d) { dst[d][0] = c->buf[d]; dst[d][1] = nullptr; select[d] = nullptr; }
  *seg_stride = stride;
  if (start_clock) *start_clock = c->stamp;                 // the producing kernel stamps its own start: no launch of ours in front of it
  return FNX_OK;
}
// what went "down" (buf[0]) arrives in this slab's ghost planes ABOVE (a periodic stack of this slab), and the other way round
int model_direct_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) {
  ModelCtx* c = (ModelCtx*)vctx;
  if (nsegs > kModelSegs || !c->buf[0]) return fnx::set_error(FNX_EINVAL, "link model direct exchange without direct_begin");
  size_t bytes = 0;
  ModelArgs a{};
  for (int i = 0; i < nsegs; ++i) {
    const FnxSlabSeg& g = segs[i];
    bytes += g.bytes;
    if (g.recv_hi) { a.src[a.n] = c->buf[0] + (size_t)i * c->stride; a.dst[a.n] = (char*)g.recv_hi; a.bytes[a.n++] = g.bytes; }
    if (g.recv_lo) { a.src[a.n] = c->buf[1] + (size_t)i * c->stride; a.dst[a.n] = (char*)g.recv_lo; a.bytes[a.n++] = g.bytes; }
  }
  const double us = c->gbps > 0.0 ? c->latency_us + (double)bytes / (c->gbps * 1e3) : 0.0;
  a.ticks = (unsigned long long)(us * 100.0);
  a.since = c->stamp;
  a.tail_ticks = c->gbps > 0.0 ? (unsigned long long)(c->latency_us * 100.0) : 0ull;
  link_model_xfer_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a);
  SLAB_HIP(hipGetLastError());
  return FNX_OK;
}
int model_allreduce(void*, float*, int, void*) { return FNX_OK; }        // one rank: its own value
void model_destroy(void* vctx) {
  ModelCtx* c = (ModelCtx*)vctx;
  for (int d = 0; d < 2; ++d) if (c->buf[d]) (void)hipFree(c->buf[d]);
  if (c->stamp) (void)hipFree(c->stamp);
  delete c;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
struct FnxSlab {
  FnxSlabConfig cfg;
  FnxSlabComm comm;          // copy of the caller's table (ctx borrowed); exchange == nullptr when nranks == 1
  int owned, lo, hi, z_offset, D_local, w;
  long steps = 0;
  bool mask_valid = false, cls_valid = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_post = nullptr, ev_done = nullptr;
  // deep_beside: the edge chain's stream (the exchanges of the sweep blocks ride on it too), its fork / join events and one
  // event per deep pass of a block
  hipStream_t edge_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_deep[FNX_SLAB_MAX_HALO] = {};
  bool pending = false;
  hipStream_t pending_on = nullptr;   // the pending exchange was enqueued on this stream itself (post_on), not on comm_stream
  float* h_cfl = nullptr;    // pinned host float
  // optional statistics (fnx_slab_stats_enable): bytes posted per neighbour and direction, exchanges, and how long the
  // compute stream stood at each wait -- an event pair around the stream wait, read back in fnx_slab_stats_read
  bool stats_on = false;
  FnxSlabStats stats{};
  std::vector<hipEvent_t> wait_ev;   // pairs: [2i] before, [2i+1] after the i-th wait since the last read
  size_t wait_used = 0;
};

namespace {

int layout_of(const FnxSlabConfig* c, int* owned, int* lo, int* hi, int* zoff) {
  if (!c) return fnx::set_error(FNX_EINVAL, "slab config is NULL");
  if (c->nranks < 1 || c->rank < 0 || c->rank >= c->nranks) return fnx::set_error(FNX_EINVAL, "slab: rank %d of %d", c->rank, c->nranks);
  if (c->B < 1 || c->H < 3 || c->W < 3 || c->D_global < 3) return fnx::set_error(FNX_EINVAL, "slab: bad domain %dx%dx%dx%d", c->B, c->D_global, c->H, c->W);
  if (c->D_global % c->nranks) return fnx::set_error(FNX_EINVAL, "slab: D must divide evenly across ranks");
  const int ow = c->D_global / c->nranks;
  if (c->nranks > 1 && c->halo < 5) return fnx::set_error(FNX_EINVAL, "slab: advection + projection need 5 valid ghost planes (CFL <= 1)");
  if (c->halo > FNX_SLAB_MAX_HALO) return fnx::set_error(FNX_EINVAL, "slab: halo %d > %d planes is not supported", c->halo, FNX_SLAB_MAX_HALO);
  if (c->schedule < FNX_SLAB_DEEP_FIRST || c->schedule > FNX_SLAB_DEEP_BESIDE) return fnx::set_error(FNX_EINVAL, "slab: unknown schedule %d", c->schedule);
  if (c->nranks > 1 && ow < c->halo) return fnx::set_error(FNX_EINVAL, "slab thinner than its halo");
  if (c->method != 0 && c->method != 1) return fnx::set_error(FNX_EINVAL, "slab: method %d (0: Jacobi, 1: CNN projection)", c->method);
  if (c->method == 1 && c->nranks > 1 && (c->halo < FNX_SLAB_NET_MARGIN + 1 || c->halo % 4 != 0 || ow % 4 != 0))
    return fnx::set_error(FNX_EINVAL, "slab: the CNN projection needs halo >= %d ghost planes, a multiple of 4, and owned planes a multiple of 4",
                          FNX_SLAB_NET_MARGIN + 1);
  *owned = ow;
  *lo = c->rank > 0 ? c->halo : 0;
  *hi = c->rank < c->nranks - 1 ? c->halo : 0;
  *zoff = c->rank * ow - *lo;
  return FNX_OK;
}

FnxGrid grid_of(const FnxSlab* s, int kb = 0, int ke = 0) {
  FnxGrid g{};
  g.B = s->cfg.B; g.D = s->D_local; g.H = s->cfg.H; g.W = s->cfg.W; g.is3D = 1; g.ref_quirks = 0;
  g.z_offset = s->z_offset; g.D_global = s->cfg.D_global; g.k_begin = kb; g.k_end = ke;
  return g;
}

// the planes [e0, e1) of the local array the CNN projection evaluates the net on: owned +- FNX_SLAB_NET_MARGIN, clipped at the domain ends
void net_range(const FnxSlab* s, int* e0, int* e1) {
  const int lo = s->lo, top = s->lo + s->owned;
  *e0 = s->cfg.rank > 0 ? lo - FNX_SLAB_NET_MARGIN : 0;
  *e1 = s->cfg.rank < s->cfg.nranks - 1 ? top + FNX_SLAB_NET_MARGIN : s->D_local;
}

struct Work {                      // the step's scratch, carved from the caller's workspace
  float *rho_adv, *U_adv, *div, *pbuf, *cfl;   // cfl: max(64, B) floats (CFL number / per-sample sums of squares)
  // CNN projection (cfg.method 1): net input on the local array / on the crop, the net's output on the crop, the gathered sums, scale
  float *x_local, *x_crop, *p_crop, *red, *scale;
  double* wpart;
  void* msws;
  double* part;                                // the residual's fixed-order partial sums (pTol > 0)
  unsigned char* cls;
  void *jac, *adv;
  size_t jac_bytes, adv_bytes;
};
size_t carve(const FnxSlab* s, void* ws, Work* w) {
  const FnxGrid g = grid_of(s);
  const size_t n1 = (size_t)g.B * g.D * g.H * g.W;
  char* base = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = base ? base + off : nullptr; off += al(bytes); return r; };
  Work t;
  t.rho_adv = (float*)take(n1 * 4); t.U_adv = (float*)take(n1 * 12); t.div = (float*)take(n1 * 4); t.pbuf = (float*)take(n1 * 4);
  t.cfl = (float*)take((size_t)(g.B > 64 ? g.B : 64) * 4); t.part = (double*)take(fnx::residual_scratch_bytes(g.B)); t.cls = (unsigned char*)take(n1);
  t.jac_bytes = fnx_workspace_bytes(&g, FNX_OP_JACOBI); t.jac = take(t.jac_bytes);
  t.adv_bytes = fnx_workspace_bytes(&g, FNX_OP_ADVECT_STEP); t.adv = take(t.adv_bytes);
  t.x_local = t.x_crop = t.p_crop = t.red = t.scale = nullptr; t.wpart = nullptr; t.msws = nullptr;
  if (s->cfg.method == 1) {
    int e0, e1;
    net_range(s, &e0, &e1);
    const size_t nc1 = (size_t)g.B * (e1 - e0) * g.H * g.W;
    t.x_local = (float*)take(n1 * 8); t.x_crop = (float*)take(nc1 * 8); t.p_crop = (float*)take(nc1 * 4);
    t.red = (float*)take((size_t)s->cfg.nranks * g.B * 6 * 4); t.scale = (float*)take((size_t)g.B * 4);
    t.wpart = (double*)take(fnx::window_sums_scratch_bytes(g.B));
    t.msws = take(fnx::multiscale_ws_bytes(make_dims(g.B, e1 - e0, g.H, g.W), true));
  }
  if (w) *w = t;
  return off;
}

// ghost exchange of `width` planes of `nf` fields (channels[i] channels each; B samples) with both neighbours
int build_segs(const FnxSlab* s, float* const* fields, float* const* sources, const int* channels, int nf, int width,
               std::vector<FnxSlabSeg>& segs) {
  const size_t plane = (size_t)s->cfg.H * s->cfg.W, vol = plane * s->D_local;
  const int lo = s->lo, top = s->lo + s->owned;
  segs.clear();
  for (int f = 0; f < nf; ++f)
    for (int b = 0; b < s->cfg.B; ++b)
      for (int c = 0; c < channels[f]; ++c) {
        float* dst = fields[f] + ((size_t)b * channels[f] + c) * vol;
        const float* src = (sources ? sources[f] : fields[f]) + ((size_t)b * channels[f] + c) * vol;
        FnxSlabSeg g{};
        g.bytes = (size_t)width * plane * 4;
        if (s->cfg.rank > 0) { g.send_lo = src + (size_t)lo * plane; g.recv_lo = dst + (size_t)(lo - width) * plane; }
        if (s->cfg.rank < s->cfg.nranks - 1) { g.send_hi = src + (size_t)(top - width) * plane; g.recv_hi = dst + (size_t)top * plane; }
        segs.push_back(g);
      }
  return FNX_OK;
}
// post on the communication stream behind everything `stream` has been given so far
int post(FnxSlab* s, float* const* fields, float* const* sources, const int* channels, int nf, int width, hipStream_t stream, bool direct = false) {
  if (s->cfg.nranks == 1) return FNX_OK;
  if (width > s->cfg.halo) return fnx::set_error(FNX_EINVAL, "slab: exchange wider than the halo");
  if (s->pending) return fnx::set_error(FNX_EINVAL, "slab: two exchanges in flight");
  std::vector<FnxSlabSeg> segs;
  build_segs(s, fields, sources, channels, nf, width, segs);
  SLAB_HIP(hipEventRecord(s->ev_post, stream));
  SLAB_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_post, 0));
  // direct: the send sides are already where the communicator wants them (the producing kernel mirrored them there)
  SLAB_OK((direct ? s->comm.direct_exchange : s->comm.exchange)(s->comm.ctx, segs.data(), (int)segs.size(), s->comm_stream));
  SLAB_HIP(hipEventRecord(s->ev_done, s->comm_stream));
  s->pending = true;
  s->pending_on = nullptr;
  if (s->stats_on) {
    size_t b = 0;
    for (const FnxSlabSeg& g : segs) b += g.bytes;
    s->stats.bytes_per_neighbour += (double)b;
    s->stats.exchanges += 1;
  }
  return FNX_OK;
}
// the same exchange enqueued on `on` itself (work enqueued on `on` afterwards is behind it without an event; any other stream
// waits for ev_done as for post)
int post_on(FnxSlab* s, float* const* fields, const int* channels, int nf, int width, hipStream_t on, bool direct = false) {
  if (s->cfg.nranks == 1) return FNX_OK;
  if (width > s->cfg.halo) return fnx::set_error(FNX_EINVAL, "slab: exchange wider than the halo");
  if (s->pending) return fnx::set_error(FNX_EINVAL, "slab: two exchanges in flight");
  std::vector<FnxSlabSeg> segs;
  build_segs(s, fields, nullptr, channels, nf, width, segs);
  SLAB_OK((direct ? s->comm.direct_exchange : s->comm.exchange)(s->comm.ctx, segs.data(), (int)segs.size(), on));
  SLAB_HIP(hipEventRecord(s->ev_done, on));
  s->pending = true;
  s->pending_on = on;
  if (s->stats_on) {
    size_t b = 0;
    for (const FnxSlabSeg& g : segs) b += g.bytes;
    s->stats.bytes_per_neighbour += (double)b;
    s->stats.exchanges += 1;
  }
  return FNX_OK;
}
int wait(FnxSlab* s, hipStream_t stream) {
  if (!s->pending) return FNX_OK;
  if (s->pending_on && s->pending_on == stream) {          // posted on this very stream (post_on): already ordered
    s->pending = false; s->pending_on = nullptr;
    return FNX_OK;
  }
  s->pending_on = nullptr;
  hipEvent_t *e0 = nullptr, *e1 = nullptr;
  if (s->stats_on) {
    if (s->wait_used + 2 > s->wait_ev.size()) {
      hipEvent_t a, b;
      SLAB_HIP(hipEventCreate(&a)); SLAB_HIP(hipEventCreate(&b));
      s->wait_ev.push_back(a); s->wait_ev.push_back(b);
    }
    e0 = &s->wait_ev[s->wait_used]; e1 = &s->wait_ev[s->wait_used + 1];
    s->wait_used += 2;
    SLAB_HIP(hipEventRecord(*e0, stream));
  }
  SLAB_HIP(hipStreamWaitEvent(stream, s->ev_done, 0));
  if (e1) SLAB_HIP(hipEventRecord(*e1, stream));
  s->pending = false;
  return FNX_OK;
}
int xchg(FnxSlab* s, float* const* fields, const int* channels, int nf, int width, hipStream_t stream) {
  SLAB_OK(post(s, fields, nullptr, channels, nf, width, stream));
  return wait(s, stream);
}

}  // namespace

extern "C" {

int fnx_slab_rccl_unique_id(void* out128) {
  if (!out128) return fnx::set_error(FNX_EINVAL, "unique id buffer is NULL");
  RcclApi* api;
  SLAB_OK(rccl_api(&api));
  NcclUniqueId id;
  const int r = api->GetUniqueId(&id);
  if (r != 0) return fnx::set_error(FNX_ECOMM, "ncclGetUniqueId failed (%d)", r);
  memcpy(out128, &id, sizeof(id));
  return FNX_OK;
}

int fnx_slab_comm_rccl(FnxSlabComm* out, int rank, int nranks, const void* unique_id128) {
  if (!out || !unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) return fnx::set_error(FNX_EINVAL, "slab_comm_rccl: bad arguments");
  RcclApi* api;
  SLAB_OK(rccl_api(&api));
  RcclCtx* c = new (std::nothrow) RcclCtx{api, nullptr, rank, nranks};
  if (!c) return fnx::set_error(FNX_EINVAL, "out of host memory");
  NcclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  const int r = api->CommInitRank(&c->comm, nranks, id, rank);
  if (r != 0) { delete c; return fnx::set_error(FNX_ECOMM, "ncclCommInitRank failed (%d: %s)", r, api->GetErrorString ? api->GetErrorString(r) : "?"); }
  out->ctx = c; out->exchange = rccl_exchange; out->allreduce_max = rccl_allreduce_max; out->allreduce_sum = rccl_allreduce_sum; out->destroy = rccl_destroy;
  out->abort = rccl_abort; out->direct_begin = nullptr; out->direct_exchange = nullptr;
  return FNX_OK;
}

int fnx_slab_comm_link_model(FnxSlabComm* out, double latency_us, double gbytes_per_s) {
  if (!out || latency_us < 0.0 || gbytes_per_s < 0.0) return fnx::set_error(FNX_EINVAL, "slab_comm_link_model: bad arguments");
  ModelCtx* mc = new ModelCtx();
  mc->latency_us = latency_us; mc->gbps = gbytes_per_s;
  out->ctx = mc;
  out->exchange = model_exchange; out->allreduce_max = model_allreduce; out->allreduce_sum = model_allreduce;
  out->destroy = model_destroy; out->abort = nullptr;
  out->direct_begin = model_direct_begin; out->direct_exchange = model_direct_exchange;
  return FNX_OK;
}

int fnx_slab_loopback_group(void** group, int nranks) {
  if (!group || nranks < 1) return fnx::set_error(FNX_EINVAL, "loopback group: bad arguments");
  *group = new (std::nothrow) LoopGroup(nranks);
  return *group ? FNX_OK : fnx::set_error(FNX_EINVAL, "out of host memory");
}
int fnx_slab_comm_loopback(FnxSlabComm* out, void* group, int rank) {
  LoopGroup* g = (LoopGroup*)group;
  if (!out || !g || rank < 0 || rank >= g->nranks) return fnx::set_error(FNX_EINVAL, "loopback comm: bad arguments");
  out->ctx = new LoopCtx{g, rank}; out->exchange = loop_exchange; out->allreduce_max = loop_allreduce_max; out->allreduce_sum = loop_allreduce_sum; out->destroy = loop_destroy;
  out->abort = loop_abort; out->direct_begin = nullptr; out->direct_exchange = nullptr;
  return FNX_OK;
}
int fnx_slab_loopback_group_set_timeout(void* group, double seconds) {
  LoopGroup* g = (LoopGroup*)group;
  if (!g || !(seconds > 0.0) || seconds > 86400.0) return fnx::set_error(FNX_EINVAL, "loopback group: timeout must be in (0, 86400] s");
  g->timeout_ms.store((int)(seconds * 1000.0 + 0.5) < 1 ? 1 : (int)(seconds * 1000.0 + 0.5));
  return FNX_OK;
}
int fnx_slab_loopback_group_reset(void* group) {
  LoopGroup* g = (LoopGroup*)group;
  if (!g) return fnx::set_error(FNX_EINVAL, "loopback group is NULL");
  // (the caller's promise: no rank is inside an exchange or all-reduce of this group)
  for (LoopPair& p : g->pairs) { std::lock_guard<std::mutex> lk(p.mu); p.phase = 0; p.rc = FNX_OK; p.segs.clear(); }
  { std::lock_guard<std::mutex> lk(g->mu); g->red_count = 0; ++g->red_gen; }
  g->aborted.store(false);
  return FNX_OK;
}
void fnx_slab_loopback_group_free(void* group) {
  LoopGroup* g = (LoopGroup*)group;
  if (!g) return;
  for (LoopPair& p : g->pairs) for (hipEvent_t e : p.ev) if (e) (void)hipEventDestroy(e);
  delete g;
}
void fnx_slab_comm_free(FnxSlabComm* comm) {
  if (comm && comm->destroy && comm->ctx) comm->destroy(comm->ctx);
  if (comm) { comm->ctx = nullptr; comm->exchange = nullptr; comm->allreduce_max = nullptr; comm->allreduce_sum = nullptr; comm->destroy = nullptr; comm->abort = nullptr;
              comm->direct_begin = nullptr; comm->direct_exchange = nullptr; }
}

int fnx_slab_layout(const FnxSlabConfig* cfg, int* owned, int* ghost_lo, int* ghost_hi, int* z_offset) {
  int o, l, h, z;
  SLAB_OK(layout_of(cfg, &o, &l, &h, &z));
  if (owned) *owned = o;
  if (ghost_lo) *ghost_lo = l;
  if (ghost_hi) *ghost_hi = h;
  if (z_offset) *z_offset = z;
  return FNX_OK;
}

size_t fnx_slab_workspace_bytes(const FnxSlabConfig* cfg) {
  FnxSlab t{};
  if (layout_of(cfg, &t.owned, &t.lo, &t.hi, &t.z_offset) != FNX_OK) return 0;
  t.cfg = *cfg; t.D_local = t.owned + t.lo + t.hi;
  return carve(&t, nullptr, nullptr);
}

int fnx_slab_create(FnxSlab** out, const FnxSlabConfig* cfg, const FnxSlabComm* comm) {
  if (!out) return fnx::set_error(FNX_EINVAL, "slab_create: out is NULL");
  FnxSlab* s = new (std::nothrow) FnxSlab();
  if (!s) return fnx::set_error(FNX_EINVAL, "out of host memory");
  int rc = layout_of(cfg, &s->owned, &s->lo, &s->hi, &s->z_offset);
  if (rc != FNX_OK) { delete s; return rc; }
  s->cfg = *cfg;
  s->D_local = s->owned + s->lo + s->hi;
  s->w = cfg->sweeps_per_exchange < cfg->halo ? cfg->sweeps_per_exchange : cfg->halo;
  if (s->w < 1) s->w = 1;
  if (cfg->nranks > 1) {
    if (!comm || !comm->exchange || !comm->allreduce_max || !comm->allreduce_sum) { delete s; return fnx::set_error(FNX_EINVAL, "slab_create: nranks > 1 needs a communicator"); }
    if (s->owned < 2 * s->w) { delete s; return fnx::set_error(FNX_EINVAL, "slab too thin for the sweep block"); }
    s->comm = *comm;
    if (hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_post, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming) != hipSuccess) {
      fnx_slab_destroy(s);
      return fnx::set_error(FNX_EHIP, "slab_create: stream / event creation failed");
    }
    if (cfg->schedule == FNX_SLAB_DEEP_BESIDE) {
      bool ok = hipStreamCreateWithFlags(&s->edge_stream, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) == hipSuccess;
      for (int i = 0; ok && i < (s->w + 1) / 2; ++i) ok = hipEventCreateWithFlags(&s->ev_deep[i], hipEventDisableTiming) == hipSuccess;
      if (!ok) { fnx_slab_destroy(s); return fnx::set_error(FNX_EHIP, "slab_create: stream / event creation failed"); }
    }
  } else {
    s->comm = FnxSlabComm{};
  }
  if (hipHostMalloc((void**)&s->h_cfl, (size_t)(cfg->B > 64 ? cfg->B : 64) * sizeof(float)) != hipSuccess) {     // CFL number / per-sample residuals
    fnx_slab_destroy(s);
    return fnx::set_error(FNX_EHIP, "slab_create: pinned host allocation failed");
  }
  *out = s;
  return FNX_OK;
}

void fnx_slab_destroy(FnxSlab* s) {
  if (!s) return;
  if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
  if (s->ev_post) (void)hipEventDestroy(s->ev_post);
  if (s->ev_done) (void)hipEventDestroy(s->ev_done);
  if (s->edge_stream) (void)hipStreamDestroy(s->edge_stream);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  for (hipEvent_t e : s->ev_deep) if (e) (void)hipEventDestroy(e);
  if (s->h_cfl) (void)hipHostFree(s->h_cfl);
  for (hipEvent_t e : s->wait_ev) (void)hipEventDestroy(e);
  delete s;
}

int fnx_slab_stats_enable(FnxSlab* s, int on) {
  if (!s) return fnx::set_error(FNX_EINVAL, "slab_stats_enable: NULL slab");
  s->stats_on = on != 0;
  s->stats = FnxSlabStats{};
  s->wait_used = 0;
  return FNX_OK;
}

int fnx_slab_stats_read(FnxSlab* s, FnxSlabStats* out) {
  if (!s || !out) return fnx::set_error(FNX_EINVAL, "slab_stats_read: NULL argument");
  double ms = 0.0;
  for (size_t i = 0; i + 1 < s->wait_used; i += 2) {
    float t = 0.f;
    SLAB_HIP(hipEventSynchronize(s->wait_ev[i + 1]));
    if (hipEventElapsedTime(&t, s->wait_ev[i], s->wait_ev[i + 1]) == hipSuccess) ms += t;
  }
  s->stats.wait_ms += ms;
  s->wait_used = 0;
  *out = s->stats;
  return FNX_OK;
}

int fnx_slab_comm_probe(const FnxSlabComm* comm, void* scratch, size_t bytes, int reps, float* ms_per_exchange, void* vstream) {
  if (!comm || !comm->exchange || !scratch || bytes == 0 || reps < 1 || !ms_per_exchange) return fnx::set_error(FNX_EINVAL, "slab_comm_probe: bad arguments");
  hipStream_t stream = (hipStream_t)vstream;
  char* b = (char*)scratch;
  FnxSlabSeg g{};
  g.send_lo = b; g.recv_lo = b + bytes; g.send_hi = b + 2 * bytes; g.recv_hi = b + 3 * bytes; g.bytes = bytes;
  hipEvent_t e0, e1;
  SLAB_HIP(hipEventCreate(&e0)); SLAB_HIP(hipEventCreate(&e1));
  int rc = comm->exchange(comm->ctx, &g, 1, stream);                      // warm-up (connection set-up)
  if (rc == FNX_OK && hipEventRecord(e0, stream) != hipSuccess) rc = fnx::set_error(FNX_EHIP, "hipEventRecord failed");
  for (int i = 0; i < reps && rc == FNX_OK; ++i) rc = comm->exchange(comm->ctx, &g, 1, stream);
  if (rc == FNX_OK && (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fnx::set_error(FNX_EHIP, "probe: event failed");
  float t = 0.f;
  if (rc == FNX_OK && hipEventElapsedTime(&t, e0, e1) != hipSuccess) rc = fnx::set_error(FNX_EHIP, "probe: elapsed time failed");
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc == FNX_OK) *ms_per_exchange = t / reps;
  return rc;
}

static int slab_step_body(FnxSlab* s, const FnxStepParams* prm, const FnxState* st, void* ws, size_t ws_bytes, void* vstream);

int fnx_slab_step(FnxSlab* s, const FnxStepParams* prm, const FnxState* st, void* ws, size_t ws_bytes, void* vstream) {
  const int rc = slab_step_body(s, prm, st, ws, ws_bytes, vstream);
  if (rc != FNX_OK && s && s->cfg.nranks > 1) {
    // A step that fails half-way may have forked work onto the internal streams (the edge chain of deep_beside, a posted
    // exchange) that nothing has joined yet: the caller's stream is made to wait for both, so that a graph capture is left with
    // no unjoined branch and neither the caller nor the next step can touch p / pbuf / the ghost planes while they are written.
    hipStream_t stream = (hipStream_t)vstream;
    if (s->edge_stream && s->ev_join && hipEventRecord(s->ev_join, s->edge_stream) == hipSuccess) (void)hipStreamWaitEvent(stream, s->ev_join, 0);
    if (s->comm_stream && s->ev_done && hipEventRecord(s->ev_done, s->comm_stream) == hipSuccess) (void)hipStreamWaitEvent(stream, s->ev_done, 0);
    (void)hipGetLastError();
    // a rank that fails must not leave its neighbours waiting in an exchange it will never join
    if (rc != FNX_ECFL && s->comm.abort) s->comm.abort(s->comm.ctx);
  }
  return rc;
}

static int slab_step_body(FnxSlab* s, const FnxStepParams* prm, const FnxState* st, void* ws, size_t ws_bytes, void* vstream) {
  if (!s || !prm || !st) return fnx::set_error(FNX_EINVAL, "slab_step: NULL argument");
  if (!st->p || !st->U || !st->flags || !st->density) return fnx::set_error(FNX_EINVAL, "slab_step: the z-slab driver needs p, U, flags and a density field");
  if (prm->method != 0 && prm->method != 1) return fnx::set_error(FNX_EINVAL, "slab_step: unknown method %d", prm->method);
  if (prm->method == 1 && (s->cfg.method != 1 || !st->net))
    return fnx::set_error(FNX_EINVAL, "slab_step: the CNN projection needs a driver created with cfg.method = 1 and st->net (the packed weights)");
  if (prm->method == 1 && (prm->precision_mode < FNX_PRECISION_FP32 || prm->precision_mode > FNX_PRECISION_BF16X3))
    return fnx::set_error(FNX_EINVAL, "slab_step: unknown precision_mode %d", prm->precision_mode);
  if (prm->method == 0 && prm->jacobi_iter < 1) return fnx::set_error(FNX_EINVAL, "At least 1 iteration of the solver is needed.");
  if (prm->viscosity != 0.f || prm->gravity_scale != 0.f || prm->correct_scalar || prm->periodic)
    return fnx::set_error(FNX_EINVAL, "slab_step: the optional stages (viscosity, gravityScale, correctScalar, periodic) are single-domain only");
  Work W;
  if (!ws || carve(s, ws, &W) > ws_bytes) return fnx::set_error(FNX_EWORKSPACE, "slab_step: workspace too small (%zu < %zu)", ws_bytes, carve(s, nullptr, nullptr));
  hipStream_t stream = (hipStream_t)vstream;
  s->pending = false; s->pending_on = nullptr;   // (a step that failed half-way must not block the next one)
  const int world = s->cfg.nranks, rank = s->cfg.rank, w = s->w;
  const int lo = s->lo, top = s->lo + s->owned, DL = s->D_local;
  const bool has_lo = rank > 0, has_hi = rank < world - 1;
  const float dt = prm->dt;
  const bool keep = s->cfg.static_flags && s->steps > 0;      // flags / BC arrays promised unchanged since the last step

  // CFL guard (control path): max |U| dt over all ranks
  if (s->cfg.cfl_check_every > 0 && s->steps % s->cfg.cfl_check_every == 0) {
    const FnxGrid g = grid_of(s);
    SLAB_OK(fnx_max_abs(&g, st->U, 3, W.cfl, stream));
    if (world > 1) SLAB_OK(s->comm.allreduce_max(s->comm.ctx, W.cfl, 1, stream));
    SLAB_HIP(hipMemcpyAsync(s->h_cfl, W.cfl, sizeof(float), hipMemcpyDeviceToHost, stream));
    SLAB_HIP(hipStreamSynchronize(stream));
    const float cfl = *s->h_cfl * (dt < 0 ? -dt : dt);
    if (cfl > 1.0f)
      return fnx::set_error(FNX_ECFL, "z-slab step: max |U| dt = %.3f cells > 1 -- the slab decomposition (ghost widths, advection "
                            "windows) is only valid for CFL <= 1; reduce dt or run the single-domain step", cfl);
  }
  if (!keep) { s->mask_valid = false; s->cls_valid = false; }
  ++s->steps;

  FnxState state = *st;
  state.net = nullptr;
  state.bc_class = nullptr;
  const bool has_bc = (st->UBC && st->UBCInvMask) || (st->densityBC && st->densityBCInvMask);
  if (keep && has_bc) {                    // class map of the BC arrays: built on the first step under the promise, then reused
    if (!s->cls_valid) { const FnxGrid g = grid_of(s); SLAB_OK(fnx_bc_classify(&g, st, W.cls, stream)); s->cls_valid = true; }
    state.bc_class = W.cls;
  }

  // ---- 1. advection: the U / density ghost exchange is in flight while the ghost-free planes are advected
  const int a_ = lo - 1 > 0 ? lo - 1 : 0, b_ = top + 1 < DL ? top + 1 : DL;
  const int ia_ = has_lo ? lo + 3 : a_, ib_ = has_hi ? top - 3 : b_;
  float* f2[2] = {st->U, st->density};
  const int c2[2] = {3, 1};
  const int gw = s->cfg.halo < 4 ? s->cfg.halo : 4;
  auto advect = [&](int kb, int ke) {
    const FnxGrid g = grid_of(s, kb, ke);
    return fnx_advect_step(&g, dt, st->density, st->U, st->flags, W.rho_adv, W.U_adv, prm->sample_outside_fluid,
                           prm->maccormack_strength, W.adv, W.adv_bytes, stream);
  };
  if (world > 1 && s->owned - 6 >= 8) {
    SLAB_OK(post(s, f2, nullptr, c2, 2, gw, stream));
    SLAB_OK(advect(ia_, ib_));
    SLAB_OK(wait(s, stream));
    if (ia_ > a_) SLAB_OK(advect(a_, ia_));
    if (b_ > ib_) SLAB_OK(advect(ib_, b_));
  } else {
    SLAB_OK(xchg(s, f2, c2, 2, gw, stream));
    if (world > 1) SLAB_OK(advect(a_, b_)); else SLAB_OK(advect(0, 0));
  }
  if (prm->method == 1) {
    // ---- the CNN projection (slab.py:_convnet_projection; lib/model.py:118-227, simulate.py:96-168 with 'convnet')
    const FnxGrid gw = world > 1 ? grid_of(s, lo, top) : grid_of(s);
    SLAB_OK(fnx_pre_projection(&gw, prm, &state, W.U_adv, W.rho_adv, nullptr, stream));     // setConstVals, addBuoyancy, setConstVals (owned planes)
    const GridDims dl = make_dims(s->cfg.B, DL, s->cfg.H, s->cfg.W, s->z_offset, s->cfg.D_global);
    // _ScaleNet: (sum, sumsq) of U over the owned planes -> every rank's pair on every rank -> std in rank order
    fnx::launch_window_sums_encode(dl, 3, lo, top, st->U, rank, world, W.wpart, W.red, stream);
    if (world > 1) {
      SLAB_OK(s->comm.allreduce_sum(s->comm.ctx, W.red, world * s->cfg.B * 6, stream));
      SLAB_HIP(hipStreamSynchronize(stream));                  // (the communicator's ordering contract: no exchange behind an all-reduce in flight)
    }
    fnx::launch_scale_from_sums(world, s->cfg.B, 3.0 * (double)s->cfg.D_global * s->cfg.H * s->cfg.W, prm->normalize_threshold, W.red, W.scale, stream);
    // the un-normalised velocity's ghost planes, once (the net's input is div(U / s) on owned +- 48 planes)
    float* fu[1] = {st->U};
    const int c3[1] = {3};
    SLAB_OK(xchg(s, fu, c3, 1, FNX_SLAB_NET_MARGIN + 1, stream));
    int e0, e1;
    net_range(s, &e0, &e1);
    const int De = e1 - e0;
    GridDims de = dl; de.K0 = e0; de.KN = De;
    fnx::launch_pack_div(de, true, st->U, st->flags, W.scale, W.x_local, stream);
    const size_t plane = (size_t)s->cfg.H * s->cfg.W, vol = plane * DL, volc = plane * De;
    for (int b = 0; b < s->cfg.B; ++b)
      for (int c = 0; c < 2; ++c)
        SLAB_HIP(hipMemcpyAsync(W.x_crop + ((size_t)b * 2 + c) * volc, W.x_local + ((size_t)b * 2 + c) * vol + (size_t)e0 * plane, volc * 4,
                                hipMemcpyDeviceToDevice, stream));
    // nested crops (slab.py: NET_MARGIN_FULL / _HALF): where the window ends at an artificial face the full- / half-resolution
    // towers stop 8 / 24 planes outside the owned ones; at a domain face nothing is trimmed
    const bool cut_lo = rank > 0 && lo - e0 == FNX_SLAB_NET_MARGIN, cut_hi = rank < world - 1 && e1 - top == FNX_SLAB_NET_MARGIN;
    const int trim[4] = { cut_lo ? FNX_SLAB_NET_MARGIN - FNX_SLAB_NET_MARGIN_FULL : 0, cut_hi ? FNX_SLAB_NET_MARGIN - FNX_SLAB_NET_MARGIN_FULL : 0,
                          cut_lo ? FNX_SLAB_NET_MARGIN - FNX_SLAB_NET_MARGIN_HALF : 0, cut_hi ? FNX_SLAB_NET_MARGIN - FNX_SLAB_NET_MARGIN_HALF : 0 };
    fnx::multiscale_forward_crop(make_dims(s->cfg.B, De, s->cfg.H, s->cfg.W), true, st->net, W.x_crop, W.p_crop, prm->precision_mode, W.msws,
                                 stream, trim);
    // the net's output back at its place in a local array
    float* pn = W.div;
    const int p0 = e0 + trim[0], pd = De - trim[0] - trim[1];
    for (int b = 0; b < s->cfg.B; ++b)
      SLAB_HIP(hipMemcpyAsync(pn + (size_t)b * vol + (size_t)p0 * plane, W.p_crop + (size_t)b * plane * pd, plane * pd * 4, hipMemcpyDeviceToDevice, stream));
    // ... exact on the owned planes; velocityUpdate also reads the plane below them: the lower neighbour's top plane
    if (world > 1) {
      float* fp[1] = {pn};
      const int c1p[1] = {1};
      SLAB_OK(xchg(s, fp, c1p, 1, 1, stream));
    }
    // velocityUpdate on U / s, un-normalise, setWallBcs, the step's last setConstVals: one pass over the owned planes
    GridDims dw = dl;
    if (world > 1) { dw.K0 = lo; dw.KN = s->owned; }
    const bool ubc = st->UBC && st->UBCInvMask, rbc = st->densityBC && st->densityBCInvMask;
    fnx::launch_post_projection(dw, true, pn, st->U, st->density, st->flags, ubc ? st->UBC : nullptr, ubc ? st->UBCInvMask : nullptr,
                                rbc ? st->densityBC : nullptr, rbc ? st->densityBCInvMask : nullptr, stream, state.bc_class, true, W.scale, st->p);
    SLAB_HIP(hipGetLastError());
    return FNX_OK;
  }
  // ---- 2. BC / buoyancy / wall stage + divergence on the owned planes
  {
    const FnxGrid g = world > 1 ? grid_of(s, lo, top) : grid_of(s);
    SLAB_OK(fnx_pre_projection(&g, prm, &state, W.U_adv, W.rho_adv, W.div, stream));
  }
  float* fd[1] = {W.div};
  const int c1[1] = {1};
  // sweep blocks with an exchange between them (else: thin slabs, short solves, pTol: the "last_pass" code below)
  const bool blocked = world > 1 && s->owned >= 4 * w && prm->jacobi_iter > w && !(prm->p_tol > 0.f);
  const bool beside = blocked && s->cfg.schedule == FNX_SLAB_DEEP_BESIDE;
  const bool deep = blocked && (s->cfg.schedule == FNX_SLAB_DEEP_FIRST || beside);
  // deep_first: the deep parts of the first sweep block read no ghost plane of div, its exchange is in flight behind them
  if (deep) SLAB_OK(post(s, fd, nullptr, c1, 1, w - 1 > 1 ? w - 1 : 1, stream));
  else SLAB_OK(xchg(s, fd, c1, 1, w - 1 > 1 ? w - 1 : 1, stream));

  // ---- 3. Jacobi: blocks of w sweeps between ghost exchanges of p
  const FnxGrid gj = grid_of(s);
  // lay: the row-quad layout of fnx_jacobi_pass_layout (bit 0: pin, bit 1: pout)
  auto pass = [&](const float* pin, float* pout, int n, int kb, int ke, int kb2 = -1, int lay = 0, hipStream_t on = nullptr) {
    const int rc = fnx_jacobi_pass_layout(&gj, st->flags, W.div, pin, pout, n, kb, ke, kb2, pin ? lay : (lay & 2), W.jac, W.jac_bytes,
                                          s->mask_valid ? 1 : 0, on ? on : stream);
    if (rc == FNX_OK) s->mask_valid = true;     // (a failed call may not have built the mask)
    return rc;
  };
  // The LAST edge part of a sweep block, with the w planes each neighbour needs next mirrored into the communicator's direct-send
  // windows (FnxSlabComm.direct_begin: the peer-store communicator's mailbox slots on the neighbours): the march stores them there
  // as it finishes them and the exchange that follows has nothing to copy on the sending side -- the transport's launch and this
  // part's own run time leave the chain exchange -> edge chain -> exchange.  *sent: the exchange must be posted as direct.  Falls back
  // to the plain part (communicator without direct sends, launch shape, single-sweep last part).
  auto edge_last = [&](const float* pin, float* pout, int n, int done, int sp, int lay, hipStream_t on, bool* sent) {
    *sent = false;
    hipStream_t q = on ? on : stream;
    const int len = w - done + sp;                            // planes of one face's part: [face - w + done, face + sp)
    const bool both = has_lo && has_hi;
    const bool wanted = s->cfg.direct_sends == 2 || (s->cfg.direct_sends == 0 && s->cfg.schedule == FNX_SLAB_DEEP_BESIDE);
    const bool can = wanted && s->comm.direct_begin && s->comm.direct_exchange && n == 2 && pin && (lay == 0 || lay == 3) &&
                     s->mask_valid && fnx_jacobi_pass_mirror_ok(&gj, len, both ? 1 : 0, lay) != 0;
    void* dst[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    const unsigned* sel[2] = {nullptr, nullptr};
    size_t stride = 0;
    const size_t plane = (size_t)s->cfg.H * s->cfg.W;
    void* clk = nullptr;
    if (can && s->comm.direct_begin(s->comm.ctx, (size_t)w * plane * 4, s->cfg.B, dst, sel, &stride, &clk, q) == FNX_OK) {
      FnxPlaneMirror m{};
      m.planes = w; m.sample_stride = stride / 4; m.start_clock = (unsigned long long*)clk;
      int r = 0;
      if (has_lo) { m.out[r][0] = (float*)dst[0][0]; m.out[r][1] = (float*)dst[0][1]; m.slot_select[r] = sel[0]; m.k_first[r] = lo; ++r; }
      if (has_hi) { m.out[r][0] = (float*)dst[1][0]; m.out[r][1] = (float*)dst[1][1]; m.slot_select[r] = sel[1]; m.k_first[r] = top - w; ++r; }
      const int kb = has_lo ? lo - w + done : top - sp;
      SLAB_OK(fnx_jacobi_pass_mirror(&gj, st->flags, W.div, pin, pout, kb, kb + len, both ? top - sp : -1, lay, &m, W.jac, W.jac_bytes, 1, q));
      *sent = true;
      return (int)FNX_OK;
    }
    if (both) return pass(pin, pout, n, lo - w + done, lo + sp, top - sp, lay, on);
    if (has_lo) SLAB_OK(pass(pin, pout, n, lo - w + done, lo + sp, -1, lay, on));
    if (has_hi) SLAB_OK(pass(pin, pout, n, top - sp, top + w - done, -1, lay, on));
    return (int)FNX_OK;
  };
  float *cur = st->p, *nxt = W.pbuf;
  int remaining = prm->jacobi_iter;
  bool zero_in = true;                       // the solve starts from p = 0 everywhere: the first pass reads nothing
  // passes of a sweep block (an odd block runs its single sweep first), and whether they hand each other the pressure in the
  // solver's row-quad layout (both arrays, every plane range, the ghost planes the neighbours send -- they run the same
  // schedule); the last pass of the solve writes rows
  int npass = 0, passes[FNX_SLAB_MAX_HALO];
  if (w % 2) passes[npass++] = 1;
  for (int i = 0; i < w / 2; ++i) passes[npass++] = 2;
  const bool quad = w % 2 == 0 && prm->jacobi_iter % 2 == 0 && fnx_jacobi_quad_ok(&gj) != 0;
  const int Q = quad ? 3 : 0;
  // the last block of a blocked solve (<= w sweeps, no exchange after it): whole shrinking ranges
  auto last_block = [&]() {
    int done = 0;
    for (int left = remaining; left > 0;) {
      const int n = left >= 2 ? 2 : 1;
      left -= n; done += n;
      const int g = w - done > 0 ? w - done : 0;
      SLAB_OK(pass(cur, nxt, n, has_lo ? lo - g : 0, has_hi ? top + g : DL, -1, left > 0 ? Q : (Q & 1)));
      float* t = cur; cur = nxt; nxt = t;
    }
    return (int)FNX_OK;
  };
  if (prm->p_tol > 0.f) {
    // the reference's convergence test (fluids_init.cpp:961-979; slab.py:_jacobi_ptol): one sweep per ghost exchange, the
    // squared differences over the OWNED planes summed over the ranks, max over the samples of the root against pTol
    const size_t plane = (size_t)s->cfg.H * s->cfg.W, vol = plane * DL;
    SLAB_HIP(hipMemsetAsync(cur, 0, (size_t)s->cfg.B * vol * 4, stream));
    const int a = has_lo ? lo : 0, b = has_hi ? top : DL;
    for (int it = 0; it < prm->jacobi_iter; ++it) {
      float* fc[1] = {cur};
      if (it > 0) SLAB_OK(xchg(s, fc, c1, 1, 1, stream));
      SLAB_OK(pass(cur, nxt, 1, a, b));
      // per-sample sums of squares over the owned planes, in a fixed order (reproducible: fnx_residual)
      fnx::launch_residual(s->cfg.B, vol, (size_t)lo * plane, (size_t)s->owned * plane, nxt, cur, W.part, W.cfl, nullptr, stream);
      if (world > 1) SLAB_OK(s->comm.allreduce_sum(s->comm.ctx, W.cfl, s->cfg.B, stream));
      SLAB_HIP(hipMemcpyAsync(s->h_cfl, W.cfl, s->cfg.B * sizeof(float), hipMemcpyDeviceToHost, stream));
      SLAB_HIP(hipStreamSynchronize(stream));
      float worst = 0.f;
      for (int bb = 0; bb < s->cfg.B; ++bb) { const float r = sqrtf(s->h_cfl[bb]); if (r > worst) worst = r; }
      float* t = cur; cur = nxt; nxt = t;
      if (worst < prm->p_tol) break;
    }
  } else if (world == 1) {
    // a single rank has nobody to exchange with: the whole solve in one call (its passes hand each other p in the
    // solver's row-quad layout, which the plane-range passes below do not)
    SLAB_OK(fnx_jacobi_sweeps_ex(&gj, st->flags, W.div, cur, prm->jacobi_iter, W.jac, W.jac_bytes, (s->mask_valid ? 1 : 0) | 2, stream));
    s->mask_valid = true;
  } else if (beside) {
    // "deep_beside" (slab.py:_jacobi_deep_beside): the plane ranges of "deep_first", the edge chain of a block issued on its
    // own stream beside the deep chain: E_k waits for D_(k-1) only (what it reads just inside its split, and the array it
    // overwrites below it); the edge stream also waits for the previous exchange and posts the next one; the next block's
    // D_0 waits for the edge chain (join).  A block takes max(deep chain, exchange + edge chain) instead of their sum.
    int split[FNX_SLAB_MAX_HALO];
    for (int k = 0, sp = 0; k < npass; ++k) {
      sp = k == 0 ? passes[k] : (sp + passes[k] > w ? sp + passes[k] : w);
      split[k] = sp;
    }
    if (s->owned < 2 * split[npass - 1] + 1) return fnx::set_error(FNX_EINVAL, "slab too thin for the deep_first sweep block");
    hipStream_t es = s->edge_stream;
    bool first_launch = true;
    while (remaining > w) {
      remaining -= w;
      SLAB_HIP(hipEventRecord(s->ev_fork, stream));        // fork: behind the previous block's join (first block: the staging pass)
      SLAB_HIP(hipStreamWaitEvent(es, s->ev_fork, 0));
      SLAB_OK(wait(s, es));                                // the ghost planes of `cur` (first block: of div): the EDGE stream waits
      float *src = cur, *dst = nxt;
      int done = 0;
      bool sent = false;
      for (int pi = 0; pi < npass; ++pi) {
        const int n = passes[pi];
        done += n;
        const float* pin = (zero_in && pi == 0) ? nullptr : src;
        SLAB_OK(pass(pin, dst, n, has_lo ? lo + split[pi] : 0, has_hi ? top - split[pi] : DL, -1, Q));
        SLAB_HIP(hipEventRecord(s->ev_deep[pi], stream));
        // E_pi behind D_(pi-1); the very first launch of a step may have built the solver's obstacle mask: E_0 behind it.
        // (One wait for the whole deep chain instead -- an event wait costs an in-order stream ~6 us even when the event has long
        // fired -- was measured: no better with a link, 4.69 against 4.15 ms per step without transfer time.)
        if (pi > 0 || first_launch) SLAB_HIP(hipStreamWaitEvent(es, s->ev_deep[pi > 0 ? pi - 1 : 0], 0));
        first_launch = false;
        if (pi == npass - 1) SLAB_OK(edge_last(pin, dst, n, done, split[pi], Q, es, &sent));
        else if (has_lo && has_hi) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + split[pi], top - split[pi], Q, es));
        else {
          if (has_lo) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + split[pi], -1, Q, es));
          if (has_hi) SLAB_OK(pass(pin, dst, n, top - split[pi], top + w - done, -1, Q, es));
        }
        float* t = src; src = dst; dst = t;
      }
      if (src != cur) { float* t = cur; cur = nxt; nxt = t; }
      float* ff[1] = {cur};
      SLAB_HIP(hipEventRecord(s->ev_join, es));            // join: the next block's D_0 reads what the last edge part wrote
      SLAB_HIP(hipStreamWaitEvent(stream, s->ev_join, 0));
      // The exchange rides on the EDGE stream itself, behind the edge chain and ahead of the next block's: the serial chain
      // exchange -> edge chain -> exchange that bounds a block with a real link is then one in-order stream -- no stream hand-over
      // (an event record + wait costs ~10 us on this runtime) anywhere on it.  The next block's deep chain was released by the
      // join above and runs beside the transfer: it reads owned planes only, the transfer writes ghost planes.
      SLAB_OK(post_on(s, ff, c1, 1, w, es, sent));
      zero_in = false;
    }
    SLAB_OK(wait(s, stream));
    SLAB_OK(last_block());
  } else if (deep) {
    // "deep_first" (slab.py:_jacobi_deep_first): pass k of a block is cut split[k] planes inside each internal face; the deep
    // parts of all passes run first (no ghost plane read: the previous exchange is still in flight), then the wait, then
    // the edge parts (a chain of short launches), then the w owned planes next to each face go to the neighbours
    int split[FNX_SLAB_MAX_HALO];
    for (int k = 0, sp = 0; k < npass; ++k) {
      sp = k == 0 ? passes[k] : (sp + passes[k] > w ? sp + passes[k] : w);
      split[k] = sp;
    }
    if (s->owned < 2 * split[npass - 1] + 1) return fnx::set_error(FNX_EINVAL, "slab too thin for the deep_first sweep block");
    while (remaining > w) {
      remaining -= w;
      float *src = cur, *dst = nxt;
      for (int pi = 0; pi < npass; ++pi) {
        SLAB_OK(pass((zero_in && pi == 0) ? nullptr : src, dst, passes[pi], has_lo ? lo + split[pi] : 0, has_hi ? top - split[pi] : DL, -1, Q));
        float* t = src; src = dst; dst = t;
      }
      SLAB_OK(wait(s, stream));              // the ghost planes of `cur` (first block: of div)
      src = cur; dst = nxt;
      int done = 0;
      bool sent = false;
      for (int pi = 0; pi < npass; ++pi) {
        const int n = passes[pi];
        done += n;
        const float* pin = (zero_in && pi == 0) ? nullptr : src;
        if (pi == npass - 1) SLAB_OK(edge_last(pin, dst, n, done, split[pi], Q, nullptr, &sent));
        else if (has_lo && has_hi) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + split[pi], top - split[pi], Q));
        else {
          if (has_lo) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + split[pi], -1, Q));
          if (has_hi) SLAB_OK(pass(pin, dst, n, top - split[pi], top + w - done, -1, Q));
        }
        float* t = src; src = dst; dst = t;
      }
      if (src != cur) { float* t = cur; cur = nxt; nxt = t; }
      float* ff[1] = {cur};
      SLAB_OK(post(s, ff, nullptr, c1, 1, w, stream, sent));
      zero_in = false;
    }
    SLAB_OK(wait(s, stream));
    SLAB_OK(last_block());
  } else if (blocked && s->cfg.schedule == FNX_SLAB_EDGE_FIRST) {
    // "edge_first" (slab.py:_jacobi_edge_first)
    int block = 0;
    while (remaining > w) {
      remaining -= w;
      if (++block > 1) SLAB_OK(wait(s, stream));
      float *src = cur, *dst = nxt;
      int done = 0;
      for (int pi = 0; pi < npass; ++pi) {
        const int n = passes[pi];
        done += n;
        const float* pin = (zero_in && pi == 0) ? nullptr : src;
        if (has_lo && has_hi) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + 2 * w - done, top - 2 * w + done, Q));
        else {
          if (has_lo) SLAB_OK(pass(pin, dst, n, lo - w + done, lo + 2 * w - done, -1, Q));
          if (has_hi) SLAB_OK(pass(pin, dst, n, top - 2 * w + done, top + w - done, -1, Q));
        }
        float* t = src; src = dst; dst = t;
      }
      float* fin = src;
      float* ff[1] = {fin};
      SLAB_OK(post(s, ff, nullptr, c1, 1, w, stream));
      src = cur; dst = nxt; done = 0;
      for (int pi = 0; pi < npass; ++pi) {
        const int n = passes[pi];
        done += n;
        SLAB_OK(pass((zero_in && pi == 0) ? nullptr : src, dst, n, has_lo ? lo + 2 * w - done : 0, has_hi ? top - 2 * w + done : DL, -1, Q));
        float* t = src; src = dst; dst = t;
      }
      if (fin != cur) { float* t = cur; cur = nxt; nxt = t; }
      zero_in = false;
    }
    SLAB_OK(wait(s, stream));
    SLAB_OK(last_block());
  } else {
    // "last_pass" (slab.py:_jacobi_last_pass): thin slabs, short solves
    bool pending = false, first = true;
    while (remaining > 0) {
      const int k = w < remaining ? w : remaining;
      remaining -= k;
      if (pending) { SLAB_OK(wait(s, stream)); pending = false; }
      int done = 0;
      for (int left = k; left > 0;) {
        const int n = left >= 2 ? 2 : 1;
        left -= n; done += n;
        const float* pin = first ? nullptr : cur;
        first = false;
        const int g = w - done > 0 ? w - done : 0;
        const int a = has_lo ? lo - g : 0, b = has_hi ? top + g : DL;
        if (left == 0 && remaining > 0 && world > 1) {
          const int ia = has_lo ? lo + w : a, ib = has_hi ? top - w : b;
          if (has_lo) SLAB_OK(pass(pin, nxt, n, lo, lo + w));
          if (has_hi) SLAB_OK(pass(pin, nxt, n, top - w, top));
          float* ff[1] = {nxt};
          SLAB_OK(post(s, ff, nullptr, c1, 1, w, stream));
          pending = true;
          if (ib > ia) SLAB_OK(pass(pin, nxt, n, ia, ib));
        } else if (world > 1) {
          SLAB_OK(pass(pin, nxt, n, a, b));
        } else {
          SLAB_OK(pass(pin, nxt, n, 0, 0));
        }
        float* t = cur; cur = nxt; nxt = t;
      }
    }
    if (pending) SLAB_OK(wait(s, stream));
  }
  if (cur != st->p) {
    const FnxGrid g = grid_of(s);
    SLAB_HIP(hipMemcpyAsync(st->p, cur, (size_t)g.B * g.D * g.H * g.W * 4, hipMemcpyDeviceToDevice, stream));
  }
  // ---- 4. velocity update on the owned planes
  float* fp[1] = {st->p};
  SLAB_OK(xchg(s, fp, c1, 1, 1, stream));
  {
    const FnxGrid g = world > 1 ? grid_of(s, lo, top) : grid_of(s);
    state.density_bc_applied = 1;            // the staging pass of this step, on the same planes, with the same BC arrays
    SLAB_OK(fnx_post_projection(&g, &state, stream));
  }
  return FNX_OK;
}

}  // extern "C"
