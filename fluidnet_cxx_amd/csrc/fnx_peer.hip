// Peer-store communicator of the z-slab driver (include/fluidnet_hip.h, fnx_slab_peer_create / fnx_slab_comm_peer): ghost planes
// travel as DEVICE STORES into a mailbox the neighbour has mapped (hipIpc handles between processes, the pointer itself inside one
// process), ordered by flags that live in the same mapping.  No RCCL kernel, no proxy thread, no host round trip on the data path:
// one short launch per exchange on the stream the driver names, whose first workgroups push this rank's planes over xGMI and whose
// last workgroups wait (device side) for the neighbours' planes and put them in place.
//
//   region of a rank (one uncached device allocation, mapped by both neighbours):
//     header   flags + the small all-reduce slots                      written remotely, polled locally
//     mailbox  [side: 0 filled by the lower neighbour, 1 by the upper one][slot: 2 chunks in flight][slot_bytes]
//
//   chunk n towards side s (n = 1, 2, ... per direction; both ends count alike, the step is deterministic; the counters live in the
//   header, on the device: a launch reads n = seq + 1 and its last workgroup stores it back, so that no launch argument depends on
//   how many exchanges came before and a step with its exchanges can be replayed as a HIP graph), cut into sub-chunks
//   of >= 64 KiB that one workgroup moves on its own:
//     push  waits for credit[s] >= n - 2 in ITS header (the neighbour has emptied the slot); per sub-chunk j: stores the bytes into the
//           neighbour's mailbox[1 - s][n & 1], fences, and sets ready[1 - s][n & 1][j] = n in the neighbour's header
//     pull  per sub-chunk j: waits for ready[s][n & 1][j] >= n in ITS header and copies the bytes from mailbox[s][n & 1] to where the
//           driver wants the planes; when the whole chunk is out, fences and sets credit[1 - s] = n in the neighbour's header
//
// DIRECT sends (direct_begin / direct_exchange): the kernel that produces the planes stores them into the neighbour's slot itself (the
// z-slab driver's last edge part of a sweep block, fnx_jacobi_pass_mirror; the slot's turn read from the push counter) and the
// exchange's push workgroups only raise the ready words.
//
// A wait that lasts longer than the timeout (a dead peer) or sees the abort word gives up, raises the rank's error word (pinned host
// memory: the next call of the communicator fails with FNX_ECOMM without a synchronisation) AND the rank's own abort word, and leaves
// the planes alone.  Every exchange launch looks at the abort word when it starts: the launches already queued behind the failed one --
// the rest of the step, the rest of a replayed graph -- return at once, so a dead neighbour costs ONE time-out, not one per exchange (a
// spin must never outlive its peer on a GPU other processes share).  A step / replay that ran into this still returns FNX_OK (nothing on
// the host waited): fnx_slab_peer_failed() after the caller's synchronisation says whether its ghost planes can be trusted.
#include <hip/hip_runtime.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <limits>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fluidnet_hip.h"
#include "fnx_kernels.h"

namespace {

#define PEER_HIP(expr)                                                                                     \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return fnx::set_error(FNX_EHIP, "HIP error: %s (%s)", hipGetErrorString(e_), #expr); \
  } while (0)

constexpr int kRedMax = 1024;            // floats per control-path all-reduce
constexpr int kMaxPieces = 16;           // plane blocks per launch and direction
constexpr int kBlocksPerRole = 32;       // workgroups per (push | pull) x (lower | upper)
constexpr int kMaxSub = 1024;            // sub-chunks of a mailbox slot: the unit a push workgroup hands to a pull workgroup
constexpr size_t kSubMin = 65536;
constexpr size_t kHeaderBytes = 65536;

struct PeerHeader {                      // all words written with system-scope atomics
  unsigned reserved0[2];
  unsigned credit[2];                    // [side I push to]: chunks that neighbour has taken out of the mailbox I fill
  unsigned abort_word;                   // != 0: every wait gives up
  unsigned done_count[4];                // workgroups of a role that have finished (this rank's own kernels only)
  // Chunk counters per role (push lower / upper, pull lower / upper), kept ON THE DEVICE: a launch reads its chunk number n = seq + 1
  // when it starts and its last workgroup stores it back -- no launch argument depends on how many exchanges came before, so a step
  // with its exchanges can be captured in a HIP graph and replayed (both ends count alike: the step is deterministic).
  unsigned seq[4];
  unsigned pad[3];
  unsigned red_up_seq, red_dn_seq;       // control-path all-reduce: partial result from rank - 1, total from rank + 1
  float red_up[kRedMax], red_dn[kRedMax];
  // ready[side the data came from][slot][sub-chunk] = number of the chunk whose bytes of that sub-chunk have arrived: a push
  // workgroup owns whole sub-chunks and raises each one's word on its own, so the pull of a chunk runs behind its push sub-chunk by
  // sub-chunk (a 6 MiB message is on its way for 80 us at 75 GB/s: moving it into place must not come on top of that)
  unsigned ready[2][2][kMaxSub];
};
static_assert(sizeof(PeerHeader) <= kHeaderBytes, "header");

struct PeerBlob {                        // what a rank publishes (FNX_PEER_HANDLE_BYTES)
  unsigned magic;
  int rank, nranks, pid;
  unsigned long long slot_bytes;
  unsigned long long raw;                // the region's address in the owner's process
  unsigned long long token;              // drawn per process: a pid alone is not an address space (two containers can share one)
  int device;                            // the owner's HIP device
  hipIpcMemHandle_t ipc;
};
// one token per process (address space): pid, a clock and an address of this library as it is mapped here
unsigned long long process_token() {
  static const unsigned long long t = [] {
    unsigned long long x = (unsigned long long)getpid() * 0x9E3779B97F4A7C15ull;
    x ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0xBF58476D1CE4E5B9ull;
    x ^= (unsigned long long)(uintptr_t)&t;
    return x | 1ull;
  }();
  return t;
}
static_assert(sizeof(PeerBlob) <= FNX_PEER_HANDLE_BYTES, "handle blob");
constexpr unsigned kMagic = 0x464e5850u;

struct Peer {
  int rank = 0, nranks = 1, device = 0;
  size_t slot_bytes = 0, region_bytes = 0;
  char* region = nullptr;                // my region
  char* nb[2] = {nullptr, nullptr};      // the neighbours' regions as mapped here (0: lower, 1: upper)
  bool nb_ipc[2] = {false, false};       // opened with hipIpcOpenMemHandle (to be closed)
  unsigned red_seq = 0;
  size_t sub_bytes = kSubMin;            // (a multiple of 256: pieces start on 256-byte boundaries of a slot)
  int* h_err = nullptr;                  // pinned host word the kernels raise
  double timeout_s = 30.0;
  PeerHeader* hdr() const { return (PeerHeader*)region; }
  char* mailbox(char* base, int side, int slot) const { return base + kHeaderBytes + ((size_t)side * 2 + slot) * slot_bytes; }
};

struct Piece { const char* src; char* dst; unsigned long long bytes, off; };       // off: offset in the mailbox slot
struct XferArgs {
  Piece push[2][kMaxPieces]; int npush[2];       // [side pushed to]
  Piece pull[2][kMaxPieces]; int npull[2];       // [side pulled from]: src is filled in by the kernel (the local mailbox)
  char* remote_slot[2][2];                       // [side][chunk parity]: the neighbour's mailbox slots this rank fills
  const char* local_slot[2][2];                  // my mailbox slots this rank empties
  unsigned* remote_ready[2][2];                  // the neighbour's ready[1 - s][slot]
  const unsigned* local_ready[2][2];             // my ready[s][slot]
  unsigned* remote_credit[2];                    // the neighbour's credit[1 - s]
  int active[2];                                 // a neighbour on that side
  unsigned long long used[2];                    // bytes of the slot this launch fills / empties per direction
  unsigned long long sub_bytes;
  PeerHeader* me;
  int* h_err;
  unsigned long long timeout_ticks;              // wall_clock64 ticks (100 MHz)
};

// bytes [r0, r1) of a slot's byte stream, the whole workgroup: slot -> pieces (TO_SLOT false) or pieces -> slot (true).  Four
// independent 16-byte loads per lane are in flight before the first store (uncached HBM and remote stores have microseconds of latency).
template <bool TO_SLOT>
__device__ __forceinline__ void copy_range(const Piece* pc, int np, char* slot, unsigned long long r0, unsigned long long r1) {
  const unsigned tid = threadIdx.x, nt = blockDim.x;
  for (int i = 0; i < np; ++i) {
    const unsigned long long a = pc[i].off > r0 ? pc[i].off : r0, e = pc[i].off + pc[i].bytes < r1 ? pc[i].off + pc[i].bytes : r1;
    if (a >= e) continue;
    const char* src = TO_SLOT ? pc[i].src + (a - pc[i].off) : slot + a;
    char* dst = TO_SLOT ? slot + a : pc[i].dst + (a - pc[i].off);
    const unsigned long long bytes = e - a;
    if ((((unsigned long long)src | (unsigned long long)dst | bytes) & 15ull) == 0) {
      const uint4* sp = (const uint4*)src; uint4* dp = (uint4*)dst;
      const unsigned long long n = bytes >> 4;
      unsigned long long q = tid;
      for (; q + 3ull * nt < n; q += 4ull * nt) {
        const uint4 v0 = sp[q], v1 = sp[q + nt], v2 = sp[q + 2ull * nt], v3 = sp[q + 3ull * nt];
        dp[q] = v0; dp[q + nt] = v1; dp[q + 2ull * nt] = v2; dp[q + 3ull * nt] = v3;
      }
      for (; q < n; q += nt) dp[q] = sp[q];
    } else if ((((unsigned long long)src | (unsigned long long)dst | bytes) & 3ull) == 0) {
      const unsigned* sp = (const unsigned*)src; unsigned* dp = (unsigned*)dst;
      for (unsigned long long q = tid; q < (bytes >> 2); q += nt) dp[q] = sp[q];
    } else {
      for (unsigned long long q = tid; q < bytes; q += nt) dst[q] = src[q];
    }
  }
}

// waits until *word >= want (system scope).  false: timed out or aborted (the rank's error word is raised)
__device__ bool wait_word(const unsigned* word, unsigned want, const XferArgs& a) {
  __shared__ int ok;
  __syncthreads();                                           // (the previous wait's result has been read by everyone)
  if (threadIdx.x == 0) {
    int good = 1;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      if (__hip_atomic_load(&a.me->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || wall_clock64() - t0 > a.timeout_ticks) {
        good = 0;
        // this rank's communicator is dead from here on: the abort word makes every workgroup of this launch and every launch already
        // queued behind it (the rest of the step, the rest of a replayed graph) leave at once instead of spinning for the time-out again
        __hip_atomic_store(&a.me->abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// gridDim.x = 4 * kBlocksPerRole: roles 0/1 push to the lower/upper neighbour, 2/3 pull from them.  The push roles come first in
// the dispatch order, so a launch's pulls never keep its own pushes from starting.  Workgroup `part` of a role owns the sub-chunks
// part, part + kBlocksPerRole, ... of the slot.
__global__ __launch_bounds__(256) void peer_xfer_kernel(XferArgs a) {
  const int role = blockIdx.x / kBlocksPerRole, part = blockIdx.x % kBlocksPerRole;
  const int s = role & 1;
  const bool push = role < 2;
  if (!a.active[s]) return;
  // an earlier wait of this rank timed out, or the group was aborted (peer_abort, a neighbour's failure): nothing is moved, nothing
  // is waited for, and the host finds the error word raised (one load of a device word per wave)
  if (__hip_atomic_load(&a.me->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
    if (threadIdx.x == 0 && part == 0) __hip_atomic_store(a.h_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // this launch's chunk number: the role's device counter + 1 (stored back by the role's last workgroup, below: every workgroup of
  // the role has read it by then -- it counts itself done only after its work)
  const unsigned n = __hip_atomic_load(&a.me->seq[role], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const unsigned par = n & 1u;
  const unsigned nsub = (unsigned)((a.used[s] + a.sub_bytes - 1) / a.sub_bytes);
  bool ok = true;
  if (push) {
    // the slot is free once the neighbour has taken chunk n - 2 out of it
    if (n > 2) ok = wait_word(&a.me->credit[s], n - 2, a);
    for (unsigned j = part; ok && j < nsub; j += kBlocksPerRole) {
      const unsigned long long r0 = (unsigned long long)j * a.sub_bytes, r1 = r0 + a.sub_bytes < a.used[s] ? r0 + a.sub_bytes : a.used[s];
      // npush < 0: a DIRECT send -- the kernel that produced the planes has stored them into the slot itself (it ran before this
      // launch on the stream: its stores are complete); only the ready words remain to be raised
      if (a.npush[s] >= 0) copy_range<true>(a.push[s], a.npush[s], a.remote_slot[s][par], r0, r1);
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&a.remote_ready[s][par][j], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  } else {
    for (unsigned j = part; ok && j < nsub; j += kBlocksPerRole) {
      ok = wait_word(&a.local_ready[s][par][j], n, a);
      if (!ok) break;
      const unsigned long long r0 = (unsigned long long)j * a.sub_bytes, r1 = r0 + a.sub_bytes < a.used[s] ? r0 + a.sub_bytes : a.used[s];
      copy_range<false>(a.pull[s], a.npull[s], const_cast<char*>(a.local_slot[s][par]), r0, r1);
    }
  }
  if (ok) {
    // the role's last workgroup: stores the chunk number back and -- pull -- hands the credit for the emptied slot to the neighbour.
    // (The role is active in every launch of its direction, so after chunk n its counter stands at n * kBlocksPerRole.)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(&a.me->done_count[role], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == n * (unsigned)kBlocksPerRole) {
        __hip_atomic_store(&a.me->seq[role], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!push) __hip_atomic_store(a.remote_credit[s], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

struct PeerComm { Peer* p; };

int peer_failed(Peer* p) {
  if (p->h_err && *(volatile int*)p->h_err)
    return fnx::set_error(FNX_ECOMM, "peer-store exchange: a wait for a neighbour timed out (%.0f s) or the group was aborted", p->timeout_s);
  return FNX_OK;
}

int peer_exchange_impl(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream, bool direct);
int peer_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) { return peer_exchange_impl(vctx, segs, nsegs, stream, false); }
// DIRECT sends (FnxSlabComm.direct_begin / direct_exchange): the next chunk's slot in each neighbour's mailbox, segment i at i * stride.
// The slot is free without a wait: this rank's previous exchange launch has completed (stream order) and saw the neighbour's chunk
// n - 1 arrive, which that neighbour pushed behind ITS launch n - 2, whose pull emptied the slot chunk n goes into.
int peer_direct_begin(void* vctx, size_t bytes, int nsegs, void* dst[2][2], const unsigned* select[2], size_t* seg_stride,
                      void** start_clock, void* /*stream*/) {
  Peer* p = ((PeerComm*)vctx)->p;
  if (int rc = peer_failed(p)) return rc;
  const size_t stride = (bytes + 255) & ~(size_t)255;
  if (bytes == 0 || nsegs < 1 || nsegs > kMaxPieces || stride * (size_t)nsegs > p->slot_bytes)
    return fnx::set_error(FNX_EINVAL, "peer-store direct send: %d segments of %zu bytes do not fit a mailbox slot", nsegs, bytes);
  // both slots of each neighbour's mailbox; the producing kernel takes the one of chunk *select + 1 (the push counter of that side)
  for (int s = 0; s < 2; ++s) {
    const bool has = s == 0 ? p->rank > 0 : p->rank < p->nranks - 1;
    for (int par = 0; par < 2; ++par) dst[s][par] = has ? (void*)p->mailbox(p->nb[s], 1 - s, par) : nullptr;
    select[s] = has ? &p->hdr()->seq[s] : nullptr;
  }
  *seg_stride = stride;
  if (start_clock) *start_clock = nullptr;
  return FNX_OK;
}
int peer_direct_exchange(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream) { return peer_exchange_impl(vctx, segs, nsegs, stream, true); }

int peer_exchange_impl(void* vctx, const FnxSlabSeg* segs, int nsegs, void* stream, bool direct) {
  PeerComm* c = (PeerComm*)vctx;
  Peer* p = c->p;
  if (int rc = peer_failed(p)) return rc;
  const bool has[2] = {p->rank > 0, p->rank < p->nranks - 1};
  // cut the segments into pieces that fit a mailbox slot, then fill launches with up to kMaxPieces pieces and slot_bytes bytes per
  // direction.  A segment's lower and upper halves have the same size, and both neighbours cut their segments the same way (same
  // sizes, same order), so piece i of my push towards s is piece i of that neighbour's pull from 1 - s.
  struct Cut { int seg; size_t at, bytes; };
  std::vector<Cut> cuts;
  for (int i = 0; i < nsegs; ++i)
    for (size_t at = 0; at < segs[i].bytes;) {
      const size_t n = segs[i].bytes - at < p->slot_bytes ? segs[i].bytes - at : p->slot_bytes;
      cuts.push_back(Cut{i, at, n});
      at += n;
    }
  if (direct && (cuts.size() != (size_t)nsegs || nsegs > kMaxPieces))
    return fnx::set_error(FNX_EINVAL, "peer-store direct exchange: the segments are not the ones direct_begin sized");
  size_t k = 0;
  while (k < cuts.size()) {
    XferArgs a{};
    size_t used = 0;
    int np = 0;
    Piece pl[kMaxPieces]; int segi[kMaxPieces];
    while (k < cuts.size() && np < kMaxPieces && used + cuts[k].bytes <= p->slot_bytes) {
      pl[np] = Piece{nullptr, nullptr, (unsigned long long)cuts[k].bytes, (unsigned long long)used};
      segi[np] = (int)k;
      used += (cuts[k].bytes + 255) & ~(size_t)255;
      ++np; ++k;
    }
    for (int s = 0; s < 2; ++s) {
      if (!has[s]) continue;
      int nsend = 0, nrecv = 0;
      for (int i = 0; i < np; ++i) {
        const Cut& cu = cuts[segi[i]];
        const FnxSlabSeg& g = segs[cu.seg];
        const char* snd = (const char*)(s == 0 ? g.send_lo : g.send_hi);
        char* rcv = (char*)(s == 0 ? g.recv_lo : g.recv_hi);
        if (snd) a.push[s][nsend++] = Piece{snd + cu.at, nullptr, pl[i].bytes, pl[i].off};
        if (rcv) a.pull[s][nrecv++] = Piece{nullptr, rcv + cu.at, pl[i].bytes, pl[i].off};
      }
      a.npush[s] = direct ? -1 : nsend; a.npull[s] = nrecv;
      // (a direction with nothing to send still hands over the chunk: both ends count chunks, not bytes)
      PeerHeader* nh = (PeerHeader*)p->nb[s];
      a.active[s] = 1;
      for (int par = 0; par < 2; ++par) {
        a.remote_slot[s][par] = p->mailbox(p->nb[s], 1 - s, par);
        a.local_slot[s][par] = p->mailbox(p->region, s, par);
        a.remote_ready[s][par] = nh->ready[1 - s][par];
        a.local_ready[s][par] = p->hdr()->ready[s][par];
      }
      a.remote_credit[s] = &nh->credit[1 - s];
      a.used[s] = used;
    }
    a.sub_bytes = p->sub_bytes;
    a.me = p->hdr();
    a.h_err = p->h_err;
    a.timeout_ticks = (unsigned long long)(p->timeout_s * 1e8);
    peer_xfer_kernel<<<4 * kBlocksPerRole, 256, 0, (hipStream_t)stream>>>(a);
    PEER_HIP(hipGetLastError());
  }
  return FNX_OK;
}

// ---- control path: all-reduce of n floats along the chain of ranks, through the headers, driven by the host (rank 0 -> ... -> last
// accumulates in rank order, the total travels back down): the same bits on every rank
int host_write(Peer* p, void* dst, const void* src, size_t bytes) {
  PEER_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return FNX_OK;
}
int host_wait_seq(Peer* p, const unsigned* word, unsigned want) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(p->timeout_s);
  for (;;) {
    unsigned v = 0, ab = 0;
    PEER_HIP(hipMemcpy(&v, word, 4, hipMemcpyDeviceToHost));
    if (v >= want) return FNX_OK;
    PEER_HIP(hipMemcpy(&ab, &p->hdr()->abort_word, 4, hipMemcpyDeviceToHost));
    if (ab) return fnx::set_error(FNX_ECOMM, "peer-store all-reduce: the group was aborted");
    if (std::chrono::steady_clock::now() > deadline) return fnx::set_error(FNX_ECOMM, "peer-store all-reduce: timed out after %.0f s waiting for a neighbour", p->timeout_s);
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
int peer_allreduce(void* vctx, float* x, int n, void* stream, bool sum) {
  Peer* p = ((PeerComm*)vctx)->p;
  if (int rc = peer_failed(p)) return rc;
  if (n < 1 || n > kRedMax) return fnx::set_error(FNX_EINVAL, "peer-store all-reduce: n must be in [1, %d]", kRedMax);
  std::vector<float> acc((size_t)n), in((size_t)n);
  PEER_HIP(hipMemcpyAsync(acc.data(), x, (size_t)n * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  PEER_HIP(hipStreamSynchronize((hipStream_t)stream));
  const unsigned seq = ++p->red_seq;
  PeerHeader* me = p->hdr();
  if (p->rank > 0) {
    if (int rc = host_wait_seq(p, &me->red_up_seq, seq)) return rc;
    PEER_HIP(hipMemcpy(in.data(), me->red_up, (size_t)n * 4, hipMemcpyDeviceToHost));
    // max: a NaN on EITHER side wins (a blown-up field must fail the CFL / residual guard whichever rank holds it)
    for (int i = 0; i < n; ++i) acc[i] = sum ? in[i] + acc[i] : ((acc[i] != acc[i] || in[i] != in[i]) ? std::numeric_limits<float>::quiet_NaN() : (acc[i] > in[i] ? acc[i] : in[i]));
  }
  if (p->rank < p->nranks - 1) {
    PeerHeader* up = (PeerHeader*)p->nb[1];
    if (int rc = host_write(p, up->red_up, acc.data(), (size_t)n * 4)) return rc;
    if (int rc = host_write(p, &up->red_up_seq, &seq, 4)) return rc;
    if (int rc = host_wait_seq(p, &me->red_dn_seq, seq)) return rc;
    PEER_HIP(hipMemcpy(acc.data(), me->red_dn, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  if (p->rank > 0) {
    PeerHeader* dn = (PeerHeader*)p->nb[0];
    if (int rc = host_write(p, dn->red_dn, acc.data(), (size_t)n * 4)) return rc;
    if (int rc = host_write(p, &dn->red_dn_seq, &seq, 4)) return rc;
  }
  PEER_HIP(hipMemcpyAsync(x, acc.data(), (size_t)n * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  PEER_HIP(hipStreamSynchronize((hipStream_t)stream));
  return FNX_OK;
}
int peer_allreduce_max(void* c, float* x, int n, void* s) { return peer_allreduce(c, x, n, s, false); }
int peer_allreduce_sum(void* c, float* x, int n, void* s) { return peer_allreduce(c, x, n, s, true); }
void peer_comm_destroy(void* vctx) { delete (PeerComm*)vctx; }
void peer_abort(void* vctx) {
  Peer* p = ((PeerComm*)vctx)->p;
  const unsigned one = 1;
  (void)hipMemcpy(&p->hdr()->abort_word, &one, 4, hipMemcpyHostToDevice);
  for (int s = 0; s < 2; ++s)
    if (p->nb[s]) (void)hipMemcpy(&((PeerHeader*)p->nb[s])->abort_word, &one, 4, hipMemcpyHostToDevice);
}

}  // namespace

extern "C" {

int fnx_slab_peer_create(void** peer, int rank, int nranks, size_t mailbox_bytes, void* handle_out) {
  if (!peer || !handle_out || nranks < 1 || rank < 0 || rank >= nranks || mailbox_bytes < 4096)
    return fnx::set_error(FNX_EINVAL, "slab_peer_create: bad arguments (mailbox_bytes >= 4096)");
  Peer* p = new (std::nothrow) Peer();
  if (!p) return fnx::set_error(FNX_EINVAL, "out of host memory");
  p->rank = rank; p->nranks = nranks;
  p->slot_bytes = (mailbox_bytes + 255) & ~(size_t)255;
  p->region_bytes = kHeaderBytes + 4 * p->slot_bytes;
  while (p->sub_bytes * kMaxSub < p->slot_bytes) p->sub_bytes *= 2;
  hipError_t e = hipGetDevice(&p->device);
  // uncached: the neighbour's stores must be seen by a kernel that is already running here, and ours by theirs
  if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&p->region, p->region_bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(p->region, 0, kHeaderBytes);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_err, sizeof(int), hipHostMallocMapped);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  PeerBlob b{};
  if (e == hipSuccess) e = hipIpcGetMemHandle(&b.ipc, p->region);
  if (e != hipSuccess) {
    const int rc = fnx::set_error(FNX_EHIP, "slab_peer_create: %s", hipGetErrorString(e));
    fnx_slab_peer_free(p);
    return rc;
  }
  *p->h_err = 0;
  b.magic = kMagic; b.rank = rank; b.nranks = nranks; b.pid = (int)getpid(); b.slot_bytes = p->slot_bytes;
  b.raw = (unsigned long long)(uintptr_t)p->region;
  b.token = process_token(); b.device = p->device;
  memset(handle_out, 0, FNX_PEER_HANDLE_BYTES);
  memcpy(handle_out, &b, sizeof(b));
  *peer = p;
  return FNX_OK;
}

int fnx_slab_peer_failed(void* peer) {
  Peer* p = (Peer*)peer;
  if (!p) return fnx::set_error(FNX_EINVAL, "fnx_slab_peer_failed: NULL peer");
  return peer_failed(p);
}

int fnx_slab_peer_set_timeout(void* peer, double seconds) {
  Peer* p = (Peer*)peer;
  if (!p || !(seconds > 0.0) || seconds > 3600.0) return fnx::set_error(FNX_EINVAL, "slab_peer_set_timeout: seconds in (0, 3600]");
  p->timeout_s = seconds;
  return FNX_OK;
}

int fnx_slab_comm_peer(FnxSlabComm* out, void* peer, const void* handle_lo, const void* handle_hi) {
  Peer* p = (Peer*)peer;
  if (!out || !p) return fnx::set_error(FNX_EINVAL, "slab_comm_peer: NULL argument");
  const void* hs[2] = {handle_lo, handle_hi};
  for (int s = 0; s < 2; ++s) {
    const bool need = s == 0 ? p->rank > 0 : p->rank < p->nranks - 1;
    if (!need) continue;
    if (!hs[s]) return fnx::set_error(FNX_EINVAL, "slab_comm_peer: rank %d of %d needs the handle of its %s neighbour", p->rank, p->nranks, s ? "upper" : "lower");
    if (p->nb[s]) continue;                                  // already attached
    PeerBlob b;
    memcpy(&b, hs[s], sizeof(b));
    if (b.magic != kMagic || b.nranks != p->nranks || b.rank != p->rank + (s ? 1 : -1) || b.slot_bytes != p->slot_bytes)
      return fnx::set_error(FNX_ECOMM, "slab_comm_peer: the %s handle is not that of rank %d of %d with %zu-byte mailbox slots", s ? "upper" : "lower",
                            p->rank + (s ? 1 : -1), p->nranks, p->slot_bytes);
    if (b.pid == (int)getpid() && b.token == process_token()) {
      // the same address space (slabs driven by threads): no mapping needed -- on the same device.  Two devices of one process would
      // need peer access enabled, and ranks that share a device queue their spinning exchange launches behind each other on the
      // process's few hardware queues: neither is a tested configuration, so the pointer is only taken for the owner's device
      if (b.device != p->device)
        return fnx::set_error(FNX_ECOMM, "slab_comm_peer: ranks %d and %d live in one process on devices %d and %d; same-process ranks are "
                              "supported on one device only (use one process per GPU)", p->rank, b.rank, p->device, b.device);
      p->nb[s] = (char*)(uintptr_t)b.raw;
    } else {
      void* m = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&m, b.ipc, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fnx::set_error(FNX_ECOMM, "slab_comm_peer: hipIpcOpenMemHandle failed (%s)", hipGetErrorString(e));
      p->nb[s] = (char*)m; p->nb_ipc[s] = true;
    }
  }
  PeerComm* c = new (std::nothrow) PeerComm{p};
  if (!c) return fnx::set_error(FNX_EINVAL, "out of host memory");
  out->ctx = c; out->exchange = peer_exchange; out->allreduce_max = peer_allreduce_max; out->allreduce_sum = peer_allreduce_sum;
  out->destroy = peer_comm_destroy; out->abort = peer_abort;
  out->direct_begin = peer_direct_begin; out->direct_exchange = peer_direct_exchange;
  return FNX_OK;
}

void fnx_slab_peer_free(void* peer) {
  Peer* p = (Peer*)peer;
  if (!p) return;
  for (int s = 0; s < 2; ++s)
    if (p->nb[s] && p->nb_ipc[s]) (void)hipIpcCloseMemHandle(p->nb[s]);
  if (p->region) (void)hipFree(p->region);
  if (p->h_err) (void)hipHostFree(p->h_err);
  delete p;
}

}  // extern "C"
