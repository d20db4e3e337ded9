// What does a grid-wide barrier inside ONE launch cost against a kernel boundary?  (would one cooperative
// launch per step beat the five back-to-back launches of a 128^2 step?)
// Every phase writes one float per thread and, behind the barrier / the kernel boundary, reads the value a thread of ANOTHER
// workgroup (another XCD: blockIdx + 1) wrote in the phase before -- so the data really has to cross the XCDs' L2s, as the
// phases of a fluid step would make it.  Reports us per phase for
//   (a) P kernels back to back on one stream,  (b) the same captured in a hipGraph,
//   (c) one launch with P - 1 flat barriers (one agent-scope counter),
//   (d) one launch with P - 1 two-level barriers (a counter per blockIdx % 8 "XCD", then one across the eight).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/grid_barrier.bin tools/ubench/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void phase_body(const float* in, float* out, int phase) {
  const int nb = gridDim.x, nt = blockDim.x;
  const int src = ((blockIdx.x + 1) % nb) * nt + threadIdx.x;
  out[blockIdx.x * nt + threadIdx.x] = in[src] + 1.0f + 0.f * phase;
}

__global__ void phase_kernel(const float* in, float* out, int phase) { phase_body(in, out, phase); }

__device__ __forceinline__ void barrier_flat(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// the same with relaxed polls: ONE release fence before the arrival, ONE acquire fence after the last poll (an acquire load
// invalidates the caches at every poll)
template <int SLEEP>
__device__ __forceinline__ void barrier_relaxed(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// flag array: workgroup i stores the barrier's generation into flags[i] (no read-modify-write, nothing serialises at one
// address); wave 0 polls all the flags, 64 per load, until every one has reached the generation
__device__ __forceinline__ void barrier_flags(unsigned* flags, unsigned gen, unsigned nb) {
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(flags + blockIdx.x, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (;;) {
      bool ok = true;
      for (unsigned i = threadIdx.x; i < nb; i += 64)
        ok = ok && (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) >= 0;
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// XCD-aware: the L2 write-back (release) and invalidate (acquire) are per-XCD operations, and every workgroup's fence queues at
// its XCD's L2 -- that, not the arrival atomics, is what makes a barrier cost ~40 ns per workgroup.  Here the workgroups of an XCD
// (HW_REG_XCC_ID) arrive at a per-XCD counter without fences; the last one to arrive does ONE write-back for the XCD, arrives at
// the top counter, waits for the other XCDs' leaders, does ONE invalidate and releases its XCD's workgroups, which only drop
// their own L1.  ctr: [xcd * 32] arrivals, [256 + xcd * 32] go flags, [512] top, [544 + xcd] workgroups on the XCD, [560] XCDs in use
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
__device__ __forceinline__ void barrier_xcd(unsigned* ctr, unsigned gen, unsigned xcd, unsigned nx, unsigned nxcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ctr + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1 == gen * nx) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctr + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nxcd) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(ctr + 256 + xcd * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(ctr + 256 + xcd * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

// ctr[0..7 * 32]: one counter per group of workgroups (blockIdx % 8), ctr[8 * 32]: the top counter; 128 bytes apart
__device__ __forceinline__ void barrier_two_level(unsigned* ctr, unsigned gen, unsigned per_group) {
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* mine = ctr + (blockIdx.x & 7) * 32;
    unsigned* top = ctr + 8 * 32;
    const unsigned old = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gen * per_group) __hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * 8u) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int MODE>
__global__ void fused_kernel(float* a, float* b, unsigned* ctr, int phases, unsigned base) {
  const unsigned nb = gridDim.x;
  unsigned xcd = 0, nx = 0, nxcd = 0;
  if (MODE == 1) {       // census (first launch of a run: base == 0): who is where; one flat barrier on its own counter
    xcd = xcc_id();
    if (base == 0) {
      if (threadIdx.x == 0) {
        const unsigned o = __hip_atomic_fetch_add(ctr + 544 + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o == 0) __hip_atomic_fetch_add(ctr + 560, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      barrier_flat(ctr + 576, nb);
    }
    nx = __hip_atomic_load(ctr + 544 + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nxcd = __hip_atomic_load(ctr + 560, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int p = 0; p < phases; ++p) {
    phase_body(p & 1 ? b : a, p & 1 ? a : b, p);
    if (p + 1 < phases) {
      if (MODE == 0) barrier_flat(ctr, (base + (unsigned)(p + 1)) * nb);
      else if (MODE == 2) barrier_relaxed<1>(ctr, (base + (unsigned)(p + 1)) * nb);
      else if (MODE == 3) barrier_flags(ctr, base + (unsigned)(p + 1), nb);
      else barrier_xcd(ctr, base + (unsigned)(p + 1), xcd, nx, nxcd);
    }
  }
}

int main(int argc, char** argv) {
  const int reps = 200;
  const int shapes[][2] = { {64, 256}, {128, 256}, {256, 256}, {16, 1024}, {32, 1024}, {64, 1024}, {256, 1024} };
  for (auto& sh : shapes) {
    const int nb = sh[0], nt = sh[1];
    const size_t n = (size_t)nb * nt;
    float *a, *b; unsigned* ctr;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&ctr, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int phases : {1, 9}) {
      float t_sep, t_graph, t_flat, t_two, t_rel, t_rel0;
      // (a) separate launches
      CK(hipMemsetAsync(a, 0, n * 4, s)); CK(hipMemsetAsync(b, 0, n * 4, s));
      auto run_sep = [&]() { for (int p = 0; p < phases; ++p) phase_kernel<<<nb, nt, 0, s>>>(p & 1 ? b : a, p & 1 ? a : b, p); };
      for (int i = 0; i < 20; ++i) run_sep();
      CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; ++i) run_sep(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t_sep, e0, e1));
      // (b) the same as a graph
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); run_sep(); CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t_graph, e0, e1));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      // (c), (d) one launch; the counters only ever count up (`base` = barriers passed by the launches before: a product
      // kernel would have its last workgroup reset them instead)
      unsigned base = 0;
      auto run_fused = [&](int mode) {
        if (mode == 0) fused_kernel<0><<<nb, nt, 0, s>>>(a, b, ctr, phases, base);
        else if (mode == 1) fused_kernel<1><<<nb, nt, 0, s>>>(a, b, ctr, phases, base);
        else if (mode == 2) fused_kernel<2><<<nb, nt, 0, s>>>(a, b, ctr, phases, base);
        else fused_kernel<3><<<nb, nt, 0, s>>>(a, b, ctr, phases, base);
        base += (unsigned)(phases - 1);
      };
      float* tt[4] = { &t_flat, &t_two, &t_rel, &t_rel0 };
      for (int mode = 0; mode < 4; ++mode) {
        CK(hipMemsetAsync(ctr, 0, 4096, s)); base = 0;
        CK(hipMemsetAsync(a, 0, n * 4, s)); CK(hipMemsetAsync(b, 0, n * 4, s));
        for (int i = 0; i < 20; ++i) run_fused(mode);
        CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; ++i) run_fused(mode); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(tt[mode], e0, e1));
        // one more run from zeroed buffers: the chain a -> b -> a ... adds 1 per phase
        CK(hipMemsetAsync(a, 0, n * 4, s)); CK(hipMemsetAsync(b, 0, n * 4, s));
        run_fused(mode);
        CK(hipStreamSynchronize(s));
        std::vector<float> h(n);
        CK(hipMemcpy(h.data(), (phases & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
        const float want = (float)phases;
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += h[i] != want;
        if (bad) printf("  MODE %d: %zu of %zu elements wrong (want %g, got e.g. %g)\n", mode, bad, n, want, h[0]);
      }
      printf("%4d x %4d threads, %d phases: separate %.2f us/step (%.2f/phase)  graph %.2f (%.2f)  fused flat %.2f (%.2f)  XCD-aware %.2f (%.2f)  relaxed polls %.2f (%.2f)  flag array %.2f (%.2f)\n",
             nb, nt, phases, 1e3f * t_sep / reps, 1e3f * t_sep / reps / phases, 1e3f * t_graph / reps, 1e3f * t_graph / reps / phases,
             1e3f * t_flat / reps, 1e3f * t_flat / reps / phases, 1e3f * t_two / reps, 1e3f * t_two / reps / phases,
             1e3f * t_rel / reps, 1e3f * t_rel / reps / phases, 1e3f * t_rel0 / reps, 1e3f * t_rel0 / reps / phases);
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctr));
  }
  return 0;
}
