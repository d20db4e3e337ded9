// Gather throughput on gfx950: two adjacent 4-byte loads vs one 4-byte-aligned 8-byte load per lane (the x-pair of a
// bilinear / trilinear corner set).  Addresses are "advection-like": lane's own cell plus a small per-lane offset.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_bench.hip -o tools/ubench/gather_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct __attribute__((packed, aligned(4))) F2 { float a, b; };

template <int MODE>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ p, const int* __restrict__ off, float* __restrict__ o,
                                              int W, int HW, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int d = off[t];
  float acc = 0.f;
#pragma unroll 4
  for (int it = 0; it < 32; ++it) {
    const float* q = p + t + d + (it & 3) * W + (it >> 2) * HW;
    if (MODE == 0) { acc += q[0] + q[1] + q[W] + q[W + 1]; }
    else { const F2 a = *(const F2*)q; const F2 b = *(const F2*)(q + W); acc += a.a + a.b + b.a + b.b; }
  }
  o[t] = acc;
}

int main() {
  const int W = 256, HW = 256 * 256, n = 256 * 256 * 200;
  const size_t tot = (size_t)256 * 256 * 256;
  float *p, *o; int* off;
  hipMalloc(&p, tot * 4); hipMalloc(&o, tot * 4); hipMalloc(&off, tot * 4);
  hipMemset(p, 0, tot * 4);
  std::vector<int> h(n);
  unsigned s = 1;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 20) % 3) - 1 + ((int)((s >> 24) % 3) - 1) * W + W + 1; }
  hipMemcpy(off, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) gather<0><<<n / 256, 256>>>(p, off, o, W, HW, n); else gather<1><<<n / 256, 256>>>(p, off, o, W, HW, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s: %.1f us for %d lanes x 128 corner values (%.1f G values/s)\n", mode == 0 ? "4 x dword  " : "2 x dwordx2", best * 1e3,
           n, 128.0 * n / best / 1e6);
  }
  return 0;
}
