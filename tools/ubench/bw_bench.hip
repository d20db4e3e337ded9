// Streaming bandwidth of the memory system by footprint (Infinity Cache 256 MiB): read-only and copy, float4 per lane.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bw_bench.bin tools/ubench/bw_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ p, float* o, size_t n) {
  float a = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = p[i]; a += (v.x + v.y) + (v.z + v.w); }
  if (a == 1234.5f) o[0] = a;
}
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ p, float4* __restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
int main() {
  float4 *a, *b; float* o; size_t maxb = (size_t)2048 << 20;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&o, 64); hipMemset(a, 0, maxb); hipMemset(b, 0, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {32, 64, 128, 192, 256, 512, 1024, 2048}) {
    size_t n = (mb << 20) / 16;
    for (int mode = 0; mode < 2; ++mode) {
      size_t nn = mode ? n / 2 : n;                       // copy: footprint = src + dst = mb
      for (int w = 0; w < 3; ++w) { if (mode) cp<<<2048, 256>>>(a, b, nn); else rd<<<2048, 256>>>(a, o, nn); }
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) { if (mode) cp<<<2048, 256>>>(a, b, nn); else rd<<<2048, 256>>>(a, o, nn); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double bytes = (double)(mb << 20) * reps;
      printf("%-5s footprint %5zu MiB: %7.2f TB/s (%.1f us per pass)\n", mode ? "copy" : "read", mb, bytes / ms / 1e9, ms / reps * 1e3);
    }
  }
  return 0;
}
