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
// the solver's mix: read two float fields and a byte field, write one float field (13 B per cell, reads : writes = 2.25 : 1)
__global__ __launch_bounds__(256) void mix(const float4* __restrict__ p, const float4* __restrict__ d, const unsigned* __restrict__ m,
                                           float4* __restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 a = p[i], b = d[i]; const unsigned c = m[i];
    q[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, (c & 0x01010101u) ? a.w + b.w : 0.f);
  }
}
// the same mix at the solver's access width: one dword (and one mask byte) per lane and instruction
__global__ __launch_bounds__(256) void mix1(const float* __restrict__ p, const float* __restrict__ d, const unsigned char* __restrict__ m,
                                            float* __restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256 * 4) {
    const size_t s = (size_t)gridDim.x * 256;
    float a[4], b[4]; unsigned c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t k = i + u * s; if (k < n) { a[u] = p[k]; b[u] = d[k]; c[u] = m[k]; } }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t k = i + u * s; if (k < n) q[k] = c[u] ? a[u] + b[u] : 0.f; }
  }
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
  // footprint = 13 B per cell: p (a), div (a + n), out (b), mask (b + n)
  for (size_t mb : {104, 156, 182, 208, 234, 312, 416, 832}) {
    const size_t cells = (mb << 20) / 13, n = cells / 4;
    const float4 *pp = a, *dd = a + n; float4* qq = b; const unsigned* mm = (const unsigned*)(b + n);
    for (int w = 0; w < 3; ++w) mix<<<2048, 256>>>(pp, dd, mm, qq, n);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) mix<<<2048, 256>>>(pp, dd, mm, qq, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mix   footprint %5zu MiB: %7.2f TB/s (%.1f us per pass; reads %.2f TB/s)\n", mb, (double)n * 4 * 13 * reps / ms / 1e9, ms / reps * 1e3,
           (double)n * 4 * 9 * reps / ms / 1e9);
  }
  for (size_t mb : {104, 208, 832}) {
    const size_t n = (mb << 20) / 13;
    const float *pp = (const float*)a, *dd = pp + n; float* qq = (float*)b; const unsigned char* mm = (const unsigned char*)(qq + n);
    for (int blocks : {2048, 4096, 8192}) {
      for (int w = 0; w < 3; ++w) mix1<<<blocks, 256>>>(pp, dd, mm, qq, n);
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) mix1<<<blocks, 256>>>(pp, dd, mm, qq, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mix, dword accesses, %d blocks, footprint %5zu MiB: %7.2f TB/s (%.1f us per pass)\n", blocks, mb, (double)n * 13 * reps / ms / 1e9, ms / reps * 1e3);
    }
  }
  return 0;
}
