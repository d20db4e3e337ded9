// Times the 3D two-sweep Jacobi pass (fnx_jacobi_pass, nsweeps = 2) with HIP events on the shapes the benchmarks use.
// build: hipcc -O2 -o jacobi3d_bench.bin jacobi3d_bench.cpp -I../../include -L../../fluidnet_cxx_amd -lfluidnet_hip -Wl,-rpath,'$ORIGIN/../../fluidnet_cxx_amd'
// usage: jacobi3d_bench.bin [obstacles=0|1] [shape index]      env FNX_JACOBI_ZCHUNK to vary the z-chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fluidnet_hip.h"

int main(int argc, char** argv) {
  const int obstacles = argc > 1 ? atoi(argv[1]) : 0;
  const int only = argc > 2 ? atoi(argv[2]) : -1;      // shape index, -1 = all
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int shapes[][3] = {{256, 256, 256}, {64, 512, 512}, {76, 512, 512}, {128, 128, 128}};
  int idx = -1;
  for (auto& sh : shapes) {
    if (++idx != only && only >= 0) continue;
    FnxGrid g{1, sh[0], sh[1], sh[2], 1, 0, 0, 0};
    const size_t n = (size_t)sh[0] * sh[1] * sh[2];
    float *flags, *div, *p, *q; void* ws;
    const size_t wsb = fnx_workspace_bytes(&g, FNX_OP_JACOBI);
    hipMalloc(&flags, n * 4); hipMalloc(&div, n * 4); hipMalloc(&p, n * 4); hipMalloc(&q, n * 4); hipMalloc(&ws, wsb);
    std::vector<float> h(n), f(n, 1.f);
    unsigned st = 12345;
    for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) / 16777216.f - 0.5f; }
    for (int k = 0; k < sh[0]; ++k) for (int j = 0; j < sh[1]; ++j) for (int i = 0; i < sh[2]; ++i) {
      const bool border = k == 0 || j == 0 || i == 0 || k == sh[0] - 1 || j == sh[1] - 1 || i == sh[2] - 1;
      const bool obst = obstacles && ((i / 7 + j / 5 + k / 3) % 11 == 0);
      if (border || obst) f[((size_t)k * sh[1] + j) * sh[2] + i] = 2.f;
    }
    hipMemcpy(flags, f.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(div, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    fnx_jacobi_pass(&g, flags, div, p, q, 2, 0, 0, ws, wsb, 0, s);      // builds the mask
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, s);
      for (int it = 0; it < 20; ++it) {
        fnx_jacobi_pass(&g, flags, div, p, q, 2, 0, 0, ws, wsb, 1, s);
        fnx_jacobi_pass(&g, flags, div, q, p, 2, 0, 0, ws, wsb, 1, s);
      }
      hipEventRecord(e1, s); hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double us = best * 1000.0 / 40.0;
    printf("%3dx%3dx%3d obstacles=%d: %.1f us per 2-sweep pass  (%.2f Gcell-sweeps/s, algorithmic %.0f GB/s)\n", sh[0], sh[1],
           sh[2], obstacles, us, 2.0 * n / us / 1e3, 32.0 * n / us / 1e3);
    hipFree(flags); hipFree(div); hipFree(p); hipFree(q); hipFree(ws);
  }
  return 0;
}
