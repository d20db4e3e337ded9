// Times fnx_jacobi (no residual -> kernel launches only) and an empty-kernel chain with HIP events.
// build: hipcc -O2 -o jacobi_bench jacobi_bench.cpp -I../../include -L../../fluidnet_cxx_amd -lfluidnet_hip -Wl,-rpath,'$ORIGIN/../../fluidnet_cxx_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fluidnet_hip.h"

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

int main(int argc, char** argv) {
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  // empty kernel chain
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, s);
    for (int i = 0; i < 1000; ++i) empty_kernel<<<256, 256, 0, s>>>(nullptr);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("empty kernel chain: %.2f us per launch\n", ms);
  for (int res : {128, 256, 512, 1024, 2048, 4096}) {
    for (int iters : {8, 28, 100}) {
      FnxGrid g{1, 1, res, res, 0, 0};
      size_t n = (size_t)res * res;
      float *flags, *div, *p; void* ws;
      size_t wsb = fnx_workspace_bytes(&g, FNX_OP_JACOBI);
      hipMalloc(&flags, n * 4); hipMalloc(&div, n * 4); hipMalloc(&p, n * 4); hipMalloc(&ws, wsb);
      fnx_empty_domain(&g, flags, 1, s);
      std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
      hipMemcpy(div, h.data(), n * 4, hipMemcpyHostToDevice);
      for (int w = 0; w < 3; ++w) fnx_jacobi(&g, flags, div, p, nullptr, 0.f, iters, nullptr, ws, wsb, s);
      hipStreamSynchronize(s);
      const int reps = 50;
      hipEventRecord(e0, s);
      for (int r = 0; r < reps; ++r) fnx_jacobi(&g, flags, div, p, nullptr, 0.f, iters, nullptr, ws, wsb, s);
      hipEventRecord(e1, s); hipStreamSynchronize(s);
      hipEventElapsedTime(&ms, e0, e1);
      double us = ms * 1e3 / reps;
      printf("res=%4d iters=%3d: %8.1f us/solve  %6.2f us/sweep  %7.1f GB/s algorithmic (16 B/cell/sweep)\n", res, iters, us, us / iters,
             16.0 * n * iters / (us * 1e-6) / 1e9);
      hipFree(flags); hipFree(div); hipFree(p); hipFree(ws);
    }
  }
  return 0;
}
