// Practical fp32 MFMA peak on this GPU: independent v_mfma_f32_32x32x2_f32 / 16x16x4_f32 chains, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDE>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float r = 0.f;
  if (WIDE) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wide = 1; wide >= 0; --wide) {
    for (int blocks : {512, 2048}) {
      const int iters = 20000;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (wide) burn<1><<<blocks, 256>>>(out, iters); else burn<0><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double flops = (double)blocks * 4 * iters * (wide ? 8 * 2.0 * 32 * 32 * 2 : 16 * 2.0 * 16 * 16 * 4);
      printf("%s, %d blocks x 4 waves: %.1f ms -> %.1f TFLOP/s\n", wide ? "32x32x2 f32" : "16x16x4 f32", blocks, best, flops / best / 1e9);
    }
  }
  return 0;
}
