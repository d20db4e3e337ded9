// Microbenchmark: cost of cross-lane primitives on gfx950 (cycles per wave-instruction, one wave per SIMD and 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void bench(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.5f, b = 1.f, c = 2.f, d = 3.f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { a = a + b; b = b + c; c = c + d; d = d + a; }                  // 4 dependent-ish adds
      if (MODE == 1) {  // wave_shr
        a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x138, 0xf, 0xf, true)) + b;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x130, 0xf, 0xf, true)) + c;
        c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138, 0xf, 0xf, true)) + d;
        d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x130, 0xf, 0xf, true)) + a;
      }
      if (MODE == 2) {  // row_shr:1
        a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x111, 0xf, 0xf, true)) + b;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x101, 0xf, 0xf, true)) + c;
        c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x111, 0xf, 0xf, true)) + d;
        d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x101, 0xf, 0xf, true)) + a;
      }
      if (MODE == 3) {  // ds_bpermute via __shfl
        a = __shfl_up(a, 1, 64) + b; b = __shfl_down(b, 1, 64) + c; c = __shfl_up(c, 1, 64) + d; d = __shfl_down(d, 1, 64) + a;
      }
      if (MODE == 4) {  // bfe + bfi
        int m = __builtin_amdgcn_sbfe(__builtin_bit_cast(int, a), u, 1);
        float r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(b), "v"(c)); a = r + d;
        m = __builtin_amdgcn_sbfe(__builtin_bit_cast(int, b), u, 1);
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(c), "v"(d)); b = r + a;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const char* names[] = {"v_add x4 (dep chain of 4)", "wave_shr/shl dpp + add", "row_shr/shl dpp + add", "__shfl (ds_bpermute) + add", "bfe+bfi+add"};
  for (int threads : {64, 256, 1024}) {
    for (int mode = 0; mode < 5; ++mode) {
      const int iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto run = [&]() {
        switch (mode) {
          case 0: bench<0><<<256, threads>>>(out, cyc, iters); break;
          case 1: bench<1><<<256, threads>>>(out, cyc, iters); break;
          case 2: bench<2><<<256, threads>>>(out, cyc, iters); break;
          case 3: bench<3><<<256, threads>>>(out, cyc, iters); break;
          case 4: bench<4><<<256, threads>>>(out, cyc, iters); break;
        }
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double groups = (double)iters * 16;   // per group: mode0: 4 ops; mode1-3: 4 x (xlane + add); mode4: 2 x (bfe,bfi,add)
      printf("threads/block=%4d  %-28s  %8.3f ms  %7.1f cyc/group (counter)  %.2f ns/group\n", threads, names[mode], ms, h / groups, ms * 1e6 / groups);
    }
  }
  return 0;
}
