// Do VALU / LDS instructions of a wave run beside the matrix pipe?  Two waves per SIMD (one 512-thread workgroup per CU), each
// an endless chain of independent v_mfma_f32_32x32x2_f32 with K filler instructions behind every MFMA:
//   K v_add_f32 (independent registers) | K v_pk_add_f32 | K ds_read_b64 (conflict-free, waited for once per 8 MFMAs)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_coexec.hip -o tools/ubench/mfma_coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int K, bool MF>
__global__ __launch_bounds__(512, 2) void burn(float* out, int iters) {
  __shared__ float lds[24 * 1024];                    // 96 KB: one workgroup per CU
  for (int i = threadIdx.x; i < 24 * 1024; i += 512) lds[i] = i * 1e-6f;
  __syncthreads();
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, c = 1e-7f;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x[16]; f32x2 y[16]; f32x2 z[16];
  for (int j = 0; j < 16; ++j) { x[j] = j; y[j] = (f32x2){(float)j, 1.f}; z[j] = y[j]; }
  const f32x2 c2 = {c, c};
  const float* lp = lds + (threadIdx.x & 63) * 2 + (threadIdx.x >> 6) * 128;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)lp;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 q4[4] = {};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, (short)0, 256 * 512 * 4, 0x00020000);
  const unsigned goff = (blockIdx.x * 512 + threadIdx.x) * 4u, goff4 = (blockIdx.x * 512 + (threadIdx.x & ~3u)) * 4u;
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + 16384u * 4u + (threadIdx.x >> 6) * 1024u);
  int sreg = 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int j = (i * K + k) & 15;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[j]) : "v"(c2));
        if (KIND == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(z[j]) : "v"(la), "n"(0));
        if (KIND == 3) asm volatile("ds_write_b32 %0, %1" :: "v"(la), "v"(x[j]) : "memory");
        if (KIND == 4) asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(y[j]) : "memory");
        if (KIND == 5) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(x[j]) : "v"(goff), "s"(rs) : "memory");
        if (KIND == 6) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q4[j & 3]) : "v"(goff4), "s"(rs) : "memory");
        if (KIND == 7) asm volatile("ds_read2st64_b64 %0, %1 offset1:4" : "=v"(q4[j & 3]) : "v"(la));
        if (KIND == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
        if (KIND == 9) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[j]) : "v"(c));
        if (KIND == 10) asm volatile("s_mov_b32 m0, %2\n s_nop 0\n buffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(goff4), "s"(rs), "s"(m0v) : "memory");
        if (KIND == 11) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sreg));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 2 || KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 5 || KIND == 6 || KIND == 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float r = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  for (int j = 0; j < 16; ++j) r += x[j] + y[j].x + y[j].y + z[j].x + z[j].y;
  for (int j = 0; j < 4; ++j) r += q4[j].x + q4[j].w;
  r += sreg;
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND, int K, bool MF>
void run(float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    burn<KIND, K, MF><<<blocks, 512>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // per SIMD: 2 waves x 8 MFMA x iters; cycles at 2.4 GHz per MFMA slot of the SIMD
  const double cyc = best * 1e-3 * 2.4e9 / (2.0 * 8 * iters);
  printf("%-12s K=%2d %s: %7.2f ms  = %6.1f cycles (at 2.4 GHz) per MFMA slot of a SIMD", name, K, MF ? "with MFMA" : "no MFMA  ", best, cyc);
  if (MF) printf("  -> %.1f TFLOP/s", 256.0 * 8 * 8 * iters * 2.0 * 32 * 32 * 2 / best / 1e9);
  printf("\n");
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 0, true>(out, "none");
  run<0, 2, true>(out, "v_add_f32"); run<0, 4, true>(out, "v_add_f32"); run<0, 8, true>(out, "v_add_f32"); run<0, 16, true>(out, "v_add_f32");
  run<0, 8, false>(out, "v_add_f32"); run<0, 16, false>(out, "v_add_f32");
  run<1, 2, true>(out, "v_pk_add_f32"); run<1, 4, true>(out, "v_pk_add_f32"); run<1, 8, true>(out, "v_pk_add_f32");
  run<1, 8, false>(out, "v_pk_add_f32");
  run<2, 1, true>(out, "ds_read_b64"); run<2, 2, true>(out, "ds_read_b64"); run<2, 4, true>(out, "ds_read_b64");
  run<2, 4, false>(out, "ds_read_b64");
  run<7, 1, true>(out, "ds_read2st64"); run<7, 2, true>(out, "ds_read2st64"); run<7, 2, false>(out, "ds_read2st64");
  run<3, 1, true>(out, "ds_write_b32"); run<3, 4, true>(out, "ds_write_b32"); run<3, 4, false>(out, "ds_write_b32");
  run<4, 1, true>(out, "ds_write_b64"); run<4, 4, true>(out, "ds_write_b64");
  run<5, 1, true>(out, "buf_load_b32"); run<5, 2, true>(out, "buf_load_b32"); run<5, 2, false>(out, "buf_load_b32");
  run<6, 1, true>(out, "buf_load_b128"); run<6, 1, false>(out, "buf_load_b128");
  run<10, 1, true>(out, "buf_b128_lds"); run<10, 1, false>(out, "buf_b128_lds");
  run<8, 4, true>(out, "v_max_f32"); run<9, 4, true>(out, "v_max_i32");
  run<11, 8, true>(out, "s_mul_i32");
  return 0;
}
