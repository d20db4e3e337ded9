// Issue rate of packed fp32 VALU instructions on gfx950 against their scalar forms:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_bench.hip -o /tmp/pk_bench && /tmp/pk_bench
// Each kernel runs a loop of 8 independent chains x 32 instructions; waves per SIMD = 1, 2, 4; reports cycles per
// instruction per wave as seen by one SIMD (s_memtime deltas) and wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[8], z[8]; v2f y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { z[i] = b + i; x[i] = a + i + threadIdx.x; y[i] = v2f{a + i, b + threadIdx.x}; }
  const v2f bb = {b, a};
  unsigned long long cond = __builtin_amdgcn_ballot_w64(a > threadIdx.x), cond2 = 0;
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[0]), "v"(b) : "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(bb));
        if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(bb));
        if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(bb));
        if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
        if (MODE == 7) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 8) { int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(x[i])); asm volatile("" :: "s"(t)); }
        if (MODE == 9) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[i]), "v"(b) : "vcc");
        if (MODE == 11) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 13) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        if (MODE == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 15) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(y[i]) : "v"(bb));
        if (MODE == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "s"(cond));
        if (MODE == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x[i]) : "v"(a), "v"(b), "s"(cond));
        if (MODE == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
        if (MODE == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(b));
        if (MODE == 21) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x[i]));
        if (MODE == 22) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(cond2) : "v"(x[i]), "v"(b));
        if (MODE == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 24) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 25) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 26) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (MODE == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
        if (MODE == 29) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %1, %2, vcc\n\tv_cndmask_b32 %3, %3, %1, vcc\n\tv_cndmask_b32 %4, %1, %4, vcc" : "+v"(x[i]), "+v"(y[i].x), "+v"(y[i].y), "+v"(z[i]) : "v"(b) : "vcc");
        if (MODE == 30) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1\n\tv_cndmask_b32_e64 %0, %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, %1\n\tv_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(x[i]), "+s"(cond2) : "v"(b));
        if (MODE == 31) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %2, %2, %1" : "+v"(x[i]), "+v"(z[i]) : "v"(b));
        if (MODE == 27) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : );
      }
    }
  }
  float s = (float)cond2;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + z[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    const int blocks = 256 * wps;
    k<MODE><<<blocks, 256>>>(out, 10, 1.0f, 1.0001f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0f, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 32 * wps;   // instructions issued per SIMD
    printf("%-28s waves/SIMD %d: %7.3f ms  -> %5.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms,
           ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  run<0>("v_mul_f32", out); run<1>("v_pk_mul_f32", out); run<2>("v_add_f32", out); run<3>("v_pk_add_f32", out);
  run<4>("v_fma_f32", out); run<5>("v_pk_fma_f32", out); run<6>("v_cndmask_b32", out); run<7>("v_min3_f32", out);
  run<8>("v_readlane_b32", out); run<9>("v_fract_f32", out); run<10>("v_cmp_lt_f32", out); run<11>("v_cvt_i32_f32", out);
  run<12>("v_rcp_f32", out); run<13>("v_sqrt_f32", out); run<14>("v_mul_lo_u32", out); run<15>("v_pk_mul_f32 op_sel_hi", out);
  run<16>("v_cndmask e64 sgpr in-place", out); run<17>("v_cndmask e64 sgpr fresh dst", out); run<18>("v_cmp+v_cndmask vcc (pair)", out);
  run<19>("v_and_b32", out); run<20>("v_mov_b32", out); run<21>("v_bfe_u32", out); run<22>("v_cmp_lt_f32_e64 ->sgpr", out);
  run<23>("v_max_f32", out); run<24>("v_sub_f32", out); run<25>("v_add_u32", out); run<26>("v_mul_f32_e64", out);
  run<27>("v_cndmask vcc (vcc set once)", out);
  run<28>("cmp + 4 cndmask vcc same dst (x5)", out); run<29>("cmp + 4 cndmask vcc 4 dsts (x5)", out); run<30>("cmp_e64 + 4 cndmask sgpr (x5)", out);
  run<31>("cndmask vcc + v_add (x2)", out);
  return 0;
}
