// How many cycles does the texture-addresser / L1 path charge per wave-level VMEM instruction, by access width?
// Every block streams over an L2-resident array (2 MiB, re-read `iters` times) with row-shaped accesses like the stencil
// kernels issue: lane <-> consecutive x.  Reports bytes/clk/CU = bytes / (time * clock * CUs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ta_bench.bin tools/ubench/ta_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int N_FLOATS = 512 * 1024;          // 2 MiB

template <int MODE>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ p, float* __restrict__ out, int iters, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 1024 + 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + w, nw = gridDim.x * 4;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {             // dword per lane: one 256-B row per instruction
      for (int q = 0; q < N_FLOATS / 64 - 1; ++q) { const int r = (q + wave * 97) % (N_FLOATS / 64 - 1); acc += p[r * 64 + lane + shift]; }
    } else if (MODE == 1) {      // dwordx2 per lane: 512 B per instruction
      const float2* q = (const float2*)(p + shift * 2);
      for (int q2 = 0; q2 < N_FLOATS / 128 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 128 - 1); float2 v = q[r * 64 + lane]; acc += v.x + v.y; }
    } else if (MODE == 2) {      // dwordx4 per lane: 1 KiB per instruction
      const float4* q = (const float4*)(p + shift * 4);
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1); float4 v = q[r * 64 + lane]; acc += (v.x + v.y) + (v.z + v.w); }
    } else if (MODE == 3) {      // dwordx4, the 64 lanes cover FOUR separate 256-B rows (lane/16 = row): what a wave-private
                                 // row loader would issue
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) {
        const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1);
        const int row = lane >> 4, c = lane & 15;
        float4 v; __builtin_memcpy(&v, p + ((size_t)((r * 4 + row) * 37 % (N_FLOATS / 64 - 1)) * 64) + c * 4 + shift * 2, 16);
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else if (MODE == 4) {      // the same through the LDS-DMA path + ds_read_b32 (lane <-> column again)
      float* my = lds + w * 1024;
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) {
        const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1);
        const int row = lane >> 4, c = lane & 15;
        const float* src = p + ((size_t)((r * 4 + row) * 37 % (N_FLOATS / 64 - 1)) * 64) + c * 4 + shift * 2;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(my), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0)
        acc += (my[lane] + my[64 + lane]) + (my[128 + lane] + my[192 + lane]);
      }
    } else if (MODE == 7) {      // LDS-DMA, two stages in flight (wait for the older one only)
      float* my = lds + w * 1024;
      const int row = lane >> 4, c = lane & 15;
      const int n = N_FLOATS / 256 - 1;
      {
        const int r = (0 + wave * 97) % n;
        __builtin_amdgcn_global_load_lds(p + ((size_t)((r * 4 + row) * 37 % (N_FLOATS / 64 - 1)) * 64) + c * 4 + shift * 2,
                                         (__attribute__((address_space(3))) void*)(my), 16, 0, 0);
      }
      for (int q2 = 1; q2 < n; ++q2) {
        const int r = (q2 + wave * 97) % n;
        float* dst = my + (q2 & 1) * 256;
        __builtin_amdgcn_global_load_lds(p + ((size_t)((r * 4 + row) * 37 % (N_FLOATS / 64 - 1)) * 64) + c * 4 + shift * 2,
                                         (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0071 | 0x3f00);          // vmcnt(1)
        const float* src = my + ((q2 - 1) & 1) * 256;
        acc += (src[lane] + src[64 + lane]) + (src[128 + lane] + src[192 + lane]);
      }
    } else if (MODE == 5) {      // dword gather: lane <-> x shifted by a per-lane 0/1 (interpolation-corner pattern)
      for (int q = 0; q < N_FLOATS / 64 - 2; ++q) { const int r = (q + wave * 97) % (N_FLOATS / 64 - 2); acc += p[r * 64 + lane + ((lane * 7 + r) & 1)]; }
    } else if (MODE == 8) {      // dword per lane at a 16-B lane stride (one component of a float4 row: 256 B useful out of 1 KiB of lines)
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1); acc += p[(r * 64 + lane) * 4 + 3 + shift * 4]; }
    } else if (MODE == 9) {      // dwordx2 per lane at a 16-B lane stride (half of a float4 row)
      const float2* q = (const float2*)(p + shift * 4);
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1); float2 v = q[(r * 64 + lane) * 2 + 1]; acc += v.x + v.y; }
    } else if (MODE == 10) {     // dword row stores
      float* o2 = const_cast<float*>(p);
      for (int q = 0; q < N_FLOATS / 64 - 1; ++q) { const int r = (q + wave * 97) % (N_FLOATS / 64 - 1); o2[r * 64 + lane + shift] = (float)q; }
    } else if (MODE == 11) {     // dwordx4 stores
      float4* o2 = (float4*)const_cast<float*>(p) + shift;
      for (int q2 = 0; q2 < N_FLOATS / 256 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 256 - 1); o2[r * 64 + lane] = make_float4((float)q2, 1.f, 2.f, 3.f); }
    } else if (MODE == 6) {      // ubyte per lane (mask rows)
      const unsigned char* q = (const unsigned char*)p;
      for (int q2 = 0; q2 < N_FLOATS / 64 - 1; ++q2) { const int r = (q2 + wave * 97) % (N_FLOATS / 64 - 1); acc += (float)q[r * 64 + lane]; }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
void run(const char* name, const float* d, float* o, int cus, double bytes_per_pass) {
  const int iters = 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int shift = 0; shift < 2; ++shift) {
    if (shift && MODE >= 5 && MODE < 8) break;
    rd<MODE><<<cus * 4, 256>>>(d, o, 2, shift);
    hipEventRecord(a);
    rd<MODE><<<cus * 4, 256>>>(d, o, iters, shift);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = bytes_per_pass * iters * (double)cus * 16;
    printf("%-44s shift %d: %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (at 2.4 GHz)\n", name, shift, ms, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 2.4e9 / cus);
  }
}

int main() {
  int dev = 0, cus = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  float *d, *o; hipMalloc(&d, N_FLOATS * 4 + 4096); hipMalloc(&o, 64); hipMemset(d, 0, N_FLOATS * 4 + 4096);
  const double full = (double)N_FLOATS * 4;
  run<0>("dword / lane (256 B per instr)", d, o, cus, full);
  run<1>("dwordx2 / lane (512 B per instr)", d, o, cus, full);
  run<2>("dwordx4 / lane (1 KiB per instr)", d, o, cus, full);
  run<3>("dwordx4, 4 separate rows per instr", d, o, cus, full);
  run<4>("global_load_lds_dwordx4 4 rows + ds_read", d, o, cus, full);
  run<7>("global_load_lds_dwordx4 4 rows, 2 in flight", d, o, cus, full);
  run<5>("dword gather (+0/+1 per lane)", d, o, cus, full);
  run<6>("ubyte / lane (64 B per instr)", d, o, cus, full / 4);
  run<8>("dword / lane, 16-B lane stride (useful bytes)", d, o, cus, full / 4);
  run<9>("dwordx2 / lane, 16-B lane stride (useful bytes)", d, o, cus, full / 2);
  run<10>("dword row stores", d, o, cus, full);
  run<11>("dwordx4 stores", d, o, cus, full);
  return 0;
}
