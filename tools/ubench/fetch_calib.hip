// Calibration of rocprofv3's FETCH_SIZE on gfx950 by access width: every kernel reads the same 1 GiB (footprint beyond
// the 256 MiB Infinity Cache) exactly once, coalesced, with 4 / 8 / 16 bytes per lane, through global loads, through
// buffer loads with out-of-range checking (the Winograd kernel's halo fetch) and through LDS DMA (its weight stream).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib.bin tools/ubench/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o r --output-format csv -- tools/ubench/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ __launch_bounds__(256) void rd_global(const T* __restrict__ p, float* o, size_t n) {
  float a = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = p[i];
    const float* f = (const float*)&v;
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) a += f[k];
  }
  if (a == 1234.5f) o[0] = a;
}
typedef __amdgpu_buffer_rsrc_t Rsrc;
__global__ __launch_bounds__(256) void rd_buffer_b32(const float* __restrict__ p, float* o, unsigned n_per_block) {
  // each block reads its own contiguous segment through a buffer resource, 4 bytes per lane per instruction
  const float* base = p + (size_t)blockIdx.x * n_per_block;
  const Rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)(n_per_block * 4), 0x00020000);
  float a = 0.f;
  for (unsigned i = threadIdx.x; i < n_per_block; i += 256)
    a += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, i * 4, 0, 0));
  if (a == 1234.5f) o[0] = a;
}
__global__ __launch_bounds__(256) void rd_lds_dma(const float* __restrict__ p, float* o, unsigned n_per_block) {
  __shared__ __attribute__((aligned(16))) float buf[4][256];            // one KiB per wave-instruction
  const float* base = p + (size_t)blockIdx.x * n_per_block;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float a = 0.f;
  for (unsigned i = 0; i < n_per_block; i += 1024) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i + w * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(&buf[w][0]), 16, 0, 0);
    __syncthreads();
    a += buf[w][lane];
    __syncthreads();
  }
  if (a == 1234.5f) o[0] = a;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  float *a, *o;
  hipMalloc(&a, bytes); hipMalloc(&o, 64); hipMemset(a, 0, bytes);
  const unsigned nblk = 8192, per = (unsigned)(bytes / 4 / nblk);
  for (int rep = 0; rep < 3; ++rep) {
    rd_global<float><<<2048, 256>>>(a, o, bytes / 4);
    rd_global<float2><<<2048, 256>>>((const float2*)a, o, bytes / 8);
    rd_global<float4><<<2048, 256>>>((const float4*)a, o, bytes / 16);
    rd_buffer_b32<<<nblk, 256>>>(a, o, per);
    rd_lds_dma<<<nblk, 256>>>(a, o, per);
  }
  hipDeviceSynchronize();
  printf("each kernel read %zu MiB once\n", bytes >> 20);
  return 0;
}
