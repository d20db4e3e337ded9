// Exhaustive check (all 2^32 bit patterns) of x / 6.0f as ONE multiplication and ONE fma with a two-word constant
// (Brisebarre & Muller, "Correctly rounded multiplication by arbitrary precision constants"): q = fma(x, ch, x * cl),
// ch + cl ~ 1/6.  Two splits: ch = RN(1/6) (cl < 0: inf * cl = -inf, the fma gives NaN) and ch = RD(1/6) (cl > 0).
// Reports, per split, the mismatches against x / 6.0f and how many of them a guard |q| < 2^-96 would NOT catch.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/div6_two_op_test.hip -o tools/ubench/div6_two_op_test.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int V>
__device__ __forceinline__ float div6_two(float x) {
  const float ch = V == 0 ? 0x1.555556p-3f : 0x1.555554p-3f;
  const float cl = V == 0 ? -0x1.555556p-28f : 0x1.555556p-27f;
  const float t = x * cl;
  return __builtin_fmaf(x, ch, t);
}

__global__ void check(unsigned long long* bad, uint32_t* example) {
  const uint32_t u0 = (blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long n[2] = {0, 0}, ng[2] = {0, 0}, ninf[2] = {0, 0};
  for (uint32_t hi = 0; hi < 16; ++hi) {
    const uint32_t u = u0 | (hi << 28);
    const float x = __uint_as_float(u);
    if (x != x) continue;
    const uint32_t t = __float_as_uint(x / 6.0f);
    const float q[2] = {div6_two<0>(x), div6_two<1>(x)};
#pragma unroll
    for (int v = 0; v < 2; ++v)
      if (__float_as_uint(q[v]) != t) {
        ++n[v];
        const bool guarded = !(fabsf(q[v]) >= 0x1p-96f);       // (true for NaN too)
        if (!guarded) { ++ng[v]; example[v] = u; }
        if ((u & 0x7fffffffu) == 0x7f800000u) ++ninf[v];
        else atomicMax(&example[2 + v], u & 0x7fffffffu);      // the largest finite |x| that differs
      }
  }
  for (int v = 0; v < 2; ++v) {
    if (n[v]) atomicAdd(&bad[3 * v], n[v]);
    if (ng[v]) atomicAdd(&bad[3 * v + 1], ng[v]);
    if (ninf[v]) atomicAdd(&bad[3 * v + 2], ninf[v]);
  }
}

int main() {
  unsigned long long* bad; uint32_t* ex;
  hipMalloc(&bad, 64); hipMemset(bad, 0, 64);
  hipMalloc(&ex, 16); hipMemset(ex, 0, 16);
  check<<<(1u << 28) / 256, 256>>>(bad, ex);
  unsigned long long h[8]; uint32_t he[4];
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
  const char* name[2] = {"ch = RN(1/6), cl < 0", "ch = RD(1/6), cl > 0"};
  for (int v = 0; v < 2; ++v)
    printf("%s: mismatches vs x/6.0f over all non-NaN floats %llu, of them with |q| >= 2^-96 (not caught by the guard) %llu "
           "(example x bits %08x), +-inf among the mismatches %llu; largest finite |x| that differs: bits %08x = %g\n", name[v], h[3 * v],
           h[3 * v + 1], he[v], h[3 * v + 2], he[2 + v], *(float*)&he[2 + v]);
  return 0;
}
