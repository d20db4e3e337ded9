// Exhaustive check (all 2^32 bit patterns) of cheap exact replacements for x / 6.0f on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/div6_test.hip -o /tmp/div6_test && /tmp/div6_test
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float div6_a(float x) {          // mul + 2 fma: exact for 2^-125 <= |x| < inf and +0
  const float zh = 0x1.555556p-3f;
  const float q1 = x * zh;
  const float r = __builtin_fmaf(-6.0f, q1, x);
  return __builtin_fmaf(r, zh, q1);
}
__device__ __forceinline__ float div6_b(float x) {          // + v_div_fixup: -0, inf, nan
  return __builtin_amdgcn_div_fixupf(div6_a(x), 6.0f, x);
}
__device__ __forceinline__ float div6_c(float x) {          // v_div_scale + mul/fma + v_div_fmas + v_div_fixup
  const float zh = 0x1.555556p-3f;
  bool vcc;
  const float xs = __builtin_amdgcn_div_scalef(x, 6.0f, true, &vcc);
  const float q1 = xs * zh;
  const float r = __builtin_fmaf(-6.0f, q1, xs);
  const float q2 = __builtin_amdgcn_div_fmasf(r, zh, q1, vcc);
  return __builtin_amdgcn_div_fixupf(q2, 6.0f, x);
}

__global__ void check(unsigned long long* bad, uint32_t* example) {
  const uint32_t u0 = (blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long na = 0, nb = 0, nc = 0, nbd = 0, nct = 0;
  for (uint32_t hi = 0; hi < 16; ++hi) {
    const uint32_t u = u0 | (hi << 28);
    const float x = __uint_as_float(u);
    if (x != x) continue;
    const uint32_t t = __float_as_uint(x / 6.0f);
    if (__float_as_uint(div6_a(x)) != t) ++na;
    const float b = div6_b(x);
    if (__float_as_uint(b) != t) {
      ++nb;
      // would the "result is denormal" guard catch it?
      const uint32_t e = __float_as_uint(b) & 0x7f800000u;
      if (!(e == 0 && (__float_as_uint(b) & 0x7fffffu) != 0)) { ++nbd; example[0] = u; }
    }
    if (__float_as_uint(div6_c(x)) != t) { ++nc; example[1] = u; if ((u & 0x7fffffffu) < 0x20000000u) ++nct; }   // nct: |x| < 2^-63
  }
  if (na) atomicAdd(&bad[0], na);
  if (nb) atomicAdd(&bad[1], nb);
  if (nc) atomicAdd(&bad[2], nc);
  if (nbd) atomicAdd(&bad[3], nbd);
  if (nct) atomicAdd(&bad[4], nct);
}

int main() {
  unsigned long long* bad; uint32_t* ex;
  hipMalloc(&bad, 64); hipMemset(bad, 0, 64);
  hipMalloc(&ex, 8); hipMemset(ex, 0, 8);
  check<<<(1u << 28) / 256, 256>>>(bad, ex);
  unsigned long long h[8]; uint32_t he[2];
  hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 8, hipMemcpyDeviceToHost);
  printf("mismatches vs x/6.0f over all non-NaN floats: a(mul,fma,fma)=%llu  b(+div_fixup)=%llu  c(div_scale..div_fmas,div_fixup)=%llu\n",
         h[0], h[1], h[2]);
  printf("c mismatches with |x| < 2^-63 (the range whose quotient can be denormal): %llu\n", h[4]);
  printf("b mismatches NOT flagged by 'result is denormal': %llu (example x bits %08x); c example %08x\n", h[3], he[0], he[1]);
  return 0;
}
