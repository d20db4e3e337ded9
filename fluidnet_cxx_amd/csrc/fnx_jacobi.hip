// Jacobi pressure solve for gfx950 -- replaces solveLinearSystemJacobi (cpp/fluids_init.cpp:809-1004),
// which issues ~120 ATen ops plus a host sync per sweep.
//
// 2D: temporal blocking in REGISTERS.  A 1024^2 field is 4 MiB, so one sweep per launch would be launch-latency
// bound (2 us of HBM time per sweep vs ~2 us per kernel boundary).  Each wavefront instead loads a
// 64 x (32+2K) halo tile of p and div into VGPRs once, runs K sweeps with DPP lane shifts for the x
// neighbours (no LDS, no barrier) and writes back its (64-2K) x 32 centre: HBM traffic per K sweeps is
// ~1 read + 1 write of the field instead of K, and a 28-sweep solve is 4 launches.
// 3D: one sweep per launch, z-marching with coalesced 256-B rows (HBM-bound at 16 B/cell/sweep).
//
// Arithmetic per cell is exactly the reference's: ((((((n1+n2)+n3)+n4)+n5)+n6)+div)/denom, with
// obstacle neighbours replaced by the centre value (Neumann) and border cells held at 0 (Dirichlet).
#include "fnx_device.h"
#include "fnx_kernels.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------
// 2D: register-resident temporal blocking, one independent tile per wavefront.
//   lane  <-> one grid column (64 columns per wave, OX = 64-2K of them are output)
//   regs  <-> V = OY+2K rows of p and div per lane (OY = 32 output rows)
//   x neighbours: DPP wave_shr:1 / wave_shl:1 (one VALU op, no LDS); y neighbours: the adjacent registers.
// K sweeps run with no LDS, no barrier and no memory traffic; ring s of the tile goes stale at sweep s and the
// 2K-wide halo is simply recomputed by the neighbouring waves (redundancy 64*V/(OX*OY): 1.43x at K=4, 2.0x at K=8).
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ float dpp_from_left(float v) {    // value held by lane-1 (0 into lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {   // value held by lane+1 (0 into lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned dpp_from_left_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned dpp_from_right_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

__device__ __forceinline__ float bfi_blend(int m, float a, float b) {   // m ? a : b for m in {0,-1}
  // The empty asm hides that m is a sign-extended bit, otherwise LLVM rewrites the blend into
  // shift+compare+cndmask (5 ops); as written it selects the single v_bfi_b32.
  asm("" : "+v"(m));
  return __builtin_bit_cast(float, (__builtin_bit_cast(int, a) & m) | (__builtin_bit_cast(int, b) & ~m));
}

// One Jacobi update of rows [R0, R0+N) of the register tile, the N rows advanced in LOCKSTEP: a single wave
// issues a dependent VALU op only every ~4.5 cycles but independent ones every ~2.3 (tools/ubench), and hipcc
// does not interleave the per-row chains on its own.  MASKED = obstacle-aware path.
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void jacobi_rows(float (&p)[V], const float (&d)[V], float& carry, const unsigned (&mL)[2],
                                            const unsigned (&mR)[2], const unsigned (&mD)[2], const unsigned (&mU)[2],
                                            const unsigned (&mC)[2], float (&delta)[N]) {
  float pc[N], pl[N], pr[N], sum[N], v[N];
#pragma unroll
  for (int n = 0; n < N; ++n) pc[n] = p[R0 + n];
  const float up_last = (R0 + N < V) ? p[(R0 + N < V) ? R0 + N : 0] : 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) pl[n] = dpp_from_left(pc[n]);
#pragma unroll
  for (int n = 0; n < N; ++n) pr[n] = dpp_from_right(pc[n]);
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n) pl[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mL[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], pl[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) pr[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mR[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], pr[n]);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = pl[n] + pr[n];
  float dn[N], un[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { dn[n] = n == 0 ? carry : pc[n > 0 ? n - 1 : 0]; un[n] = n == N - 1 ? up_last : pc[n < N - 1 ? n + 1 : 0]; }
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n) dn[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mD[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], dn[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) un[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mU[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], un[n]);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + dn[n];
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + un[n];
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + d[R0 + n];
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = sum[n] / 4.f;
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n)
      v[n] = __builtin_bit_cast(float, __builtin_bit_cast(int, v[n]) & __builtin_amdgcn_sbfe((int)mC[(R0 + n) >> 5], (R0 + n) & 31, 1));
  }
  carry = pc[N - 1];
#pragma unroll
  for (int n = 0; n < N; ++n) { delta[n] = v[n] - pc[n]; p[R0 + n] = v[n]; }
}

template <int V, int R0, int N, bool MASKED, int K>
__device__ __forceinline__ void jacobi_sweep_rows(float (&p)[V], const float (&d)[V], float& carry,
                                                  const unsigned (&mL)[2], const unsigned (&mR)[2],
                                                  const unsigned (&mD)[2], const unsigned (&mU)[2],
                                                  const unsigned (&mC)[2], bool last, bool lane_ok, int rows_in_grid,
                                                  float& local) {
  if constexpr (R0 < V) {
    constexpr int M = (V - R0 >= N) ? N : (V - R0);
    float delta[M];
    jacobi_rows<V, R0, M, MASKED>(p, d, carry, mL, mR, mD, mU, mC, delta);
    if (last) {                                          // wave-uniform
#pragma unroll
      for (int n = 0; n < M; ++n)
        if (R0 + n >= K && R0 + n < V - K && lane_ok && R0 + n < rows_in_grid) local += delta[n] * delta[n];
    }
    jacobi_sweep_rows<V, R0 + M, N, MASKED, K>(p, d, carry, mL, mR, mD, mU, mC, last, lane_ok, rows_in_grid, local);
  }
}

template <int K, int OY>
__global__ __launch_bounds__(256) void jacobi2d_reg_kernel(GridDims g, const float* __restrict__ flags,
                                                           const float* __restrict__ div,
                                                           const float* __restrict__ p_in, float* __restrict__ p_out,
                                                           int from_zero, float* __restrict__ sumsq, int tiles_x,
                                                           int tiles_y) {
  constexpr int V = OY + 2 * K, OX = 64 - 2 * K;
  constexpr int NI = 4;                                  // rows advanced in lockstep
  const int lane = threadIdx.x & 63;
  // the tile index is wave-uniform: say so, and all row addressing below becomes scalar work
  const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int b = blockIdx.y;
  if (tile >= tiles_x * tiles_y) return;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x = tx * OX - K + lane, y0 = ty * OY - K;
  const bool xin = (x >= 0) & (x < g.W), xint = (x >= 1) & (x <= g.W - 2);
  const size_t base = (size_t)b * g.DHW;
  const int xc = x < 0 ? 0 : (x > g.W - 1 ? g.W - 1 : x);      // clamped column: loads are unconditional, then masked

  float p[V], d[V];
  unsigned long long ob = 0, cont = 0;
#pragma unroll
  for (int r = 0; r < V; ++r) {
    const int y = y0 + r;                                        // scalar
    const bool yin = (y >= 0) & (y < g.H);
    const int yc = y < 0 ? 0 : (y > g.H - 1 ? g.H - 1 : y);
    const size_t row = base + (size_t)yc * g.W;                  // scalar
    const float f = (flags + row)[xc];
    const float dv = (div + row)[xc];
    float pv = 0.f;
    if (!from_zero) pv = (p_in + row)[xc];
    const bool in = xin & yin;
    d[r] = in ? dv : 0.f;
    p[r] = in ? pv : 0.f;
    const bool isob = !in | (f == FNX_OBST);
    ob |= (unsigned long long)isob << r;
    cont |= (unsigned long long)(xint & (y >= 1) & (y <= g.H - 2) & !isob) << r;
  }
  // obstacle masks of the four neighbours; the outermost ring of the tile is never evaluated
  const unsigned long long obL = ((unsigned long long)dpp_from_left_u((unsigned)(ob >> 32)) << 32) | dpp_from_left_u((unsigned)ob);
  const unsigned long long obR = ((unsigned long long)dpp_from_right_u((unsigned)(ob >> 32)) << 32) | dpp_from_right_u((unsigned)ob);
  const unsigned long long obD = ob << 1, obU = ob >> 1;
  cont &= ~(1ull | (1ull << (V - 1)));
  if (lane == 0 || lane == 63) cont = 0;

  unsigned mL[2] = {(unsigned)obL, (unsigned)(obL >> 32)}, mR[2] = {(unsigned)obR, (unsigned)(obR >> 32)};
  unsigned mD[2] = {(unsigned)obD, (unsigned)(obD >> 32)}, mU[2] = {(unsigned)obU, (unsigned)(obU >> 32)};
  unsigned mC[2] = {(unsigned)cont, (unsigned)(cont >> 32)};
  const bool lane_ok = (lane >= K) & (lane < 64 - K) & xin;
  const int rows_in_grid = g.H - y0;                     // rows r < rows_in_grid lie inside the grid
  float local = 0.f;
  // Fast path (wave-uniform): every evaluated cell of this tile is a plain fluid cell with no obstacle neighbour
  // (all tiles away from walls and obstacles) -> the selects disappear: 2 DPP moves + 7 VALU ops per cell.
  constexpr unsigned long long RING = ~(1ull | (1ull << (V - 1))) & ((V < 64) ? ((1ull << V) - 1) : ~0ull);
  const bool edge_lane = (lane == 0) | (lane == 63);
  const bool plain = edge_lane | ((cont == RING) & (((obL | obR | obD | obU) & RING) == 0));
  if (__all(plain)) {
#pragma unroll 1
    for (int s = 0; s < K; ++s) {
      const bool last = (s == K - 1) && (sumsq != nullptr);
      float carry = 0.f;
      jacobi_sweep_rows<V, 0, NI, false, K>(p, d, carry, mL, mR, mD, mU, mC, last, lane_ok, rows_in_grid, local);
    }
  } else {
#pragma unroll 1
    for (int s = 0; s < K; ++s) {
      // keeps the compiler from hoisting 5*V bit tests out of the sweep loop into (spilled) SGPR pairs
      asm volatile("" : "+v"(mL[0]), "+v"(mL[1]), "+v"(mR[0]), "+v"(mR[1]), "+v"(mD[0]), "+v"(mD[1]), "+v"(mU[0]),
                   "+v"(mU[1]), "+v"(mC[0]), "+v"(mC[1]));
      const bool last = (s == K - 1) && (sumsq != nullptr);
      float carry = 0.f;
      jacobi_sweep_rows<V, 0, NI, true, K>(p, d, carry, mL, mR, mD, mU, mC, last, lane_ok, rows_in_grid, local);
    }
  }
  if (lane_ok) {
#pragma unroll
    for (int r = K; r < V - K; ++r) {
      const int y = y0 + r;
      if (y < g.H) (p_out + base + (size_t)y * g.W)[x] = p[r];
    }
  }
  if (sumsq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) atomicAdd(&sumsq[b], local);
  }
}

// Generic single sweep (3D, and any 2D shape): one thread per cell.
constexpr int BX = 64, BY = 4;

template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void jacobi_sweep_kernel(GridDims g, const float* __restrict__ flags,
                                                              const float* __restrict__ div,
                                                              const float* __restrict__ p_in,
                                                              float* __restrict__ p_out, bool from_zero,
                                                              float* __restrict__ sumsq) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int bk = blockIdx.z;
  const int b = IS3D ? bk / g.D : bk, k = IS3D ? bk - b * g.D : 0;
  float d2 = 0.f;
  if (i < g.W && j < g.H) {
    const size_t o = (size_t)b * g.DHW + (size_t)k * g.HW + j * g.W + i;
    float v = 0.f;
    const float pc = from_zero ? 0.f : p_in[o];
    if (!is_border<IS3D>(g, i, j, k) && flags[o] != FNX_OBST) {
      if (from_zero) {
        float sum = 0.f + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f;
        v = (sum + div[o]) / (IS3D ? 6.f : 4.f);
      } else {
        const float n1 = flags[o - 1] == FNX_OBST ? pc : p_in[o - 1];
        const float n2 = flags[o + 1] == FNX_OBST ? pc : p_in[o + 1];
        const float n3 = flags[o - g.W] == FNX_OBST ? pc : p_in[o - g.W];
        const float n4 = flags[o + g.W] == FNX_OBST ? pc : p_in[o + g.W];
        float sum = n1 + n2;
        sum = sum + n3;
        sum = sum + n4;
        if (IS3D) {
          // reference applies no Neumann substitution in z (fluids_init.cpp:935-943) -> QUIRKS
          const float n5 = (!QUIRKS && flags[o - g.HW] == FNX_OBST) ? pc : p_in[o - g.HW];
          const float n6 = (!QUIRKS && flags[o + g.HW] == FNX_OBST) ? pc : p_in[o + g.HW];
          sum = sum + n5;
          sum = sum + n6;
        } else {
          sum = sum + 0.f;
          sum = sum + 0.f;
        }
        v = (sum + div[o]) / (IS3D ? 6.f : 4.f);
      }
    }
    p_out[o] = v;
    const float d = v - pc;
    d2 = d * d;
  }
  if (sumsq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d2 += __shfl_down(d2, off, 64);
    if (threadIdx.x == 0) atomicAdd(&sumsq[b], d2);      // blockDim.x == 64: one wave per row of the block
  }
}

__global__ void residual_finish_kernel(int B, const float* __restrict__ sumsq, float* __restrict__ res) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) m = fmaxf(m, sqrtf(sumsq[b]));
    *res = m;
  }
}

__global__ __launch_bounds__(256) void residual_kernel(GridDims g, const float* __restrict__ a,
                                                       const float* __restrict__ bq, float* __restrict__ sumsq) {
  const int b = blockIdx.y;
  float acc = 0.f;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < (size_t)g.DHW; q += (size_t)gridDim.x * 256) {
    const float d = a[(size_t)b * g.DHW + q] - bq[(size_t)b * g.DHW + q];
    acc += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sumsq[b], acc);
}

template <int K, int OY>
void launch_reg_oy(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out,
                   bool from_zero, float* sumsq, hipStream_t s) {
  constexpr int OX = 64 - 2 * K;
  const int tiles_x = (g.W + OX - 1) / OX, tiles_y = (g.H + OY - 1) / OY;
  const dim3 grid((tiles_x * tiles_y + 3) / 4, g.B);
  jacobi2d_reg_kernel<K, OY><<<grid, 256, 0, s>>>(g, flags, div, p_in, p_out, from_zero ? 1 : 0, sumsq, tiles_x, tiles_y);
}

// Tile height selection.  Small grids are latency-bound by the serial instruction chain of ONE wave
// (~K*V*20 VALU ops, no other wave on the SIMD to hide behind), large grids by total VALU work, which grows with
// the halo redundancy 64*V/(OX*OY).  So: short tiles when there are few waves, tall tiles when there are many.
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <int K>
void launch_reg(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out,
                bool from_zero, float* sumsq, hipStream_t s) {
  constexpr int OX = 64 - 2 * K;
  static const int forced = env_int("FNX_JACOBI_OY", 0);
  int oy = forced;
  if (!oy) {
    // measured on MI355X (tools/ubench/jacobi_bench.cpp): <= 2 Mcells: 8-row tiles; up to 8 Mcells: 16; else 32
    const long cells = (long)g.W * g.H * g.B;
    oy = cells <= (2l << 20) ? 8 : (cells <= (8l << 20) ? 16 : 32);
  }
  if (oy == 32) launch_reg_oy<K, 32>(g, flags, div, p_in, p_out, from_zero, sumsq, s);
  else if (oy == 16) launch_reg_oy<K, 16>(g, flags, div, p_in, p_out, from_zero, sumsq, s);
  else launch_reg_oy<K, 8>(g, flags, div, p_in, p_out, from_zero, sumsq, s);
}

}  // namespace

namespace fnx {

constexpr int KMAX_2D = 8;

int jacobi_max_sweeps_per_launch(const GridDims& g, bool is3d) {
  if (is3d || g.D != 1) return 1;
  static const int forced = env_int("FNX_JACOBI_K", 0);
  if (forced >= 1 && forced <= KMAX_2D) return forced;
  // small grids are bound by the serial chain of one wave (K*(OY+2K) row updates): fewer sweeps per launch win
  return (long)g.W * g.H * g.B <= (2l << 20) ? 4 : KMAX_2D;
}

// nsweeps in [1, jacobi_max_sweeps_per_launch]; sumsq (B floats, pre-zeroed) receives ||p_n - p_{n-1}||^2 of the
// LAST sweep of this launch when non-null.
void launch_jacobi(const GridDims& g, bool is3d, bool quirks, const float* flags, const float* div, const float* p_in,
                   float* p_out, int nsweeps, bool from_zero, float* sumsq, hipStream_t s) {
  if (!is3d && g.D == 1 && nsweeps > 1 && nsweeps <= KMAX_2D) {
    switch (nsweeps) {
      case 2: launch_reg<2>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 3: launch_reg<3>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 4: launch_reg<4>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 5: launch_reg<5>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 6: launch_reg<6>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 7: launch_reg<7>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 8: launch_reg<8>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
    }
  }
  // generic path: exactly one sweep
  const dim3 grid((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B * g.D), block(BX, BY);
  if (is3d) {
    if (quirks) jacobi_sweep_kernel<true, true><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
    else jacobi_sweep_kernel<true, false><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
  } else {
    jacobi_sweep_kernel<false, false><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
  }
}

void launch_residual_finish(int B, const float* sumsq, float* res, hipStream_t s) {
  residual_finish_kernel<<<1, 64, 0, s>>>(B, sumsq, res);
}

void launch_residual(const GridDims& g, const float* a, const float* b, float* sumsq, float* res, hipStream_t s) {
  int blocks = (g.DHW + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  residual_kernel<<<dim3(blocks, g.B), 256, 0, s>>>(g, a, b, sumsq);
  residual_finish_kernel<<<1, 64, 0, s>>>(g.B, sumsq, res);
}

}  // namespace fnx
