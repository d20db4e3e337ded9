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

// ---------------------------------------------------------------------------------------------------
// 3D: z-marching sweep.  Flags are folded once per solve into a 7-bit neighbour mask per cell (1 B instead of seven
// 4-B flag reads per sweep).  A thread owns 4 consecutive rows of one column and marches along z keeping the planes
// z-1, z, z+1 of its rows in registers: per cell and sweep it loads 1 new p value (+0.5 for the row halos), div
// and the mask byte, x neighbours come from DPP lane shifts (only lanes 0/63 fetch theirs), and writes 1 value --
// ~13-15 B/cell of HBM traffic against 16 B/cell algorithmic.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned MZ_CONT = 1, MZ_L = 2, MZ_R = 4, MZ_D = 8, MZ_U = 16, MZ_B = 32, MZ_F = 64;
constexpr int ZR = 4;        // rows per thread
constexpr int ZCHUNK = 16;   // planes marched per thread

template <bool QUIRKS>
__global__ __launch_bounds__(256) void jacobi3d_mask_kernel(GridDims g, const float* __restrict__ flags,
                                                            unsigned char* __restrict__ mask) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int bk = blockIdx.z;
  const int b = bk / g.D, k = bk - b * g.D;
  if (i >= g.W || j >= g.H) return;
  const size_t o = (size_t)b * g.DHW + (size_t)k * g.HW + j * g.W + i;
  unsigned m = 0;
  if (!is_border<true>(g, i, j, k) && flags[o] != FNX_OBST) {
    m = MZ_CONT;
    if (flags[o - 1] == FNX_OBST) m |= MZ_L;
    if (flags[o + 1] == FNX_OBST) m |= MZ_R;
    if (flags[o - g.W] == FNX_OBST) m |= MZ_D;
    if (flags[o + g.W] == FNX_OBST) m |= MZ_U;
    if (!QUIRKS) {       // the reference applies no Neumann substitution in z (fluids_init.cpp:935-943)
      if (flags[o - g.HW] == FNX_OBST) m |= MZ_B;
      if (flags[o + g.HW] == FNX_OBST) m |= MZ_F;
    }
  }
  mask[o] = (unsigned char)m;
}

// first sweep from p = 0: ((((((0+0)+0)+0)+0)+0)+div)/6 == div/6 on 'cont' cells
__global__ __launch_bounds__(256) void jacobi3d_first_kernel(size_t n, int B, size_t per, const float* __restrict__ div,
                                                             const unsigned char* __restrict__ mask,
                                                             float* __restrict__ p_out, float* __restrict__ sumsq) {
  for (int b = 0; b < B; ++b) {
    float local = 0.f;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
      const size_t o = (size_t)b * per + q;
      float sum = 0.f + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f;
      const float v = (mask[o] & MZ_CONT) ? (sum + div[o]) / 6.f : 0.f;
      p_out[o] = v;
      local += v * v;
    }
    if (sumsq) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
      if ((threadIdx.x & 63) == 0) atomicAdd(&sumsq[b], local);
    }
  }
  (void)n;
}

__global__ __launch_bounds__(256) void jacobi3d_march_kernel(GridDims g, const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ div,
                                                             const float* __restrict__ p_in, float* __restrict__ p_out,
                                                             float* __restrict__ sumsq, int nzc, int kb, int ke) {
  const int lane = threadIdx.x;                          // blockDim = (64, 4): one wave per threadIdx.y
  const int i = blockIdx.x * 64 + lane;
  const int j0 = (blockIdx.y * 4 + threadIdx.y) * ZR;
  int bz = blockIdx.z;
  const int zc = bz % nzc; const int b = bz / nzc;
  const int k_lo = kb + zc * ZCHUNK, k_hi = min(k_lo + ZCHUNK, ke); // planes [k_lo, k_hi) of the requested range [kb, ke)
  const bool xin = i < g.W;
  const int ic = xin ? i : g.W - 1;
  const size_t base = (size_t)b * g.DHW;
  // row validity (rows beyond H are clamped for loads and never stored)
  int jr[ZR]; bool jin[ZR];
#pragma unroll
  for (int r = 0; r < ZR; ++r) { jin[r] = j0 + r < g.H; jr[r] = jin[r] ? j0 + r : g.H - 1; }
  const int jd = j0 > 0 ? j0 - 1 : 0, ju = j0 + ZR < g.H ? j0 + ZR : g.H - 1;   // halo rows (clamped: masks never select them at the border)
  const int il = ic > 0 ? ic - 1 : 0, ir = ic < g.W - 1 ? ic + 1 : g.W - 1;
  float pb[ZR], pc[ZR], pf[ZR];                         // planes k-1, k, k+1 of my rows
  auto ld = [&](int k, int j, int x) { return p_in[base + (size_t)k * g.HW + (size_t)j * g.W + x]; };
  auto clampk = [&](int k) { return k < 0 ? 0 : (k > g.D - 1 ? g.D - 1 : k); };
  // everything plane k needs besides its own rows; loaded one plane ahead of its use (software pipeline)
  struct Aux { float dv[ZR]; unsigned mk[ZR]; float hd, hu, el[ZR], er[ZR]; };
  auto load_aux = [&](int k, Aux& a) {
    const size_t ok = base + (size_t)k * g.HW;
#pragma unroll
    for (int r = 0; r < ZR; ++r) {
      a.dv[r] = div[ok + (size_t)jr[r] * g.W + ic];
      a.mk[r] = mask[ok + (size_t)jr[r] * g.W + ic];
      a.el[r] = 0.f; a.er[r] = 0.f;
      if (lane == 0) a.el[r] = ld(k, jr[r], il);         // x neighbours of the wave's edge lanes
      if (lane == 63) a.er[r] = ld(k, jr[r], ir);
    }
    a.hd = ld(k, jd, ic); a.hu = ld(k, ju, ic);           // row halos
  };
#pragma unroll
  for (int r = 0; r < ZR; ++r) { pb[r] = ld(clampk(k_lo - 1), jr[r], ic); pc[r] = ld(k_lo, jr[r], ic); pf[r] = ld(clampk(k_lo + 1), jr[r], ic); }
  Aux cur, nxt;
  load_aux(k_lo, cur);
  float local = 0.f;
  for (int k = k_lo; k < k_hi; ++k) {
    // issue the loads of plane k+1 (aux) and k+2 (own rows) before touching plane k
    float pn[ZR];
    const int k2 = clampk(k + 2), k1 = clampk(k + 1);
#pragma unroll
    for (int r = 0; r < ZR; ++r) pn[r] = ld(k2, jr[r], ic);
    load_aux(k1, nxt);
    const size_t ok = base + (size_t)k * g.HW;
#pragma unroll
    for (int r = 0; r < ZR; ++r) {
      const float c = pc[r];
      float xl = dpp_from_left(c), xr = dpp_from_right(c);
      if (lane == 0) xl = cur.el[r];
      if (lane == 63) xr = cur.er[r];
      const float yd = r == 0 ? cur.hd : pc[r > 0 ? r - 1 : 0];
      const float yu = r == ZR - 1 ? cur.hu : pc[r < ZR - 1 ? r + 1 : 0];
      const unsigned m = cur.mk[r];
      const float n1 = (m & MZ_L) ? c : xl;
      const float n2 = (m & MZ_R) ? c : xr;
      const float n3 = (m & MZ_D) ? c : yd;
      const float n4 = (m & MZ_U) ? c : yu;
      const float n5 = (m & MZ_B) ? c : pb[r];
      const float n6 = (m & MZ_F) ? c : pf[r];
      float sum = n1 + n2;
      sum = sum + n3;
      sum = sum + n4;
      sum = sum + n5;
      sum = sum + n6;
      float v = (sum + cur.dv[r]) / 6.f;
      v = (m & MZ_CONT) ? v : 0.f;
      if (xin && jin[r]) {
        p_out[ok + (size_t)jr[r] * g.W + i] = v;
        const float d = v - c;
        local += d * d;
      }
    }
#pragma unroll
    for (int r = 0; r < ZR; ++r) { pb[r] = pc[r]; pc[r] = pf[r]; pf[r] = pn[r]; }
    cur = nxt;
  }
  if (sumsq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) atomicAdd(&sumsq[b], local);
  }
}

// ---------------------------------------------------------------------------------------------------
// 3D, two sweeps per pass (temporal blocking along the z-march).  The thread keeps p^0 on rows j0-2..j0+5 and
// p^1 on rows j0-1..j0+4 for three planes each; at step t it builds p^1(plane t) and, from p^1(t-2..t), the
// finished p^2(plane t-1) for its 4 rows.  The extra halo rows / the two extra planes per chunk / the wave's two
// edge columns are recomputed instead of exchanged: HBM traffic per sweep drops from ~20 B/cell to ~10 B/cell.
// Wave tile: 60 output columns (lanes 2..61; lanes 0,1,62,63 are halo columns, so no edge loads) x 4 rows x Z2C planes.
// ---------------------------------------------------------------------------------------------------
constexpr int Z2R = 4, Z2C = 16;

__global__ __launch_bounds__(256, 4) void jacobi3d_march2_kernel(GridDims g, const unsigned char* __restrict__ mask,
                                                              const float* __restrict__ div,
                                                              const float* __restrict__ p_in,
                                                              float* __restrict__ p_out, float* __restrict__ sumsq,
                                                              int nzc, int kb, int ke) {
  constexpr int R0 = Z2R + 4, R1 = Z2R + 2;
  const int lane = threadIdx.x;
  const int x = blockIdx.x * 60 - 2 + lane;
  const int j0 = (blockIdx.y * 4 + threadIdx.y) * Z2R;
  int bz = blockIdx.z;
  const int zc = bz % nzc; const int b = bz / nzc;
  const int k_lo = kb + zc * Z2C, k_hi = min(k_lo + Z2C, ke);      // output planes of the requested range [kb, ke)
  const bool xin = (x >= 0) & (x < g.W);
  const int xc = x < 0 ? 0 : (x > g.W - 1 ? g.W - 1 : x);
  const size_t base = (size_t)b * g.DHW;
  int jc0[R0];                                           // clamped row of p^0 row slot rr (j = j0-2+rr)
#pragma unroll
  for (int rr = 0; rr < R0; ++rr) { const int j = j0 - 2 + rr; jc0[rr] = j < 0 ? 0 : (j > g.H - 1 ? g.H - 1 : j); }
  unsigned rin1 = 0;                                     // bit rr: p^1 row slot rr (j = j0-1+rr) lies inside the grid
#pragma unroll
  for (int rr = 0; rr < R1; ++rr) { const int j = j0 - 1 + rr; rin1 |= (unsigned)((j >= 0) & (j < g.H)) << rr; }
  auto clampk = [&](int k) { return k < 0 ? 0 : (k > g.D - 1 ? g.D - 1 : k); };
  auto ldrow = [&](int k, int rr, int xx) { return p_in[base + (size_t)clampk(k) * g.HW + (size_t)jc0[rr] * g.W + xx]; };
  struct Aux { float dv[R1]; unsigned mk[R1]; };
  auto load_aux = [&](int k, Aux& a) {                   // everything sweep 1 of plane k needs besides p^0 itself
    const bool kin = (k >= 0) & (k < g.D);
    const size_t ok = base + (size_t)clampk(k) * g.HW;
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) {
      const size_t o = ok + (size_t)jc0[rr + 1] * g.W + xc;
      a.dv[rr] = div[o];
      const bool in = kin & xin & (((rin1 >> rr) & 1) != 0);
      a.mk[rr] = in ? (unsigned)mask[o] : 0u;            // cells outside the grid are never 'cont'
    }
  };
  auto relax = [](unsigned m, float c, float xl, float xr, float yd, float yu, float zb, float zf, float dv) {
    const float n1 = (m & MZ_L) ? c : xl;
    const float n2 = (m & MZ_R) ? c : xr;
    const float n3 = (m & MZ_D) ? c : yd;
    const float n4 = (m & MZ_U) ? c : yu;
    const float n5 = (m & MZ_B) ? c : zb;
    const float n6 = (m & MZ_F) ? c : zf;
    float sum = n1 + n2;
    sum = sum + n3;
    sum = sum + n4;
    sum = sum + n5;
    sum = sum + n6;
    const float v = (sum + dv) / 6.f;
    return (m & MZ_CONT) ? v : 0.f;
  };

  float p0m[R0], p0c[R0], p0p[R0], p0n[R0];              // p^0 planes t-1, t, t+1, (prefetch) t+2
  float p1m[R1], p1c[R1], p1n[R1];                       // p^1 planes t-2, t-1, t
  float dprev[Z2R]; unsigned mprev[Z2R];                 // div / mask of plane t-1 on the output rows
  int t = k_lo - 1;
#pragma unroll
  for (int rr = 0; rr < R0; ++rr) { p0m[rr] = ldrow(t - 1, rr, xc); p0c[rr] = ldrow(t, rr, xc); p0p[rr] = ldrow(t + 1, rr, xc); }
#pragma unroll
  for (int rr = 0; rr < R1; ++rr) { p1m[rr] = 0.f; p1c[rr] = 0.f; }
#pragma unroll
  for (int r = 0; r < Z2R; ++r) { dprev[r] = 0.f; mprev[r] = 0u; }
  Aux cur, nxt;
  load_aux(t, cur);
  float local = 0.f;
  const bool lane_out = (lane >= 2) & (lane <= 61) & xin;
  for (; t <= k_hi; ++t) {
    // prefetch: p^0 plane t+2 and the aux data of plane t+1
#pragma unroll
    for (int rr = 0; rr < R0; ++rr) p0n[rr] = ldrow(t + 2, rr, xc);
    load_aux(t + 1, nxt);
    // sweep 1 on plane t, rows j0-1 .. j0+4
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) {
      const float c = p0c[rr + 1];
      const float xl = dpp_from_left(c), xr = dpp_from_right(c);     // lanes 0/63 get garbage: they are halo columns
      p1n[rr] = relax(cur.mk[rr], c, xl, xr, p0c[rr], p0c[rr + 2], p0m[rr + 1], p0p[rr + 1], cur.dv[rr]);
    }
    // sweep 2 on plane t-1, rows j0 .. j0+3 (needs p^1 of planes t-2, t-1, t)
    if (t - 1 >= k_lo) {
      const size_t ok = base + (size_t)(t - 1) * g.HW;
#pragma unroll
      for (int r = 0; r < Z2R; ++r) {
        const float c = p1c[r + 1];
        const float xl = dpp_from_left(c), xr = dpp_from_right(c);
        const float v = relax(mprev[r], c, xl, xr, p1c[r], p1c[r + 2], p1m[r + 1], p1n[r + 1], dprev[r]);
        if (lane_out && j0 + r < g.H) {
          p_out[ok + (size_t)(j0 + r) * g.W + x] = v;
          const float d = v - c;
          local += d * d;
        }
      }
    }
    // rotate the pipelines
#pragma unroll
    for (int rr = 0; rr < R0; ++rr) { p0m[rr] = p0c[rr]; p0c[rr] = p0p[rr]; p0p[rr] = p0n[rr]; }
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) { p1m[rr] = p1c[rr]; p1c[rr] = p1n[rr]; }
#pragma unroll
    for (int r = 0; r < Z2R; ++r) { dprev[r] = cur.dv[r + 1]; mprev[r] = cur.mk[r + 1]; }
    cur = nxt;
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

// 3D fast path (mask precomputed by launch_jacobi3d_mask)
void launch_jacobi3d_mask(const GridDims& g, bool quirks, const float* flags, unsigned char* mask, hipStream_t s) {
  const dim3 grid((g.W + 63) / 64, (g.H + 3) / 4, g.B * g.D), block(64, 4);
  if (quirks) jacobi3d_mask_kernel<true><<<grid, block, 0, s>>>(g, flags, mask);
  else jacobi3d_mask_kernel<false><<<grid, block, 0, s>>>(g, flags, mask);
}

// two sweeps in one pass: p_in = p^n, p_out = p^{n+2}; sumsq receives ||p^{n+2} - p^{n+1}||^2
void launch_jacobi3d_x2(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                        float* sumsq, hipStream_t s, int kb, int ke) {
  if (ke <= kb) { kb = 0; ke = g.D; }
  const int nzc = (ke - kb + Z2C - 1) / Z2C;
  const dim3 grid((g.W + 59) / 60, (g.H + 4 * Z2R - 1) / (4 * Z2R), g.B * nzc), block(64, 4);
  jacobi3d_march2_kernel<<<grid, block, 0, s>>>(g, mask, div, p_in, p_out, sumsq, nzc, kb, ke);
}

void launch_jacobi3d(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                     bool from_zero, float* sumsq, hipStream_t s, int kb, int ke) {
  if (ke <= kb) { kb = 0; ke = g.D; }
  if (from_zero) {
    size_t nb = ((size_t)g.DHW + 256 * 4 - 1) / (256 * 4);
    if (nb > 4096) nb = 4096;
    jacobi3d_first_kernel<<<(unsigned)nb, 256, 0, s>>>((size_t)g.B * g.DHW, g.B, (size_t)g.DHW, div, mask, p_out, sumsq);
    return;
  }
  const int nzc = (ke - kb + ZCHUNK - 1) / ZCHUNK;
  const dim3 grid((g.W + 63) / 64, (g.H + 4 * ZR - 1) / (4 * ZR), g.B * nzc), block(64, 4);
  jacobi3d_march_kernel<<<grid, block, 0, s>>>(g, mask, div, p_in, p_out, sumsq, nzc, kb, ke);
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
