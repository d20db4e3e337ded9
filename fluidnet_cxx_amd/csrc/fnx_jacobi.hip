// Jacobi pressure solve for gfx950 -- replaces solveLinearSystemJacobi (cpp/fluids_init.cpp:809-1004),
// which issues ~120 ATen ops plus a host sync per sweep.
//
// 2D: temporal blocking in REGISTERS.  A 1024^2 field is 4 MiB, so one sweep per launch would be launch-latency
// bound (2 us of HBM time per sweep vs ~2 us per kernel boundary).  A workgroup of 8 (4) waves instead holds one
// 64 x 64 (64 x 32) tile of p and div in VGPRs, runs K = 7-8 sweeps with DPP lane shifts for the x neighbours and one
// LDS row hand-over between the waves per sweep, and writes back the tile's centre: HBM traffic per K sweeps is ~1 read
// + 1 write of the field instead of K, and a 28-sweep solve is 4 launches.
// 3D: two sweeps per pass, z-marching with coalesced 256-B rows (16-byte row quads between the passes of a solve).
//
// Arithmetic per cell is exactly the reference's: ((((((n1+n2)+n3)+n4)+n5)+n6)+div)/denom, with
// obstacle neighbours replaced by the centre value (Neumann) and border cells held at 0 (Dirichlet).
#include "fnx_device.h"
#include "fnx_kernels.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------
// 2D: register-resident temporal blocking.
//   lane  <-> one grid column (64 columns per wave, OX = 64-2K of them are output)
//   regs  <-> RW rows of p and div per lane
//   x neighbours: DPP wave_shr:1 / wave_shl:1 (one VALU op, no LDS); y neighbours: the adjacent registers.
// Ring s of a tile goes stale at sweep s and the 2K-wide halo is recomputed by the neighbouring tiles.
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
                                            const unsigned (&mC)[2], float (&delta)[N], float top = 0.f) {
  float pc[N], pl[N], pr[N], sum[N], v[N];
#pragma unroll
  for (int n = 0; n < N; ++n) pc[n] = p[R0 + n];
  const float up_last = (R0 + N < V) ? p[(R0 + N < V) ? R0 + N : 0] : top;
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
  // the reference adds the two missing z neighbours as zeros: ((s + 0) + 0) == (s + 0) for every s (the first addition
  // already turns a -0 into +0, NaN and inf pass through), so ONE addition reproduces the bits of both
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

// The obstacle-aware update with the five masks of a row as LANE MASKS (one bool per row and mask: an SGPR pair each, the select a
// single v_cndmask) instead of bits of a per-lane word (v_bfe + v_bfi per select): for the 4-row waves of the deep launches, where
// 20 masks fit the scalar registers -- small grids run the masked path on every tile that touches the domain wall (85 -> 57 VALU
// instructions per sweep of a wave's four rows, ~45 with the unused select groups skipped; the plain path has 28).  Same operations on the same operands: same bits.
// anyL .. anyU (wave-uniform): does any live cell of the wave take that substitution at all?  A wave beside the left wall only has
// left-neighbour substitutions: the other three groups of selects are skipped by scalar branches.
template <int V> struct LaneMasks { bool L[V], R[V], D[V], U[V], C[V]; bool anyL, anyR, anyD, anyU; };
template <int V, int R0, int N>
__device__ __forceinline__ void jacobi_rows_lm(float (&p)[V], const float (&d)[V], float& carry, const LaneMasks<V>& m, float top) {
  float pc[N], pl[N], pr[N], sum[N], v[N];
#pragma unroll
  for (int n = 0; n < N; ++n) pc[n] = p[R0 + n];
  const float up_last = (R0 + N < V) ? p[(R0 + N < V) ? R0 + N : 0] : top;
#pragma unroll
  for (int n = 0; n < N; ++n) pl[n] = dpp_from_left(pc[n]);
#pragma unroll
  for (int n = 0; n < N; ++n) pr[n] = dpp_from_right(pc[n]);
  if (m.anyL) {
#pragma unroll
    for (int n = 0; n < N; ++n) pl[n] = m.L[R0 + n] ? pc[n] : pl[n];
  }
  if (m.anyR) {
#pragma unroll
    for (int n = 0; n < N; ++n) pr[n] = m.R[R0 + n] ? pc[n] : pr[n];
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = pl[n] + pr[n];
  float dn[N], un[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { dn[n] = n == 0 ? carry : pc[n > 0 ? n - 1 : 0]; un[n] = n == N - 1 ? up_last : pc[n < N - 1 ? n + 1 : 0]; }
  if (m.anyD) {
#pragma unroll
    for (int n = 0; n < N; ++n) dn[n] = m.D[R0 + n] ? pc[n] : dn[n];
  }
  if (m.anyU) {
#pragma unroll
    for (int n = 0; n < N; ++n) un[n] = m.U[R0 + n] ? pc[n] : un[n];
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + dn[n];
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + un[n];
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + 0.f;           // (see jacobi_rows)
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + d[R0 + n];
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = sum[n] / 4.f;
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = m.C[R0 + n] ? v[n] : 0.f;
  carry = pc[N - 1];
#pragma unroll
  for (int n = 0; n < N; ++n) p[R0 + n] = v[n];
}

// ---------------------------------------------------------------------------------------------------
// 2D, workgroup tiles: NW waves stacked in y form ONE tile of 64 x (NW*RW) cells and hand each other their edge rows
// through LDS once per sweep (one barrier per sweep, two LDS row images alternating), so only the workgroup's outer ring
// is recomputed halo: the work per K sweeps is 64*NW*RW/((64-2K)*(NW*RW-2K)) times the field (1.8x at K = 8 with 8 waves
// of 8 rows; one independent 64 x (16+2K) tile per wave, the round-1 kernel, did 2.7x), and a wave's serial chain per
// sweep is RW rows whatever K.
// ---------------------------------------------------------------------------------------------------
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void wg_sweep_rows(float (&p)[V], const float (&d)[V], float& carry, float top,
                                              const unsigned (&mL)[2], const unsigned (&mR)[2], const unsigned (&mD)[2],
                                              const unsigned (&mU)[2], const unsigned (&mC)[2]) {
  if constexpr (R0 < V) {
    constexpr int M = (V - R0 >= N) ? N : (V - R0);
    float delta[M];
    jacobi_rows<V, R0, M, MASKED>(p, d, carry, mL, mR, mD, mU, mC, delta, top);
    wg_sweep_rows<V, R0 + M, N, MASKED>(p, d, carry, top, mL, mR, mD, mU, mC);
  }
}

// K (sweeps of the launch = halo width) is a launch argument: it only enters the tile origin, the output window and the trip
// count.  Small grids run DEEP launches (K up to 28: a 128^2 solve of 28 sweeps is ONE launch of 256 tiles with an 8 x 8
// output window each, one per CU, instead of four launches of 7 sweeps: the grid is launch-latency bound, the recomputed halo
// costs idle CUs nothing).
// the same, skipping row groups that lie wholly in the tile's stale rings (rows < s + 1 or > rows_total - 2 - s at sweep s)
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void wg_sweep_live(float (&p)[V], const float (&d)[V], float& carry, float top,
                                              const unsigned (&mL)[2], const unsigned (&mR)[2], const unsigned (&mD)[2],
                                              const unsigned (&mU)[2], const unsigned (&mC)[2], int tr0, int s, int rows_total) {
  if constexpr (R0 < V) {
    constexpr int M = (V - R0 >= N) ? N : (V - R0);
    const bool stale = (tr0 + R0 + M - 1 < s + 1) | (tr0 + R0 > rows_total - 2 - s);
    if (stale) {
      carry = p[R0 + M - 1];                               // what the next group sees below it: this group's last row, unchanged
    } else {
      float delta[M];
      jacobi_rows<V, R0, M, MASKED>(p, d, carry, mL, mR, mD, mU, mC, delta, top);
    }
    wg_sweep_live<V, R0 + M, N, MASKED>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, rows_total);
  }
}

template <int RW, int NW>
__global__ __launch_bounds__(64 * NW) void jacobi2d_wg_kernel(GridDims g, const float* __restrict__ flags,
                                                             const float* __restrict__ div, const float* __restrict__ p_in,
                                                             float* __restrict__ p_out, int from_zero, int tiles_x, int K) {
  constexpr int V = RW;
  const int OX = 64 - 2 * K, OYW = NW * RW - 2 * K;
  constexpr int NI = 4;
  static_assert(V <= 32, "tile shape");
  __shared__ float edge[2][2][NW][64];                   // [sweep parity][0: first row, 1: last row][wave][lane]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int b = blockIdx.y;
  const int x = tx * OX - K + lane, y0 = ty * OYW - K + w * RW;      // this wave's first row
  const bool xin = (x >= 0) & (x < g.W), xint = (x >= 1) & (x <= g.W - 2);
  const size_t base = (size_t)b * g.DHW;
  const int xc = x < 0 ? 0 : (x > g.W - 1 ? g.W - 1 : x);

  float p[V], d[V];
  unsigned ob = 0, cont = 0;
  auto is_ob = [&](int y) {                              // obstacle (or outside the grid) at (x, y)
    const bool yin = (y >= 0) & (y < g.H);
    const int yc = y < 0 ? 0 : (y > g.H - 1 ? g.H - 1 : y);
    const float f = (flags + base + (size_t)yc * g.W)[xc];
    return !(xin & yin) | (f == FNX_OBST);
  };
#pragma unroll
  for (int r = 0; r < V; ++r) {
    const int y = y0 + r;
    const bool yin = (y >= 0) & (y < g.H);
    const int yc = y < 0 ? 0 : (y > g.H - 1 ? g.H - 1 : y);
    const size_t row = base + (size_t)yc * g.W;
    const float f = (flags + row)[xc];
    const float dv = (div + row)[xc];
    float pv = 0.f;
    if (!from_zero) pv = (p_in + row)[xc];
    const bool in = xin & yin;
    d[r] = in ? dv : 0.f;
    p[r] = in ? pv : 0.f;
    const bool isob = !in | (f == FNX_OBST);
    ob |= (unsigned)isob << r;
    cont |= (unsigned)(xint & (y >= 1) & (y <= g.H - 2) & !isob) << r;
  }
  const unsigned ob_below = is_ob(y0 - 1), ob_above = is_ob(y0 + V);
  const unsigned obL = dpp_from_left_u(ob), obR = dpp_from_right_u(ob);
  const unsigned obD = (ob << 1) | ob_below, obU = (ob >> 1) | (ob_above << (V - 1));
  // the workgroup's outermost ring is never evaluated (its neighbours are not in the tile)
  if (w == 0) cont &= ~1u;
  if (w == NW - 1) cont &= ~(1u << (V - 1));
  if (lane == 0 || lane == 63) cont = 0;
  unsigned mL[2] = {obL, 0}, mR[2] = {obR, 0}, mD[2] = {obD, 0}, mU[2] = {obU, 0}, mC[2] = {cont, 0};
  LaneMasks<(V <= 4 ? V : 1)> lm;
  if constexpr (V <= 4) {
#pragma unroll
    for (int r = 0; r < V; ++r) {
      lm.L[r] = (obL >> r) & 1u; lm.R[r] = (obR >> r) & 1u; lm.D[r] = (obD >> r) & 1u; lm.U[r] = (obU >> r) & 1u; lm.C[r] = (cont >> r) & 1u;
    }
    // (a cell that is not updated -- cont clear: its value is forced to 0 -- does not care which neighbours it would have read)
    lm.anyL = __any((obL & cont) != 0); lm.anyR = __any((obR & cont) != 0);
    lm.anyD = __any((obD & cont) != 0); lm.anyU = __any((obU & cont) != 0);
  }
  // output rows of this wave: tile rows [K, NW*RW - K) that lie in the grid
  int out_lo = K - w * RW, out_hi = NW * RW - K - w * RW;
  if (out_lo < 0) out_lo = 0;
  if (out_hi > V) out_hi = V;
  if (out_hi > g.H - y0) out_hi = g.H - y0;
  const bool lane_ok = (lane >= K) & (lane < 64 - K) & xin;
  constexpr unsigned ALL = V < 32 ? ((1u << V) - 1) : ~0u;
  const unsigned ring = ALL & ~((w == 0 ? 1u : 0u) | (w == NW - 1 ? 1u << (V - 1) : 0u));
  const bool edge_lane = (lane == 0) | (lane == 63);
  const bool plain = edge_lane | ((cont == ring) & (((obL | obR | obD | obU) & ring) == 0));
  const bool all_plain = __all(plain);
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    float* e = &edge[s & 1][0][0][0];
    e[w * 64 + lane] = p[0];
    e[NW * 64 + w * 64 + lane] = p[V - 1];
    __syncthreads();
    float carry = w > 0 ? e[NW * 64 + (w - 1) * 64 + lane] : 0.f;           // last row of the wave below
    const float top = w < NW - 1 ? e[(w + 1) * 64 + lane] : 0.f;            // first row of the wave above
    // Sweep s + 1 makes the tile's ring s stale (its neighbours are not in the tile); nothing inside the output window ever
    // reads a stale cell, so a group of rows that lies wholly in rings 0..s is not evaluated (wave-uniform): a deep launch
    // (K = 28) skips 40 % of its row updates, a K = 8 launch the outer groups of its first and last wave.
    const int tr0 = w * RW;                                // first tile row of this wave
    if (all_plain) wg_sweep_live<V, 0, NI, false>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, NW * RW);
    else if constexpr (V <= 4) {
      // (a wave of <= 4 rows is one row group: stale exactly when wg_sweep_live would skip it)
      const bool stale = (tr0 + V - 1 < s + 1) | (tr0 > NW * RW - 2 - s);
      if (!stale) jacobi_rows_lm<V, 0, V>(p, d, carry, lm, top);
    } else {
      asm volatile("" : "+v"(mL[0]), "+v"(mR[0]), "+v"(mD[0]), "+v"(mU[0]), "+v"(mC[0]));
      wg_sweep_live<V, 0, NI, true>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, NW * RW);
    }
  }
  if (lane_ok) {
#pragma unroll
    for (int r = 0; r < V; ++r)
      if (r >= out_lo && r < out_hi) (p_out + base + (size_t)(y0 + r) * g.W)[x] = p[r];
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

// The mask bytes in row groups of four for the two-sweep march: maskq[b][k][q][i] = bytes of rows 4q .. 4q+3 of column i
// (0 = "not a fluid cell" for rows >= H).  1 B per cell like the byte mask, built from it once per mask.
__global__ __launch_bounds__(256) void jacobi3d_maskq_kernel(GridDims g, const unsigned char* __restrict__ mask,
                                                             unsigned* __restrict__ maskq) {
  const int i = blockIdx.x * 64 + threadIdx.x, q = blockIdx.y * 4 + threadIdx.y;
  const int Hq = (g.H + 3) >> 2;
  if (i >= g.W || q >= Hq) return;
  const size_t bk = blockIdx.z;                              // b * D + k
  const unsigned char* m = mask + bk * g.HW + i;
  unsigned w = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 4 * q + r;
    if (j < g.H) w |= (unsigned)m[(size_t)j * g.W] << (8 * r);
  }
  maskq[(bk * Hq + q) * g.W + i] = w;
}

// first sweep from p = 0: ((((((0+0)+0)+0)+0)+0)+div)/6 == div/6 on 'cont' cells.  Writes the planes [kb, ke) of every
// sample only (`first` = kb*HW, `count` = (ke-kb)*HW cells per sample; `per` = cells per sample).
__global__ __launch_bounds__(256) void jacobi3d_first_kernel(int B, size_t per, size_t first, size_t count,
                                                             const float* __restrict__ div,
                                                             const unsigned char* __restrict__ mask,
                                                             float* __restrict__ p_out) {
  for (int b = 0; b < B; ++b) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < count; q += (size_t)gridDim.x * 256) {
      const size_t o = (size_t)b * per + first + q;
      float sum = 0.f + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f;
      const float v = (mask[o] & MZ_CONT) ? (sum + div[o]) / 6.f : 0.f;
      p_out[o] = v;
    }
  }
}

__global__ __launch_bounds__(256) void jacobi3d_march_kernel(GridDims g, const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ div,
                                                             const float* __restrict__ p_in, float* __restrict__ p_out,
                                                             int nzc, int kb, int ke) {
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
  // halo rows, clamped into the grid (masks never select them at the border).  A wave whose rows ALL lie beyond H (the last block of
  // four waves hangs over the grid when H is not a multiple of 4 ZR) stores nothing but still loads: its lower halo row j0 - 1 is
  // beyond the grid too and must be clamped like the rows themselves -- unclamped it read up to a plane past the tensor on the last
  // plane (a fault wherever the tensor ends its memory segment; round 6, found under rocgdb)
  const int jd = j0 > 0 ? (j0 - 1 < g.H ? j0 - 1 : g.H - 1) : 0, ju = j0 + ZR < g.H ? j0 + ZR : g.H - 1;
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
      if (xin && jin[r]) p_out[ok + (size_t)jr[r] * g.W + i] = v;
    }
#pragma unroll
    for (int r = 0; r < ZR; ++r) { pb[r] = pc[r]; pc[r] = pf[r]; pf[r] = pn[r]; }
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------
// 3D, two sweeps per pass (temporal blocking along the z-march).
//
// A wave (one 64-thread block) owns 60 output columns (lanes 2..61; lanes 0,1,62,63 are halo columns) x Z2R rows and
// marches along z.  It keeps p^0 on rows j0-2..j0+Z2R+1 and p^1 on rows j0-1..j0+Z2R for three planes each; at step
// t it builds p^1(plane t) and, from p^1(t-2..t), the finished p^2(plane t-1) of its rows.  Waves are independent (no
// LDS, no barrier): the halo rows / columns a wave re-reads are its neighbours' own rows, and the launch puts
// neighbouring tiles on the same XCD marching the same planes at the same time, so those re-reads hit L2 and HBM
// traffic stays near one read of p, div, mask and one write of p per pass.  (An earlier version handed the halo rows
// between y-stacked waves through LDS with one barrier per plane: 8-10 % slower -- the barrier couples four SIMDs.)
// The plane rings are rotated by unrolling the march 4x (no register moves); all row / plane addressing is
// wave-uniform (buffer loads: per-lane voffset = column, SGPR soffset = plane + row).
//
// VALU budget per cell-update: the obstacle-free path (wave-uniform test per plane) is 6 adds + the /6 + the 'cont'
// blend; with obstacles each neighbour costs a v_bfe_i32 + v_bfi_b32 more.
// Measured alternatives at 16.7M cells (us per pass): Z2R=4 at 4 waves/SIMD 48-52; Z2R=3 at 5 waves/SIMD 49-53;
// Z2R=2 at 6 waves/SIMD 53-59; Z2R=8 at 2 waves/SIMD 59-61; a two-step-deep load pipeline (loads into the ring slots
// that die mid-step) 48.6-51.4, i.e. no change; a third select-free path for all-'cont' planes 49.5-53.4.  At 255 MB
// of HBM traffic per pass the kernel moves 5.3 TB/s: it sits on the HBM bound of a 2-sweep pass, and more sweeps per
// pass in registers would pay (R+2K)/R redundant rows.
// ---------------------------------------------------------------------------------------------------
constexpr int Z2R = 4, Z2NW = 1;
constexpr int Z2WPS = 4;                   // waves per SIMD the register budget is sized for
constexpr int Z2C_DEFAULT = 16;

// x / 6.0f, correctly rounded, without the v_rcp/v_div_scale expansion: q = x*zh, r = x - 6q (exact), q + r*zh with
// zh = RN(1/6), then v_div_fixup for -0/inf/nan.  Checked against x / 6.0f for all 2^32 inputs on gfx950
// (tools/ubench/div6_test.hip): the only differing inputs give a denormal quotient (v_cmp_class 0x90), which the
// caller sends through div6_tiny.
__device__ __forceinline__ float div6_fast(float x) {
  const float zh = 0x1.555556p-3f;
  const float q1 = x * zh;
  const float r = __builtin_fmaf(-6.0f, q1, x);
  return __builtin_amdgcn_div_fixupf(__builtin_fmaf(r, zh, q1), 6.0f, x);
}

// The denormal-quotient case of div6_fast, exact in 5 instructions: v_div_scale lifts the numerator out of the denormal
// range, the same two-step refinement runs on the scaled value, v_div_fmas undoes the scale inside the last rounding.
// Checked against x / 6.0f for every |x| < 2^-63 (tools/ubench/div6_test.hip, variant c: 0 mismatches; for large |x|
// the hardware scheme would scale the denominator instead, which a constant reciprocal cannot follow).
__device__ __forceinline__ float div6_tiny(float x) {
  const float zh = 0x1.555556p-3f;
  bool vcc;
  const float n = __builtin_amdgcn_div_scalef(x, 6.0f, true, &vcc);
  const float q1 = n * zh;
  const float r = __builtin_fmaf(-6.0f, q1, n);
  return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r, zh, q1, vcc), 6.0f, x);
}

// SEL: 0 = no cell of the wave's rows has an obstacle neighbour, 1 = x neighbours only (every tile at an x wall of an
// otherwise empty domain), 2 = any
template <int SEL>
__device__ __forceinline__ float relax3(unsigned m, int sh, float c, float xl, float xr, float yd, float yu, float zb,
                                        float zf, float dv, float& num) {
  const int mi = (int)m;                                // the cell's mask byte sits at bit `sh` of m
  if (SEL >= 1) {                                       // Neumann: an obstacle neighbour is replaced by the centre
    xl = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 1, 1), c, xl);
    xr = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 2, 1), c, xr);
  }
  if (SEL >= 2) {
    yd = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 3, 1), c, yd);
    yu = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 4, 1), c, yu);
    zb = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 5, 1), c, zb);
    zf = bfi_blend(__builtin_amdgcn_sbfe(mi, sh + 6, 1), c, zf);
  }
  float sum = xl + xr;
  sum = sum + yd;
  sum = sum + yu;
  sum = sum + zb;
  sum = sum + zf;
  num = sum + dv;
  return div6_fast(num);
}

template <int N> struct IC { static constexpr int value = N; };

typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

// ZERO: p_in is all zeros (first pass of a solve): no p^0 loads, no p^0 halo exchange.
// LAY: bit 0 = p_in, bit 1 = p_out is in the ROW-QUAD layout p[b][k][j/4][i][j%4] (H % 4 == 0) instead of p[b][k][j][i].
// The pass is bound by the CU's vector-memory pipeline, whose cost is per instruction (docs/history/design_rounds_1-4.md, section 4): in the quad layout a
// wave's own four rows of a column are one 16-byte access and the two halo rows on either side one 8-byte access each, so a
// step reads p^0 with 3 instructions instead of 8 and writes its four finished rows with 1 instead of 4.  A plane is the same
// H*W floats in both layouts (ghost-plane exchanges do not care); the passes of a solve hand the quad layout to each other
// and only the last one writes rows.
// MIR: the launch ALSO stores the finished planes [mir.k[r], mir.k[r] + mir.n) of plane range r (0: [kb, ke), 1: the second range) to
// mir.out[r][q] + sample * mir.bstride, q = (*mir.sel[r] + 1) & 1 read on the device when the launch starts (a double-buffered
// destination whose turn a device counter tells; sel NULL: q = 0) -- the same values, the same layout within a plane.  The z-slab driver's last edge part of a sweep
// block writes the planes its neighbours need next straight into their mapped mailboxes (peer-store communicator) as the march
// finishes them: no transport launch copies them afterwards.  The extra store is issued by every step and dropped by the buffer range
// check outside the mirrored planes (an offset below the range wraps around, one above it is out of range): no branch.
struct MirrorArgs { float* out[2][2]; const unsigned* sel[2]; int k[2]; int n; unsigned long long bstride; unsigned long long* clock; };
template <bool ZERO, bool SPLIT, int LAY, bool MIR = false>
__global__ __launch_bounds__(64 * Z2NW, Z2WPS) void jacobi3d_march2_kernel(GridDims g, const unsigned* __restrict__ maskq,
                                                                      const float* __restrict__ div,
                                                                      const float* __restrict__ p_in,
                                                                      float* __restrict__ p_out, int nxt, int nyt,
                                                                      int zchunk, int kb, int ke, int kb2, MirrorArgs mir = MirrorArgs{}) {
  constexpr int R0 = Z2R + 4, R1 = Z2R + 2;
  const int lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (MIR && mir.clock && blockIdx.x == 0 && lane == 0) *mir.clock = wall_clock64();     // when the mirrored stores begin (FnxPlaneMirror.start_clock)
  // One resident set of waves.  !SPLIT: every tile is cut into the same plane chunks and a wave takes one
  // (tile, chunk); all waves of a chunk start together and march the same planes at the same pace, so the halo
  // columns/rows two neighbouring tiles both touch are fetched from HBM once and hit in L2 the second time.  Workgroup
  // ids go round-robin over the 8 XCDs; renumbering gives XCD q the tiles [q*G/8, (q+1)*G/8), i.e. whole bands of
  // neighbouring tiles.  SPLIT (more tiles than resident waves): the (tile, plane) space is cut into gridDim.x equal
  // contiguous ranges instead.
  const int G = gridDim.x, np = ke - kb;
  const int gid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int ntiles = nxt * nyt * g.B;
  int L0, L1;
  int kbase = kb;                                        // kb2 >= 0: a second plane range [kb2, kb2 + np) in the same launch
  if (!SPLIT) {
    int zc = gid / ntiles;
    const int tl = gid - zc * ntiles;
    if (kb2 >= 0) {                                      // chunks [0, nzc1) belong to the first range, [nzc1, 2 nzc1) to the second
      const int nzc1 = (np + zchunk - 1) / zchunk;
      if (zc >= nzc1) { zc -= nzc1; kbase = kb2; }
    }
    L0 = tl * np + zc * zchunk;
    L1 = min(L0 + zchunk, (tl + 1) * np);
    if (zc * zchunk >= np) L1 = L0;                      // padding block
  } else {
    const long long T = (long long)ntiles * np;
    auto cut = [&](int q) {                              // range boundary, snapped away from 1-2 plane slivers at tile ends
      int L = (int)(T * q / G);
      const int pk = L % np;
      if (pk < 3) L -= pk; else if (pk > np - 3) L += np - pk;
      return L;
    };
    L0 = cut(gid); L1 = cut(gid + 1);
  }
  for (; L0 < L1;) {                                     // one pass unless SPLIT
  const int tile = L0 / np, pk = L0 - tile * np;
  const int seg = min(np - pk, L1 - L0);
  L0 = SPLIT ? L0 + seg : L1;
  const int bx = tile % nxt, l1 = tile / nxt;
  const int by = l1 % nyt, b = l1 / nyt;
  const int x = bx * 60 - 2 + lane;
  const int j0 = (by * Z2NW + w) * Z2R;
  const int k_lo = kbase + pk, k_hi = k_lo + seg;           // output planes [k_lo, k_hi) of this segment
  const bool xin = (x >= 0) & (x < g.W);
  const int xc = x < 0 ? 0 : (x > g.W - 1 ? g.W - 1 : x);
  const size_t base = (size_t)b * g.DHW;

  auto clampk = [&](int k) { return k < 0 ? 0 : (k > g.D - 1 ? g.D - 1 : k); };
  // Rows / planes / columns outside the grid are clamped onto the border, whose mask byte is 0 (border cells are
  // never 'cont'), so no validity selects are needed: a clamped cell relaxes to 0 like the border cell it aliases.
  unsigned rowb[R0];                                     // wave-uniform cell offset of row slot rr (j = j0-2+rr) in a plane
#pragma unroll
  for (int rr = 0; rr < R0; ++rr) {
    const int j = j0 - 2 + rr;
    rowb[rr] = (unsigned)((j < 0 ? 0 : (j > g.H - 1 ? g.H - 1 : j)) * g.W);
  }
  // buffer offsets are 32-bit: they are taken relative to the first plane this segment touches, so only the segment
  // (<= a few dozen planes), not the whole field, has to stay below 4 GB
  const int k0 = clampk(k_lo - 2);
  auto planeoff = [&](int k) { return (unsigned)((clampk(k) - k0) * g.HW); };
  // buffer addressing: per-lane voffset (the column) + wave-uniform soffset (plane/row), no 64-bit VALU address math
  const unsigned xoff = (unsigned)xc * 4u;
  const size_t seg0 = base + (size_t)k0 * g.HW;
  const size_t left = (size_t)(g.D - k0) * g.HW;                       // cells from plane k0 to the end of the sample
  const unsigned ncell = left > 0x3fffffffu ? 0x3fffffffu : (unsigned)left;
  const BufRsrc r_p = make_rsrc(p_in + seg0, ncell * 4u), r_d = make_rsrc(div + seg0, ncell * 4u);
  const BufRsrc r_o = make_rsrc(p_out + seg0, ncell * 4u);
  // MIR: the mirror of this segment's plane range (byte offsets relative to its first mirrored plane)
  const int mri = (MIR && kbase != kb) ? 1 : 0;
  const int mq = (MIR && mir.sel[mri]) ? (int)((__hip_atomic_load(mir.sel[mri], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u) & 1u) : 0;
  const BufRsrc r_x = make_rsrc(MIR ? mir.out[mri][mq] + (size_t)b * mir.bstride : p_out, MIR ? (unsigned)((size_t)mir.n * g.HW * 4u) : 0u);
  const int mk0 = MIR ? mir.k[mri] : 0;
  // Mask bytes in row groups of four, maskq[b][k][j/4][i] = the bytes of rows 4(j/4) .. +3 of column i (jacobi3d_maskq_kernel):
  // the tile's own four rows are ONE dword load, the halo rows j0-1 / j0+4 byte 3 / byte 0 of the groups below / above (taking
  // them from the byte mask's 64-B rows instead was measured: slower) -- 3 vector-memory instructions and registers per plane
  // instead of 6, the same bytes of footprint as the byte mask
  static_assert(Z2R == 4 && Z2NW == 1, "a mask word holds the rows of a 4-row tile");
  const int HqM = (g.H + 3) >> 2, HqW = HqM * g.W;                      // row groups / words per plane
  const size_t leftq = (size_t)(g.D - k0) * HqW;
  const BufRsrc r_m = make_rsrc(maskq + ((size_t)b * g.D + k0) * HqW, (leftq > 0x3fffffffu ? 0x3fffffffu : (unsigned)leftq) * 4u);
  const unsigned mq_c = (unsigned)(by * g.W) * 4u, mq_m = (unsigned)((by > 0 ? by - 1 : 0) * g.W) * 4u,
                 mq_p = (unsigned)((by + 1 < HqM ? by + 1 : HqM - 1) * g.W) * 4u;
  const bool has_m = by > 0, has_p = by + 1 < HqM;                       // a group outside the grid: mask 0
  struct Mask3 { unsigned own, lo, hi; };
  auto ldm = [&](int k) {
    const unsigned po = (unsigned)((clampk(k) - k0) * HqW) * 4u;
    Mask3 m;
    m.own = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r_m, xoff, po + mq_c, 0);
    m.lo = has_m ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r_m, xoff + 3u, po + mq_m, 0) : 0u;
    m.hi = has_p ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r_m, xoff, po + mq_p, 0) : 0u;
    return m;
  };
  auto ldf = [&](const BufRsrc& r, unsigned cell) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, xoff, cell * 4u, 0));
  };
  auto ldp = [&](unsigned cell) { return ZERO ? 0.f : ldf(r_p, cell); };
  // quad layout: 16 B per column; the row groups below / of / above the tile (clamped into the grid: a clamped group stands
  // for rows outside the grid, whose mask bytes -- read through rowb[] -- are those of a border row, i.e. 0)
  static_assert(LAY == 0 || (Z2R == 4 && Z2NW == 1), "the quad layout holds the four rows of a tile");
  const int Hq = g.H >> 2;
  const unsigned xoff4 = (unsigned)xc * 16u;
  const unsigned gq_c = (unsigned)(by * 4 * g.W), gq_m = (unsigned)((by > 0 ? by - 1 : 0) * 4 * g.W),
                 gq_p = (unsigned)((by + 1 < Hq ? by + 1 : Hq - 1) * 4 * g.W);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // the 8 rows j0-2 .. j0+5 of plane offset `po` into dst[0..8)
  auto load_p0 = [&](float* dst, unsigned po) __attribute__((always_inline)) {
    if (ZERO) {
#pragma unroll
      for (int rr = 0; rr < R0; ++rr) dst[rr] = 0.f;
    } else if (LAY & 1) {
      const f32x2v lo = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r_p, xoff4 + 8u, (po + gq_m) * 4u, 0));
      const f32x4 mid = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_p, xoff4, (po + gq_c) * 4u, 0));
      const f32x2v hi = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r_p, xoff4, (po + gq_p) * 4u, 0));
      dst[0] = lo.x; dst[1] = lo.y; dst[2] = mid.x; dst[3] = mid.y; dst[4] = mid.z; dst[5] = mid.w; dst[6] = hi.x; dst[7] = hi.y;
    } else {
#pragma unroll
      for (int rr = 0; rr < R0; ++rr) dst[rr] = ldp(po + rowb[rr]);
    }
  };

  float P0[4][R0];                                       // p^0 plane ring: slot (t+d)&3 for planes t-1..t+2
  float P1[4][R1];                                       // p^1 plane ring (3 live)
  float AD[4][R1]; Mask3 AM[4];                          // div / mask ring, row slot rr <-> j = j0-1+rr (mask: lo, bytes 0..3 of own, hi)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) { P1[q][rr] = 0.f; AD[q][rr] = 0.f; }
    AM[q] = Mask3{0u, 0u, 0u};
#pragma unroll
    for (int rr = 0; rr < R0; ++rr) P0[q][rr] = 0.f;
  }
  int t = k_lo - 1;
  // prologue: planes t-1, t, t+1 of p^0 (slots 3, 0, 1) and the aux data of plane t (slot 0), all rows from memory
  {
    const unsigned pm = planeoff(t - 1), pc = planeoff(t), pp = planeoff(t + 1);
    load_p0(P0[3], pm); load_p0(P0[0], pc); load_p0(P0[1], pp);
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) AD[0][rr] = ldf(r_d, pc + rowb[rr + 1]);
    AM[0] = ldm(t);
  }
  const bool lane_out = (lane >= 2) & (lane <= 61) & xin;
  int prev_sel = 2;

  // one sweep over N rows: centre rows C[0..N), y-neighbours from the same plane, z-neighbours B / F
  auto sweep = [&](auto nn, auto off, int sel, const Mask3 M, const float* Cm1, const float* B, const float* F,
                   const float* DV, float* out) __attribute__((always_inline)) {
    constexpr int N = decltype(nn)::value, OFF = decltype(off)::value;      // row r is row slot OFF + r: slot 0 = lo, 1..4 = own bytes, 5 = hi
    auto mword = [&](int r) { return (OFF + r) == 0 ? M.lo : ((OFF + r) == 5 ? M.hi : M.own); };
    auto mshift = [&](int r) { return ((OFF + r) == 0 || (OFF + r) == 5) ? 0 : 8 * (OFF + r - 1); };
    float xs[N];
    bool bad = false;
    if (sel == 0) {
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float c = Cm1[r + 1];
        out[r] = relax3<0>(mword(r), mshift(r), c, dpp_from_left(c), dpp_from_right(c), Cm1[r], Cm1[r + 2], B[r], F[r], DV[r], xs[r]);
        bad |= __builtin_amdgcn_classf(out[r], 0x90);
      }
    } else if (sel == 1) {
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float c = Cm1[r + 1];
        out[r] = relax3<1>(mword(r), mshift(r), c, dpp_from_left(c), dpp_from_right(c), Cm1[r], Cm1[r + 2], B[r], F[r], DV[r], xs[r]);
        bad |= __builtin_amdgcn_classf(out[r], 0x90);
      }
    } else {
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float c = Cm1[r + 1];
        out[r] = relax3<2>(mword(r), mshift(r), c, dpp_from_left(c), dpp_from_right(c), Cm1[r], Cm1[r + 2], B[r], F[r], DV[r], xs[r]);
        bad |= __builtin_amdgcn_classf(out[r], 0x90);
      }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {     // a denormal quotient somewhere: scaled division
#pragma unroll
      for (int r = 0; r < N; ++r)
        if (__builtin_amdgcn_classf(out[r], 0x90)) out[r] = div6_tiny(xs[r]);
    }
#pragma unroll
    for (int r = 0; r < N; ++r)                            // cont ? v : 0
      out[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, out[r]) & __builtin_amdgcn_sbfe((int)mword(r), mshift(r), 1));
  };

  auto step = [&](auto ph, int t) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph)::value;
    constexpr int SM = (PH + 3) & 3, SC = PH, SP = (PH + 1) & 3, SN = (PH + 2) & 3, BUF = PH & 1;
    // ---- issue the loads of p^0 plane t+2 and of the aux data of plane t+1 (used one step later)
    const unsigned p2 = planeoff(t + 2), p1 = planeoff(t + 1);
    load_p0(P0[SN], p2);
#pragma unroll
    for (int rr = 0; rr < R1; ++rr) AD[SP][rr] = ldf(r_d, p1 + rowb[rr + 1]);
    AM[SP] = ldm(t + 1);
    // ---- sweep 1 on plane t, rows j0-1 .. j0+4
    const unsigned ob = AM[SC].own | ((AM[SC].lo | AM[SC].hi) & 0xffu);
    // no cell of these rows has an obstacle neighbour (0) / only x neighbours (1) / any (2)
    const int sel1 = __builtin_amdgcn_ballot_w64((ob & 0x78787878u) != 0) != 0 ? 2 : (__builtin_amdgcn_ballot_w64((ob & 0x06060606u) != 0) != 0 ? 1 : 0);
    sweep(IC<R1>{}, IC<0>{}, sel1, AM[SC], P0[SC], &P0[SM][1], &P0[SP][1], AD[SC], P1[SC]);
    // ---- sweep 2 on plane t-1, rows j0 .. j0+3 (p^1 of planes t-2, t-1, t = slots SN, SM, SC)
    if (t - 1 >= k_lo) {
      float v[Z2R];
      sweep(IC<Z2R>{}, IC<1>{}, prev_sel, AM[SM], P1[SM], &P1[SN][1], &P1[SC][1], &AD[SM][1], v);
      const unsigned ok = (unsigned)((t - 1 - k0) * g.HW + j0 * g.W) * 4u;
      // (MIR) the mirrored planes are [mk0, mk0 + mir.n): a wave-uniform test, not the buffer range check -- the scalar offset of a
      // buffer instruction is outside that check, and a plane below mk0 would wrap it around 2^32.  Inside the window the plane offset
      // travels in the bounds-checked vector offset.
      const bool mir_plane = MIR && (unsigned)(t - 1 - mk0) < (unsigned)mir.n;
      const unsigned okx = mir_plane ? (unsigned)((t - 1 - mk0) * g.HW + j0 * g.W) * 4u : 0u;
      if (LAY & 2) {                                       // j0 * W floats into the plane is the tile's row group in both layouts
        if (lane_out) {
          const f32x4 o = {v[0], v[1], v[2], v[3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_o, xoff4, ok, 0);
          if (mir_plane) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_x, xoff4 + okx, 0, 0);
        }
      } else
#pragma unroll
      for (int r = 0; r < Z2R; ++r) {
        if (lane_out && j0 + r < g.H) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), r_o, xoff, ok + (unsigned)(r * g.W) * 4u, 0);
          if (mir_plane) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), r_x, xoff + okx + (unsigned)(r * g.W) * 4u, 0, 0);
        }
      }
    }
    prev_sel = sel1;
  };

  while (true) {
    step(IC<0>{}, t); if (++t > k_hi) break;
    step(IC<1>{}, t); if (++t > k_hi) break;
    step(IC<2>{}, t); if (++t > k_hi) break;
    step(IC<3>{}, t); if (++t > k_hi) break;
  }
  }  // segments
}

constexpr int BX = 64, BY = 4;

// 2D, one sweep per launch, one thread per cell: the pTol > 0 solve (the reference's per-sweep convergence test needs every
// iterate in memory) and the last sweep of a solve whose caller wants the residual
__global__ __launch_bounds__(BX* BY) void jacobi_sweep_kernel(GridDims g, const float* __restrict__ flags,
                                                              const float* __restrict__ div,
                                                              const float* __restrict__ p_in,
                                                              float* __restrict__ p_out, bool from_zero) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int b = blockIdx.z;
  if (i < g.W && j < g.H) {
    const size_t o = (size_t)b * g.DHW + j * g.W + i;
    float v = 0.f;
    const float pc = from_zero ? 0.f : p_in[o];
    if (!is_border<false>(g, i, j, 0) && flags[o] != FNX_OBST) {
      if (from_zero) {
        float sum = 0.f + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f;
        v = (sum + div[o]) / 4.f;
      } else {
        const float n1 = flags[o - 1] == FNX_OBST ? pc : p_in[o - 1];
        const float n2 = flags[o + 1] == FNX_OBST ? pc : p_in[o + 1];
        const float n3 = flags[o - g.W] == FNX_OBST ? pc : p_in[o - g.W];
        const float n4 = flags[o + g.W] == FNX_OBST ? pc : p_in[o + g.W];
        float sum = n1 + n2;
        sum = sum + n3;
        sum = sum + n4;
        sum = sum + 0.f;
        sum = sum + 0.f;
        v = (sum + div[o]) / 4.f;
      }
    }
    p_out[o] = v;
  }
}

// ---- residual ||a - b||_2 per sample (cpp/fluids_init.cpp:961-971), reproducible: a FIXED grid of RES_BLOCKS blocks per
// sample, every block sums the squared differences of its grid-stride cells in a fixed order (fp64), the finishing kernel adds
// the RES_BLOCKS partials in index order.  No atomics: the same bits run to run, so a pTol exit does not depend on the
// order the waves retire in.  b == nullptr: b is all zeros (the first sweep of a solve).
constexpr int RES_BLOCKS = 512;
__global__ __launch_bounds__(256) void residual_partial_kernel(size_t per_sample, size_t count, size_t a_first, const float* __restrict__ a,
                                                               const float* __restrict__ bq, double* __restrict__ partials) {
  const int b = blockIdx.y;
  const float* pa = a + (size_t)b * per_sample + a_first;
  const float* pb = bq ? bq + (size_t)b * per_sample + a_first : nullptr;
  double acc = 0.0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < count; q += (size_t)RES_BLOCKS * 256) {
    const float d = pa[q] - (pb ? pb[q] : 0.f);            // the reference's p - p_prev, in fp32
    acc += (double)d * (double)d;
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {               // fixed tree: thread t adds thread t + off
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(size_t)b * RES_BLOCKS + blockIdx.x] = sh[0];
}

// sumsq[b] = sum of the sample's partials in index order (fp64 -> fp32); res (may be null) = max_b sqrt(sumsq[b])
__global__ void residual_finish_kernel(int B, const double* __restrict__ partials, float* __restrict__ sumsq, float* __restrict__ res) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) {
      double t = 0.0;
      for (int q = 0; q < RES_BLOCKS; ++q) t += partials[(size_t)b * RES_BLOCKS + q];
      if (sumsq) sumsq[b] = (float)t;
      m = fmaxf(m, (float)sqrt(t));
    }
    if (res) *res = m;
  }
}

__global__ void residual_root_kernel(int B, const float* __restrict__ sumsq, float* __restrict__ res) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) m = fmaxf(m, sqrtf(sumsq[b]));
    *res = m;
  }
}

template <int RW, int NW>
void launch_wg(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out, bool from_zero,
               int K, hipStream_t s) {
  const int OX = 64 - 2 * K, OYW = NW * RW - 2 * K;
  const int tiles_x = (g.W + OX - 1) / OX, tiles_y = (g.H + OYW - 1) / OYW;
  jacobi2d_wg_kernel<RW, NW><<<dim3(tiles_x * tiles_y, g.B), 64 * NW, 0, s>>>(g, flags, div, p_in, p_out, from_zero ? 1 : 0, tiles_x, K);
}

// Tile shape.  Measured on MI355X (bench.py, Jacobi ms per step; rows per wave x waves): 2048^2 x 100 sweeps K = 8: 8 x 8 0.450,
// 16 x 4 0.462, 8 x 4 0.533, 16 x 8 0.539, 8 x 16 0.597; 1024^2 x 28 K = 7: 8 x 8 0.0615, 8 x 4 0.0667; 128^2 x 28 K = 8:
// 8 x 4 0.036 (one 64 x 64 tile per CU leaves most of a small grid's CUs idle) -> 8 rows x 4 waves up to 160 Kcells, else
// 8 rows x 8 waves.
void launch_tiles(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out, bool from_zero,
                  int K, hipStream_t s) {
  const long cells = (long)g.W * g.H * g.B;
  if (K <= 8 && cells <= (160l << 10)) { launch_wg<8, 4>(g, flags, div, p_in, p_out, from_zero, K, s); return; }
  // deep launches (small grids, one tile per CU): a sweep costs a wave the chain over its rows, so 4 rows x 16 waves per tile
  // (128^2 x 28 in one launch: 24.0 -> 22.6 us, step 42 -> 40 us; with the lane masks of jacobi_rows_lm 17.6 us, step 35.6 us; same bits)
  if (K > 10) { launch_wg<4, 16>(g, flags, div, p_in, p_out, from_zero, K, s); return; }
  launch_wg<8, 8>(g, flags, div, p_in, p_out, from_zero, K, s);
}

}  // namespace

namespace fnx {

constexpr int KMAX_2D = 8, KLARGE_2D = 10, KDEEP_2D = 28;

// 64 x 64 tiles with an output window of (64 - 2K)^2: how many a launch of K sweeps needs
static long tiles_2d(const GridDims& g, int K) {
  const int o = 64 - 2 * K;
  return (long)((g.W + o - 1) / o) * ((g.H + o - 1) / o) * g.B;
}

int jacobi_max_sweeps_per_launch(const GridDims& g, bool is3d, int total) {
  if (is3d || g.D != 1) return 1;

  // Small grids are launch-latency bound (a launch costs ~5 us + ~0.35 us per sweep whatever the halo does to the work, as
  // long as every tile has a CU to itself): the fewest launches whose tiles all run at once, the sweeps dealt evenly.
  if (total > KMAX_2D) {
    static const long cus = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return (long)n;
    }();
    int kcap = 0;
    for (int K = KDEEP_2D; K > KMAX_2D; --K) if (tiles_2d(g, K) <= cus) { kcap = K; break; }
    if (kcap) {
      const int nl = (total + kcap - 1) / kcap;
      return (total + nl - 1) / nl;
    }
  }
  // a wave's chain per sweep is its 8 rows whatever K, so the halo (2K of the 64 columns and rows of a tile) is what limits K:
  // 7 where launches are still short (28 sweeps = 4 launches); on large grids at most 10, the sweeps dealt evenly over the
  // launches (measured at 2048^2 x 100 sweeps, ms per step: 13 launches of <= 8 0.584, 12 of <= 9 0.607, 10 of 10 0.574, 9 of <= 12
  // 0.615, 8 of <= 14 0.630)
  if ((long)g.W * g.H * g.B <= (2l << 20)) return 7;
  const int nl = (total + KLARGE_2D - 1) / KLARGE_2D;
  const int k = (total + nl - 1) / nl;
  return k < 1 ? 1 : k;
}

// 2D: nsweeps in [1, jacobi_max_sweeps_per_launch] sweeps from p_in into p_out
void launch_jacobi(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out, int nsweeps,
                   bool from_zero, hipStream_t s) {
  if (nsweeps >= 2 && nsweeps <= KDEEP_2D) { launch_tiles(g, flags, div, p_in, p_out, from_zero, nsweeps, s); return; }
  const dim3 grid((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B), block(BX, BY);           // exactly one sweep
  jacobi_sweep_kernel<<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero);
}

// 3D fast path (mask precomputed by launch_jacobi3d_mask)
// the mask allocation: B*D*H*W neighbour-mask bytes, then (256-B aligned) the same bytes in row groups of four
static size_t maskq_offset(const GridDims& g) { return (((size_t)g.B * g.DHW) + 255) & ~(size_t)255; }
size_t jacobi3d_mask_bytes(const GridDims& g) { return maskq_offset(g) + (size_t)g.B * g.D * ((g.H + 3) / 4) * g.W * 4; }

void launch_jacobi3d_mask(const GridDims& g, bool quirks, const float* flags, unsigned char* mask, hipStream_t s) {
  const dim3 grid((g.W + 63) / 64, (g.H + 3) / 4, g.B * g.D), block(64, 4);
  if (quirks) jacobi3d_mask_kernel<true><<<grid, block, 0, s>>>(g, flags, mask);
  else jacobi3d_mask_kernel<false><<<grid, block, 0, s>>>(g, flags, mask);
  const dim3 gridq((g.W + 63) / 64, ((g.H + 3) / 4 + 3) / 4, g.B * g.D);
  jacobi3d_maskq_kernel<<<gridq, block, 0, s>>>(g, mask, (unsigned*)(mask + maskq_offset(g)));
}

// can the two-sweep passes of this grid hand each other p in the row-quad layout (`lay` of launch_jacobi3d_x2)?
bool jacobi3d_quad_ok(const GridDims& g) { return g.H % 4 == 0 && Z2R == 4 && Z2NW == 1; }

// may a two-sweep launch of these plane ranges mirror its output (launch_jacobi3d_x2's `mirror`)?  One resident set of waves, both
// arrays in the same layout, not the from-zero pass
bool jacobi3d_mirror_ok(const GridDims& g, int np, bool two_ranges, bool from_zero, int lay) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long slots = (long)(Z2WPS * 4 / Z2NW) * cus;
  const long ntiles = (long)((g.W + 59) / 60) * ((g.H + Z2NW * Z2R - 1) / (Z2NW * Z2R)) * g.B;
  return !from_zero && (lay == 0 || lay == 3) && np >= 1 && ntiles * (two_ranges ? 2 : 1) <= slots && (size_t)(np + 4) * g.HW < 0x3fffffffu;
}

// two sweeps in one pass: p_in = p^n, p_out = p^{n+2}.  lay: bit 0 = p_in, bit 1 = p_out in the row-quad layout
void launch_jacobi3d_x2(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                        hipStream_t s, int kb, int ke, bool from_zero, int kb2, int lay, const JacobiMirror* mirror) {
  if (ke <= kb) { kb = 0; ke = g.D; kb2 = -1; }
  static const int slots = [] {                          // resident waves: Z2WPS per SIMD (<= 128 VGPRs each)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (Z2WPS * 4 / Z2NW) * cus;
  }();
  // Smallest plane chunk.  Every (tile, chunk) wave is resident at once, so a launch lasts (chunk + 2 lead-in steps) x
  // the per-step time of one wave, whatever the occupancy: small plane ranges (the slab driver's edge parts, small
  // grids) are cut as finely as the wave slots allow (measured 20 -> 14 us for 14 planes of 512^2).
  constexpr int zmin = 2;
  const int nxt = (g.W + 59) / 60, nyt = (g.H + Z2NW * Z2R - 1) / (Z2NW * Z2R);
  const int np = ke - kb, ntiles = nxt * nyt * g.B;
  // as many equal plane chunks per tile as fit one resident set
  int nzc = slots / ntiles;
  if (kb2 >= 0) nzc /= 2;                                // two plane ranges share the resident set
  if (nzc < 1) nzc = 1;
  int zchunk = (np + nzc - 1) / nzc;
  if (zchunk < zmin) zchunk = zmin;
  if (ntiles > slots) zchunk = 0;                        // more tiles than slots: even split of the (tile, plane) space
  if (zchunk == 0 && (size_t)(np + 4) * g.HW >= 0x3fffffffu) zchunk = 64;   // keep a segment's 32-bit offsets below 4 GB
  if (kb2 >= 0 && (zchunk <= 0 || 2 * ntiles > slots)) {   // no room for both ranges at once: one after the other
    launch_jacobi3d_x2(g, mask, div, p_in, p_out, s, kb, ke, from_zero, -1, lay, nullptr);
    launch_jacobi3d_x2(g, mask, div, p_in, p_out, s, kb2, kb2 + np, from_zero, -1, lay, nullptr);
    return;
  }
  long long G;
  if (zchunk > 0) {
    G = (long long)ntiles * ((np + zchunk - 1) / zchunk) * (kb2 >= 0 ? 2 : 1);
    G = ((G + 7) / 8) * 8;
  } else {
    G = (long long)ntiles * np / 8;
    if (G > slots) G = slots;
    G = (G / 8) * 8;
    if (G < 8) G = 8;
  }
  const dim3 grid((unsigned)G), block(64, Z2NW);
  const unsigned* maskq = (const unsigned*)(mask + maskq_offset(g));
  if (mirror) {                                            // (jacobi3d_mirror_ok has been checked: resident set, not from zero, lay 0 or 3)
    MirrorArgs m{};
    for (int r = 0; r < 2; ++r) { m.out[r][0] = mirror->out[r][0]; m.out[r][1] = mirror->out[r][1]; m.sel[r] = mirror->sel[r]; }
    m.k[0] = mirror->k[0]; m.k[1] = mirror->k[1]; m.n = mirror->n;
    m.bstride = mirror->bstride; m.clock = mirror->clock;
    if (lay == 3) jacobi3d_march2_kernel<false, false, 3, true><<<grid, block, 0, s>>>(g, maskq, div, p_in, p_out, nxt, nyt, zchunk, kb, ke, kb2, m);
    else jacobi3d_march2_kernel<false, false, 0, true><<<grid, block, 0, s>>>(g, maskq, div, p_in, p_out, nxt, nyt, zchunk, kb, ke, kb2, m);
    return;
  }
#define J3Q(Z, S, L) jacobi3d_march2_kernel<Z, S, L><<<grid, block, 0, s>>>(g, maskq, div, p_in, p_out, nxt, nyt, zchunk, kb, ke, kb2)
#define J3Q_S(Z, L) do { if (zchunk > 0) J3Q(Z, false, L); else J3Q(Z, true, L); } while (0)
  if (from_zero) lay &= 2;                                 // no input: its layout does not matter
  if (lay == 0) { if (from_zero) J3Q_S(true, 0); else J3Q_S(false, 0); }
  else if (lay == 2) { if (from_zero) J3Q_S(true, 2); else J3Q_S(false, 2); }
  else if (lay == 3) J3Q_S(false, 3);
  else J3Q_S(false, 1);
#undef J3Q_S
#undef J3Q
}

void launch_jacobi3d(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                     bool from_zero, hipStream_t s, int kb, int ke) {
  if (ke <= kb) { kb = 0; ke = g.D; }
  if (from_zero) {
    const size_t count = (size_t)(ke - kb) * g.HW;
    size_t nb = (count + 256 * 4 - 1) / (256 * 4);
    if (nb > 4096) nb = 4096;
    jacobi3d_first_kernel<<<(unsigned)nb, 256, 0, s>>>(g.B, (size_t)g.DHW, (size_t)kb * g.HW, count, div, mask, p_out);
    return;
  }
  const int nzc = (ke - kb + ZCHUNK - 1) / ZCHUNK;
  const dim3 grid((g.W + 63) / 64, (g.H + 4 * ZR - 1) / (4 * ZR), g.B * nzc), block(64, 4);
  jacobi3d_march_kernel<<<grid, block, 0, s>>>(g, mask, div, p_in, p_out, nzc, kb, ke);
}

size_t residual_scratch_bytes(int B) { return (size_t)B * RES_BLOCKS * sizeof(double); }

void launch_residual(int B, size_t per_sample, size_t first, size_t count, const float* a, const float* b, double* partials,
                     float* sumsq, float* res, hipStream_t s) {
  residual_partial_kernel<<<dim3(RES_BLOCKS, B), 256, 0, s>>>(per_sample, count, first, a, b, partials);
  residual_finish_kernel<<<1, 64, 0, s>>>(B, partials, sumsq, res);
}

void launch_residual_root(int B, const float* sumsq, float* res, hipStream_t s) {
  residual_root_kernel<<<1, 64, 0, s>>>(B, sumsq, res);
}

}  // namespace fnx
