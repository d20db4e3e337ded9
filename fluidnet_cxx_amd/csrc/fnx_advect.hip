// MacCormack / semi-Lagrangian advection kernels for gfx950.
//
// Replaces advectScalar (cpp/fluids_init.cpp:265-382) and advectVel (:656-807): the reference issues
// ~3300 / ~1400 ATen ops per call; here each advection is two launches (forward pass; backward pass fused
// with the MacCormack correction and clamp), one thread per cell, x fastest so every field read that is
// not a gather is a coalesced 256-B wave access.  Gathers (trace end points, clamp neighbourhoods) stay
// within a few cells of the thread's own cell and are served by L1/L2.
//
// Bit-parity: see fnx_device.h.  Compiled with -ffp-contract=off.
#include "fnx_device.h"
#include "fnx_kernels.h"

namespace {

constexpr int BX = 64, BY = 4;

#include "fnx_advect_cells.h"      // CellId and the per-cell functions the kernels below run (shared with fnx_small.hip)


template <bool IS3D>
__device__ __forceinline__ CellId cell_id(const GridDims& g) {
  CellId c;
  c.i = blockIdx.x * BX + threadIdx.x;
  c.j = blockIdx.y * BY + threadIdx.y;
  const int bk = blockIdx.z;
  c.b = IS3D ? bk / g.KN : bk;
  c.k = IS3D ? g.K0 + (bk - c.b * g.KN) : 0;
  c.valid = (c.i < g.W) & (c.j < g.H);
  return c;
}

template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(BX* BY) void sl_scalar_kernel(GridDims g, float dt, const float* __restrict__ src,
                                                           const float* __restrict__ U,
                                                           const float* __restrict__ flags, float* __restrict__ dst,
                                                           int* __restrict__ cell_out) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_scalar_cell<IS3D, QUIRKS, SAMPLE_OUTSIDE>(g, c, dt, src, U, flags, dst, cell_out);
}

// ---------------------------------------------------------------------------------------------------
// Clamp bounds of MacCormackClampFluidNet (:154-263) as a field: for every cell c the min / max of src over the
// 3x3(x3) box around c, taken over the cells that are inside the (local) domain and -- unless sample_outside_fluid --
// fluid.  The backward kernel then fetches ONE 8-byte pair at the traced cell instead of walking 27 cells x 2 fields
// per thread.  min/max are exact, commutative and NaN-ignoring (v_min_f32 / v_max_f32 order -0 < +0), so the
// separable evaluation gives the same bits as the reference's 27-step fold; "no cell qualified" (the reference then
// keeps the forward value) is encoded as mn = NaN.
// A wave covers 62 columns (lanes 0 / 63 are halo columns) x BOX_R rows and marches along z: per plane it reduces
// rows j0-1..j0+BOX_R in x (DPP) and y, and keeps the reduced planes k-1, k, k+1 in registers.
// ---------------------------------------------------------------------------------------------------
constexpr int BOX_R = 8, BOX_ZC = 16;

__device__ __forceinline__ float dppf_left(float v) {     // lane-1 (0.0 into lane 0: a halo lane, never stored)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dppf_right(float v) {    // lane+1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned dppu_left(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned dppu_right(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

template <bool IS3D, bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256) void box_minmax_kernel(GridDims g, const float* __restrict__ src,
                                                         const float* __restrict__ flags, float2* __restrict__ box,
                                                         int nzc) {
  const int lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int x = blockIdx.x * 62 - 1 + lane;
  const int j0 = (blockIdx.y * 4 + w) * BOX_R;
  const int bz = blockIdx.z;
  const int zc = bz % nzc, b = bz / nzc;
  const int k_lo = g.K0 + zc * BOX_ZC, k_hi = min(k_lo + BOX_ZC, g.K0 + g.KN);      // planes of the compute window
  if (j0 >= g.H) return;
  const bool xin = (x >= 0) & (x < g.W);
  const int xc = clampi(x, 0, g.W - 1);
  const size_t base = (size_t)b * g.DHW;
  // all loads of a plane are unconditional (clamped rows / planes) so they issue back to back; validity only gates `ok`
  auto plane = [&](int k, float* omn, float* omx, unsigned* oan) {
    float sv[BOX_R + 2], fv[BOX_R + 2];
    const bool kin = (k >= 0) & (k < g.D);                           // the reference's box is clipped to the grid
    const size_t ok_ = base + (size_t)clampi(k, 0, g.D - 1) * g.HW + xc;
#pragma unroll
    for (int rr = 0; rr < BOX_R + 2; ++rr) {
      const size_t o = ok_ + (size_t)clampi(j0 - 1 + rr, 0, g.H - 1) * g.W;
      sv[rr] = src[o];
      fv[rr] = SAMPLE_OUTSIDE ? FNX_FLUID : flags[o];
    }
    float rmn[BOX_R + 2], rmx[BOX_R + 2]; unsigned ran[BOX_R + 2];
#pragma unroll
    for (int rr = 0; rr < BOX_R + 2; ++rr) {
      const int j = j0 - 1 + rr;
      const bool ok = xin & kin & (j >= 0) & (j < g.H) & (fv[rr] == FNX_FLUID);
      const float vmn = ok ? sv[rr] : INFINITY, vmx = ok ? sv[rr] : -INFINITY;
      const unsigned va = ok ? 1u : 0u;
      rmn[rr] = fminf(fminf(dppf_left(vmn), vmn), dppf_right(vmn));
      rmx[rr] = fmaxf(fmaxf(dppf_left(vmx), vmx), dppf_right(vmx));
      ran[rr] = dppu_left(va) | va | dppu_right(va);
    }
#pragma unroll
    for (int r = 0; r < BOX_R; ++r) {
      omn[r] = fminf(fminf(rmn[r], rmn[r + 1]), rmn[r + 2]);
      omx[r] = fmaxf(fmaxf(rmx[r], rmx[r + 1]), rmx[r + 2]);
      oan[r] = ran[r] | ran[r + 1] | ran[r + 2];
    }
  };
  const bool lane_out = (lane >= 1) & (lane <= 62) & xin;
  auto store = [&](int k, int r, float lo, float hi, bool any) {
    if (lane_out && j0 + r < g.H)
      box[base + (size_t)k * g.HW + (size_t)(j0 + r) * g.W + x] = make_float2(any ? lo : __builtin_nanf(""), hi);
  };
  float mn[3][BOX_R], mx[3][BOX_R]; unsigned an[3][BOX_R];           // xy-reduced planes k-1, k, k+1
  if (!IS3D) {
    plane(0, mn[0], mx[0], an[0]);
#pragma unroll
    for (int r = 0; r < BOX_R; ++r) store(0, r, mn[0][r], mx[0][r], an[0][r] != 0);
    return;
  }
  plane(k_lo - 1, mn[0], mx[0], an[0]);
  plane(k_lo, mn[1], mx[1], an[1]);
  for (int k = k_lo; k < k_hi; ++k) {
    plane(k + 1, mn[2], mx[2], an[2]);
#pragma unroll
    for (int r = 0; r < BOX_R; ++r)
      store(k, r, fminf(fminf(mn[0][r], mn[1][r]), mn[2][r]), fmaxf(fmaxf(mx[0][r], mx[1][r]), mx[2][r]),
            (an[0][r] | an[1][r] | an[2][r]) != 0);
#pragma unroll
    for (int r = 0; r < BOX_R; ++r) {
      mn[0][r] = mn[1][r]; mx[0][r] = mx[1][r]; an[0][r] = an[1][r];
      mn[1][r] = mn[2][r]; mx[1][r] = mx[2][r]; an[1][r] = an[2][r];
    }
  }
}
template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(BX* BY) void sl_scalar_bwd_clamp_kernel(GridDims g, float dt, float half_s,
                                                                     const float* __restrict__ src,
                                                                     const float* __restrict__ fwd,
                                                                     const int* __restrict__ cell_in,
                                                                     const float* __restrict__ U,
                                                                     const float* __restrict__ flags,
                                                                     const float2* __restrict__ box,
                                                                     float* __restrict__ dst) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_scalar_bwd_clamp_cell<IS3D, QUIRKS, SAMPLE_OUTSIDE>(g, c, dt, half_s, src, fwd, cell_in, U, flags, box, dst);
}
template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void sl_mac_kernel(GridDims g, float dt, const float* __restrict__ src,
                                                        const float* __restrict__ U, const float* __restrict__ flags,
                                                        float* __restrict__ dst) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_mac_cell<IS3D, QUIRKS>(g, c, dt, src, U, flags, dst);
}
template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void sl_mac_bwd_clamp_kernel(GridDims g, float dt, float half_s,
                                                                  const float* __restrict__ orig,
                                                                  const float* __restrict__ fwd,
                                                                  const float* __restrict__ U,
                                                                  const float* __restrict__ flags,
                                                                  float* __restrict__ dst) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_mac_bwd_clamp_cell<IS3D, QUIRKS>(g, c, dt, half_s, orig, fwd, U, flags, dst);
}

// ---------------------------------------------------------------------------------------------------
// The density and the velocity MacCormack advections of one time step in two launches instead of four (forward passes
// together, backward/clamp passes together; 3D: the clamp-bounds pass in between).  In 2D at 128^2 .. 1024^2 each pass
// is a 7-14 us launch that is mostly ramp-up; in 3D the two chains share their velocity loads (1.13 -> 1.06 ms at
// 256^3).  The cell functions are the ones above, so the results are the same bits.
// ---------------------------------------------------------------------------------------------------
template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(BX* BY) void advect_fwd_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                              const float* __restrict__ U,
                                                              const float* __restrict__ flags,
                                                              float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                              float* __restrict__ U_fwd) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_scalar_cell<IS3D, QUIRKS, SAMPLE_OUTSIDE>(g, c, dt, rho, U, flags, rho_fwd, cell_out);
  if (QUIRKS) sl_mac_cell<IS3D, QUIRKS>(g, c, dt, U, U, flags, U_fwd);
  else sl_mac_cell_flat<IS3D>(g, c, dt, U, U, flags, U_fwd);
}

template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(BX* BY) void advect_bwd_kernel(GridDims g, float dt, float half_s,
                                                              const float* __restrict__ rho,
                                                              const float* __restrict__ rho_fwd,
                                                              const int* __restrict__ cell_in,
                                                              const float* __restrict__ U,
                                                              const float* __restrict__ U_fwd,
                                                              const float* __restrict__ flags,
                                                              const float2* __restrict__ box,
                                                              float* __restrict__ rho_dst, float* __restrict__ U_dst) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_scalar_bwd_clamp_cell<IS3D, QUIRKS, SAMPLE_OUTSIDE>(g, c, dt, half_s, rho, rho_fwd, cell_in, U, flags, box, rho_dst);
  if (QUIRKS) sl_mac_bwd_clamp_cell<IS3D, QUIRKS>(g, c, dt, half_s, U, U_fwd, U, flags, U_dst);
  else sl_mac_bwd_clamp_cell_flat<IS3D>(g, c, dt, half_s, U, U_fwd, U, flags, U_dst);
}

#include "fnx_advect_march.h"
#include "fnx_advect_tile2d.h"

inline dim3 cell_grid(const GridDims& g) { return dim3((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B * g.KN); }

}  // namespace

namespace fnx {

size_t advect_fix_words(const GridDims& g) { return (size_t)g.B * g.D * g.H * ((g.W + 63) / 64); }

#define DISPATCH3(IS3D, Q, SO, KERNEL, ...)                                                   \
  do {                                                                                        \
    if (IS3D) {                                                                               \
      if (Q) { if (SO) KERNEL<true, true, true> __VA_ARGS__; else KERNEL<true, true, false> __VA_ARGS__; } \
      else   { if (SO) KERNEL<true, false, true> __VA_ARGS__; else KERNEL<true, false, false> __VA_ARGS__; } \
    } else {                                                                                  \
      if (SO) KERNEL<false, false, true> __VA_ARGS__; else KERNEL<false, false, false> __VA_ARGS__; \
    }                                                                                         \
  } while (0)

#define DISPATCH2(IS3D, Q, KERNEL, ...)                                                       \
  do {                                                                                        \
    if (IS3D) { if (Q) KERNEL<true, true> __VA_ARGS__; else KERNEL<true, false> __VA_ARGS__; } \
    else KERNEL<false, false> __VA_ARGS__;                                                    \
  } while (0)

void launch_sl_scalar(const GridDims& g, bool is3d, bool quirks, bool sample_outside, float dt, const float* src,
                      const float* U, const float* flags, float* dst, int* cell_out, hipStream_t s) {
  const dim3 grid = cell_grid(g), block(BX, BY);
  DISPATCH3(is3d, quirks, sample_outside, sl_scalar_kernel, <<<grid, block, 0, s>>>(g, dt, src, U, flags, dst, cell_out));
}

void launch_sl_scalar_bwd_clamp(const GridDims& g, bool is3d, bool quirks, bool sample_outside, float dt, float half_s,
                                const float* src, const float* fwd, const int* cell_in, const float* U,
                                const float* flags, const float* box, float* dst, hipStream_t s) {
  const dim3 grid = cell_grid(g), block(BX, BY);
  DISPATCH3(is3d, quirks, sample_outside, sl_scalar_bwd_clamp_kernel,
            <<<grid, block, 0, s>>>(g, dt, half_s, src, fwd, cell_in, U, flags, (const float2*)box, dst));
}

void launch_box_minmax(const GridDims& g, bool sample_outside, const float* src, const float* flags, float* box,
                       hipStream_t s) {
  const int nzc = (g.KN + BOX_ZC - 1) / BOX_ZC;
  const dim3 grid((g.W + 61) / 62, (g.H + 4 * BOX_R - 1) / (4 * BOX_R), g.B * nzc), block(64, 4);
  if (g.D > 1) {
    if (sample_outside) box_minmax_kernel<true, true><<<grid, block, 0, s>>>(g, src, flags, (float2*)box, nzc);
    else box_minmax_kernel<true, false><<<grid, block, 0, s>>>(g, src, flags, (float2*)box, nzc);
  } else {
    if (sample_outside) box_minmax_kernel<false, true><<<grid, block, 0, s>>>(g, src, flags, (float2*)box, nzc);
    else box_minmax_kernel<false, false><<<grid, block, 0, s>>>(g, src, flags, (float2*)box, nzc);
  }
}

void launch_sl_mac(const GridDims& g, bool is3d, bool quirks, float dt, const float* src, const float* U,
                   const float* flags, float* dst, hipStream_t s) {
  const dim3 grid = cell_grid(g), block(BX, BY);
  DISPATCH2(is3d, quirks, sl_mac_kernel, <<<grid, block, 0, s>>>(g, dt, src, U, flags, dst));
}

// z-marching tile kernels (fnx_advect_march.h): the planes of the compute window are cut into chunks so that the
// launch is a whole number of rounds of resident workgroups (3 per CU); a chunk re-reads 2 lead-in planes.
static void tile_launch_geometry(const GridDims& g, int& ntx, int& nty, int& zchunk, unsigned& G) {
  static const int slots = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return 3 * cus;
  }();
  ntx = (g.W + 63) / 64; nty = (g.H + ATR - 1) / ATR;
  const long ntiles = (long)ntx * nty * g.B;
  long nzc = (2l * slots + ntiles - 1) / ntiles;          // two rounds of workgroups
  if (nzc < 1) nzc = 1;
  zchunk = (int)((g.KN + nzc - 1) / nzc);
  if (zchunk < 8) zchunk = g.KN < 8 ? g.KN : 8;
  nzc = (g.KN + zchunk - 1) / zchunk;
  long n = ntiles * nzc;
  G = (unsigned)(((n + 7) / 8) * 8);
}

static void launch_fwd_tile(const GridDims& g, bool sample_outside, float dt, const float* rho, const float* U,
                            const float* flags, float* rho_fwd, int* cell, float* U_fwd, float* box, unsigned long long* fix_s,
                            unsigned long long* fix_v, hipStream_t s) {
  int ntx, nty, zchunk; unsigned G;
  tile_launch_geometry(g, ntx, nty, zchunk, G);
  const unsigned nfix = (unsigned)(((size_t)g.B * g.KN * g.H * ntx + 255) / 256);
  if (sample_outside) {
    advect3d_fwd_tile_kernel<true><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, (float2*)box, fix_s, fix_v, ntx, nty, zchunk);
    advect3d_fwd_fix_kernel<true><<<dim3(nfix), 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, fix_s, fix_v, ntx);
  } else {
    advect3d_fwd_tile_kernel<false><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, (float2*)box, fix_s, fix_v, ntx, nty, zchunk);
    advect3d_fwd_fix_kernel<false><<<dim3(nfix), 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, fix_s, fix_v, ntx);
  }
}

// MacCormack self-advection of U plus advection of rho by U, both by the OLD U (simulate.py:75-93), in two launches
void launch_advect_fused(const GridDims& g, const GridDims& gfwd, bool is3d, bool quirks, bool sample_outside, float dt,
                         float half_s, const float* rho, const float* U, const float* flags, float* rho_fwd, int* cell,
                         float* U_fwd, float* box, float* rho_dst, float* U_dst, unsigned long long* fix, hipStream_t s, int plan) {
  // fix-up bitmaps of the tile kernels: 4 x (one 64-bit word per 64-cell row segment): fwd density, fwd velocity,
  // bwd density, bwd velocity
  const size_t nwords = advect_fix_words(g);
  const dim3 block(BX, BY);
  // forward passes and clamp bounds on `gfwd` (the compute window widened by what the backward pass reads)
  // 3D default semantics: the z-marching LDS tile kernels (fnx_advect_march.h); quirks mode and plane ranges beyond the
  // tile kernels' 32-bit offsets: one thread per cell
  // 2D: LDS tile kernels (fnx_advect_tile2d.h) on grids large enough to pay for the two fix-up launches (measured, advection per
  // step, tiles vs one thread per cell: 2048^2 105 vs 129 us, 1024^2 38.6 vs 37.2, 128^2 17.6 vs 12.6), wherever a row offset
  // fits the tiles' 32-bit buffer offsets
  const bool want_tiles = plan == 1 || (plan == 0 && (is3d || (size_t)g.HW * g.B >= ((size_t)3 << 19)));
  if (!is3d && want_tiles && (size_t)g.HW < 0x3fffffffu) {
    const int ntx = (g.W + 63) / 64, nty = (g.H + T2R - 1) / T2R;
    const dim3 grid((unsigned)(ntx * nty * g.B));
    const unsigned nfix = (unsigned)(((size_t)g.B * g.H * ntx + 255) / 256);
    unsigned long long *ff_s = fix, *ff_v = fix + nwords, *fb_s = fix + 2 * nwords, *fb_v = fix + 3 * nwords;
    if (sample_outside) {
      advect2d_fwd_tile_kernel<true><<<grid, 64 * T2NW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx, nty);
      advect2d_fwd_fix_kernel<true><<<nfix, 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx);
      advect2d_bwd_tile_kernel<true><<<grid, 64 * T2NW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx, nty);
      advect2d_bwd_fix_kernel<true><<<nfix, 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx);
    } else {
      advect2d_fwd_tile_kernel<false><<<grid, 64 * T2NW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx, nty);
      advect2d_fwd_fix_kernel<false><<<nfix, 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx);
      advect2d_bwd_tile_kernel<false><<<grid, 64 * T2NW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx, nty);
      advect2d_bwd_fix_kernel<false><<<nfix, 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx);
    }
    return;
  }
  // (the tile kernel also reduces the clamp bounds of the density step from the rho planes it streams)
  if (is3d && want_tiles && !quirks && (size_t)(gfwd.KN + 2) * gfwd.HW < 0x3fffffffu) {
    launch_fwd_tile(gfwd, sample_outside, dt, rho, U, flags, rho_fwd, cell, U_fwd, box, fix, fix + nwords, s);
  } else {
    DISPATCH3(is3d, quirks, sample_outside, advect_fwd_kernel, <<<cell_grid(gfwd), block, 0, s>>>(gfwd, dt, rho, U, flags, rho_fwd, cell, U_fwd));
    if (is3d) launch_box_minmax(gfwd, sample_outside, rho, flags, box, s);
  }
  if (is3d && want_tiles && !quirks && (size_t)(g.KN + 2) * g.HW < 0x3fffffffu) {
    int ntx, nty, zchunk; unsigned G;
    tile_launch_geometry(g, ntx, nty, zchunk, G);
    unsigned long long* fb_s = fix + 2 * nwords;
    unsigned long long* fb_v = fix + 3 * nwords;
    if (sample_outside) advect3d_bwd_scalar_tile_kernel<true><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, flags, (const float2*)box, rho_dst, fb_s, ntx, nty, zchunk);
    else advect3d_bwd_scalar_tile_kernel<false><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, flags, (const float2*)box, rho_dst, fb_s, ntx, nty, zchunk);
    advect3d_bwd_vel_tile_kernel<<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, U, U_fwd, flags, U_dst, fb_v, ntx, nty, zchunk);
    const unsigned nfix = (unsigned)(((size_t)g.B * g.KN * g.H * ntx + 255) / 256);
    if (sample_outside) advect3d_bwd_fix_kernel<true><<<dim3(nfix), 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, (const float2*)box, rho_dst, U_dst, fb_s, fb_v, ntx);
    else advect3d_bwd_fix_kernel<false><<<dim3(nfix), 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, (const float2*)box, rho_dst, U_dst, fb_s, fb_v, ntx);
    return;
  }
  DISPATCH3(is3d, quirks, sample_outside, advect_bwd_kernel,
            <<<cell_grid(g), block, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, (const float2*)box, rho_dst, U_dst));
}

void launch_sl_mac_bwd_clamp(const GridDims& g, bool is3d, bool quirks, float dt, float half_s, const float* orig,
                             const float* fwd, const float* U, const float* flags, float* dst, hipStream_t s) {
  const dim3 grid = cell_grid(g), block(BX, BY);
  DISPATCH2(is3d, quirks, sl_mac_bwd_clamp_kernel, <<<grid, block, 0, s>>>(g, dt, half_s, orig, fwd, U, flags, dst));
}

}  // namespace fnx
