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

struct CellId { int b, k, j, i; bool valid; };

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

// ---------------------------------------------------------------------------------------------------
// Scalar: one semi-Lagrangian pass (SemiLagrangeEulerFluidNet[SavePos], fluids_init.cpp:12-133).
// Writes dst (border -> 0) and, if cell_out != nullptr, the clamped cell of the traced position
// (what getClampBounds :175-178 derives from fwd_pos).
// ---------------------------------------------------------------------------------------------------
template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__device__ __forceinline__ void sl_scalar_cell(const GridDims& g, const CellId& c, float dt, const float* __restrict__ src,
                                                           const float* __restrict__ U,
                                                           const float* __restrict__ flags, float* __restrict__ dst,
                                                           int* __restrict__ cell_out) {
  constexpr int NC = IS3D ? 3 : 2;
  const Field fs{src + (size_t)c.b * g.DHW}, ff{flags + (size_t)c.b * g.DHW}, fu{U + (size_t)c.b * NC * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float ctr[3] = { (float)c.i + 0.5f, (float)c.j + 0.5f, (float)(c.k + g.zoff) + 0.5f };   // global z
  float val = 0.f;
  float p[3] = { ctr[0], ctr[1], ctr[2] };
  if (!is_border<IS3D>(g, c.i, c.j, c.k)) {
    if (ff.p[o] != FNX_FLUID) {
      val = fs.p[o];                                   // "don't advect solid geometry"
    } else {
      float cen[3], disp[3];
      get_centered<IS3D>(g, fu, c.i, c.j, c.k, cen);
#pragma unroll
      for (int a = 0; a < 3; ++a) disp[a] = (-dt) * cen[a];
      line_trace(g, ff, ctr, disp, p);
      val = SAMPLE_OUTSIDE ? interpol<IS3D>(g, fs, 0, p[0], p[1], p[2])
                           : interpol_with_fluid<IS3D, QUIRKS>(g, fs, ff, p[0], p[1], p[2]);
    }
  }
  dst[(size_t)c.b * g.DHW + o] = val;
  if (cell_out) {
    const int i0 = clampi((int)p[0], 0, g.W - 1), j0 = clampi((int)p[1], 0, g.H - 1);
    // Q10: k0 = 0 in the reference.  Stored with a +1 plane bias so that global plane -1.. maps to a valid int
    const int k0 = (IS3D && !QUIRKS) ? clampi((int)p[2], 0, g.Dglob - 1) - g.zoff : -g.zoff;
    cell_out[(size_t)c.b * g.DHW + o] = IS3D ? (k0 + 1) * g.HW + j0 * g.W + i0 : ((j0 << 16) | i0);   // 2D: W, H < 65536 (host check)
  }
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

// Backward pass on fwd + MacCormackCorrect (:135-148) + MacCormackClampFluidNet (:154-263)
template <bool IS3D, bool QUIRKS, bool SAMPLE_OUTSIDE>
__device__ __forceinline__ void sl_scalar_bwd_clamp_cell(const GridDims& g, const CellId& c, float dt, float half_s,
                                                                     const float* __restrict__ src,
                                                                     const float* __restrict__ fwd,
                                                                     const int* __restrict__ cell_in,
                                                                     const float* __restrict__ U,
                                                                     const float* __restrict__ flags,
                                                                     const float2* __restrict__ box,
                                                                     float* __restrict__ dst) {
  constexpr int NC = IS3D ? 3 : 2;
  const Field fs{src + (size_t)c.b * g.DHW}, fw{fwd + (size_t)c.b * g.DHW}, ff{flags + (size_t)c.b * g.DHW},
      fu{U + (size_t)c.b * NC * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const bool border = is_border<IS3D>(g, c.i, c.j, c.k);
  const bool fluid = ff.p[o] == FNX_FLUID;
  const float f = fw.p[o];
  float bwd = 0.f;
  if (!border) {
    if (!fluid) {
      bwd = f;
    } else {
      const float ctr[3] = { (float)c.i + 0.5f, (float)c.j + 0.5f, (float)(c.k + g.zoff) + 0.5f };
      float cen[3], disp[3], p[3];
      get_centered<IS3D>(g, fu, c.i, c.j, c.k, cen);
      const float ndt = -dt;
#pragma unroll
      for (int a = 0; a < 3; ++a) disp[a] = (-ndt) * cen[a];
      line_trace(g, ff, ctr, disp, p);
      bwd = SAMPLE_OUTSIDE ? interpol<IS3D>(g, fw, 0, p[0], p[1], p[2])
                           : interpol_with_fluid<IS3D, QUIRKS>(g, fw, ff, p[0], p[1], p[2]);
    }
  }
  float d = f;
  if (fluid) d = f + half_s * (fs.p[o] - bwd);          // applied on border cells too (reference :371)
  if (!border) {
    // traced cell: 3D (k0+1)*HW + j0*W + i0 (no integer division on the common path), 2D (j0 << 16) | i0
    const int cell = cell_in[(size_t)c.b * g.DHW + o];
    float mn, mx;
    bool any;
    int k0 = 0, j0 = 0, i0 = 0;
    if (IS3D && box != nullptr && cell >= g.HW && cell < g.HW + g.DHW) {
      // the traced cell lies in this slab: its clamp bounds were reduced once by box_minmax_kernel
      const float2 bb = box[(size_t)c.b * g.DHW + (size_t)(cell - g.HW)];
      mn = bb.x; mx = bb.y;
      any = !(mn != mn);
    } else {
      if (IS3D) {
        const int kb = cell / g.HW;                         // local plane + 1
        k0 = kb - 1;
        const int r = cell - kb * g.HW;
        j0 = r / g.W; i0 = r - j0 * g.W;
      } else {
        j0 = (int)((unsigned)cell >> 16); i0 = cell & 0xffff;   // (unsigned: rows >= 32768 set the sign bit)
      }
      // 2D (9 cells: cheaper than a separate pass), or traced into a plane this slab does not hold: walk the clipped
      // box directly, as the reference does
      mn = INFINITY; mx = -INFINITY; any = false;
      // 3x3(x3) neighbourhood of the traced cell: every load is unconditional (clamped address) and the
      // in-domain / is-fluid tests only gate the min/max, so the 18-54 loads are independent and issue back to back.
#pragma unroll
      for (int dk = (IS3D ? -1 : 0); dk <= (IS3D ? 1 : 0); ++dk) {
        const int kk = k0 + dk;
        const bool vk = (kk + g.zoff >= 0) & (kk + g.zoff < g.Dglob) & (kk >= 0) & (kk < g.D);   // in the domain and in this slab
        const int kc = clampi(kk, 0, g.D - 1);
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
          const int jj = j0 + dj;
          const bool vj = vk & (jj >= 0) & (jj < g.H);
          const int jc = clampi(jj, 0, g.H - 1);
#pragma unroll
          for (int di = -1; di <= 1; ++di) {
            const int ii = i0 + di;
            const bool vi = vj & (ii >= 0) & (ii < g.W);
            const size_t q = (size_t)kc * g.HW + jc * g.W + clampi(ii, 0, g.W - 1);
            const float s = fs.p[q];
            const bool ok = vi & (SAMPLE_OUTSIDE || ff.p[q] == FNX_FLUID);
            mn = ok ? fminf(mn, s) : mn;
            mx = ok ? fmaxf(mx, s) : mx;
            any = any | ok;
          }
        }
      }
    }
    d = any ? fmaxf(mn, fminf(mx, d)) : f;
  }
  dst[(size_t)c.b * g.DHW + o] = d;
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

// ---------------------------------------------------------------------------------------------------
// Velocity: SemiLagrangeEulerFluidNetMAC (:388-451), no line trace.
// ---------------------------------------------------------------------------------------------------
template <bool IS3D, bool QUIRKS, int COMP>
__device__ __forceinline__ float sl_mac_component(const GridDims& g, const Field& src, const Field& fu, int i, int j,
                                                  int k, float dt) {
  float v[3];
  get_at_mac<IS3D, QUIRKS, COMP>(g, fu, i, j, k, v);
  const float px = ((float)i + 0.5f) + v[0] * (-dt);
  const float py = ((float)j + 0.5f) + v[1] * (-dt);
  const float pz = ((float)(k + g.zoff) + 0.5f) + v[2] * (-dt);
  return interpol<IS3D>(g, src, COMP, px, py, pz);
}

template <bool IS3D, bool QUIRKS>
__device__ __forceinline__ void sl_mac_cell(const GridDims& g, const CellId& c, float dt, const float* __restrict__ src,
                                                        const float* __restrict__ U, const float* __restrict__ flags,
                                                        float* __restrict__ dst) {
  constexpr int NC = IS3D ? 3 : 2;
  const Field fs{src + (size_t)c.b * NC * g.DHW}, fu{U + (size_t)c.b * NC * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  float r[3] = { 0.f, 0.f, 0.f };
  if (!is_border<IS3D>(g, c.i, c.j, c.k)) {
    if (flags[(size_t)c.b * g.DHW + o] != FNX_FLUID) {
      r[0] = fs.p[(size_t)g.DHW + o];                    // reference writes src channel 1 into channel 0 (:413-416)
      if (IS3D) r[2] = fs.p[(size_t)2 * g.DHW + o];
    } else {
      r[0] = sl_mac_component<IS3D, QUIRKS, 0>(g, fs, fu, c.i, c.j, c.k, dt);
      r[1] = sl_mac_component<IS3D, QUIRKS, 1>(g, fs, fu, c.i, c.j, c.k, dt);
      if (IS3D && !QUIRKS) r[2] = sl_mac_component<IS3D, QUIRKS, 2>(g, fs, fu, c.i, c.j, c.k, dt);   // Q12: 0 in ref
    }
  }
  float* d = dst + (size_t)c.b * NC * g.DHW + o;
#pragma unroll
  for (int a = 0; a < NC; ++a) d[(size_t)a * g.DHW] = r[a];
}

template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void sl_mac_kernel(GridDims g, float dt, const float* __restrict__ src,
                                                        const float* __restrict__ U, const float* __restrict__ flags,
                                                        float* __restrict__ dst) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  sl_mac_cell<IS3D, QUIRKS>(g, c, dt, src, U, flags, dst);
}

// min/max of channel `comp` of orig over the 4(8) corners at trunc(pos -/+ v)  (doClampComponentMAC :500-614)
template <bool IS3D>
__device__ __forceinline__ void clamp_bounds_mac(const GridDims& g, const float* __restrict__ oc, const float pos[3],
                                                 const float v[3], float& mn, float& mx) {
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int qx = (int)(l == 0 ? pos[0] - v[0] : pos[0] + v[0]);
    const int qy = (int)(l == 0 ? pos[1] - v[1] : pos[1] + v[1]);
    const int qz = (int)(l == 0 ? pos[2] - v[2] : pos[2] + v[2]);
    const int i0 = clampi(qx, 0, g.W - 2), j0 = clampi(qy, 0, g.H - 2);
    const int k0 = IS3D ? clampi(clampi(qz, 0, g.Dglob - 2) - g.zoff, 0, g.D - 2) : 0;
    const float* q = oc + (size_t)k0 * g.HW + j0 * g.W + i0;
    // reference visiting order: 000, 100, 010, 110 [, 001, 101, 011, 111] (x is the first digit)
    float o;
    o = q[0]; mn = fminf(mn, o); mx = fmaxf(mx, o);
    o = q[1]; mn = fminf(mn, o); mx = fmaxf(mx, o);
    o = q[g.W]; mn = fminf(mn, o); mx = fmaxf(mx, o);
    o = q[g.W + 1]; mn = fminf(mn, o); mx = fmaxf(mx, o);
    if (IS3D) {
      const float* r = q + g.HW;
      o = r[0]; mn = fminf(mn, o); mx = fmaxf(mx, o);
      o = r[1]; mn = fminf(mn, o); mx = fmaxf(mx, o);
      o = r[g.W]; mn = fminf(mn, o); mx = fmaxf(mx, o);
      o = r[g.W + 1]; mn = fminf(mn, o); mx = fmaxf(mx, o);
    }
  }
}

template <bool IS3D, bool QUIRKS, int COMP>
__device__ __forceinline__ float mac_bwd_correct_clamp(const GridDims& g, const Field& forig, const Field& ffwd,
                                                       const Field& fu, const Field& ff, int i, int j, int k, float dt,
                                                       float half_s, bool fluid) {
  const size_t o = (size_t)k * g.HW + j * g.W + i;
  const float f = ffwd.p[(size_t)COMP * g.DHW + o];
  float v[3];
  get_at_mac<IS3D, QUIRKS, COMP>(g, fu, i, j, k, v);
  // backward pass: SL(fwd, -dt): displacement v * (-(-dt)) == v * dt, also the clamp velocity (:640-648)
  const float vd[3] = { v[0] * dt, v[1] * dt, v[2] * dt };
  float bwd;
  if (!fluid) {
    bwd = COMP == 0 ? ffwd.p[(size_t)g.DHW + o] : (COMP == 1 ? 0.f : f);   // Q1 pass-through of SL(fwd)
  } else if (COMP == 2 && QUIRKS) {
    bwd = 0.f;
  } else {
    bwd = interpol<IS3D>(g, ffwd, COMP, ((float)i + 0.5f) + vd[0], ((float)j + 0.5f) + vd[1], ((float)(k + g.zoff) + 0.5f) + vd[2]);
  }
  // MacCormackCorrectMAC :453-498
  bool skip = !fluid;
  const int idx = COMP == 0 ? i : (COMP == 1 ? j : k + g.zoff);
  if (idx > 0 && !(COMP == 2 && k == 0)) {
    const size_t om = o - (COMP == 0 ? 1 : (COMP == 1 ? g.W : g.HW));
    if (ff.p[om] != FNX_FLUID) skip = true;
  }
  const float corr = skip ? f : f + half_s * (forig.p[(size_t)COMP * g.DHW + o] - bwd);
  float mn = INFINITY, mx = -INFINITY;
  const float pos[3] = { (float)i, (float)j, (float)(k + g.zoff) };
  clamp_bounds_mac<IS3D>(g, forig.p + (size_t)COMP * g.DHW, pos, vd, mn, mx);
  return fmaxf(fminf(corr, mx), mn);
}

template <bool IS3D, bool QUIRKS>
__device__ __forceinline__ void sl_mac_bwd_clamp_cell(const GridDims& g, const CellId& c, float dt, float half_s,
                                                                  const float* __restrict__ orig,
                                                                  const float* __restrict__ fwd,
                                                                  const float* __restrict__ U,
                                                                  const float* __restrict__ flags,
                                                                  float* __restrict__ dst) {
  constexpr int NC = IS3D ? 3 : 2;
  const Field fo{orig + (size_t)c.b * NC * g.DHW}, fw{fwd + (size_t)c.b * NC * g.DHW},
      fu{U + (size_t)c.b * NC * g.DHW}, ff{flags + (size_t)c.b * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  float r[3] = { 0.f, 0.f, 0.f };
  if (!is_border<IS3D>(g, c.i, c.j, c.k)) {
    const bool fluid = ff.p[o] == FNX_FLUID;
    r[0] = mac_bwd_correct_clamp<IS3D, QUIRKS, 0>(g, fo, fw, fu, ff, c.i, c.j, c.k, dt, half_s, fluid);
    r[1] = mac_bwd_correct_clamp<IS3D, QUIRKS, 1>(g, fo, fw, fu, ff, c.i, c.j, c.k, dt, half_s, fluid);
    if (IS3D) r[2] = mac_bwd_correct_clamp<IS3D, QUIRKS, 2>(g, fo, fw, fu, ff, c.i, c.j, c.k, dt, half_s, fluid);
  }
  float* d = dst + (size_t)c.b * NC * g.DHW + o;
#pragma unroll
  for (int a = 0; a < NC; ++a) d[(size_t)a * g.DHW] = r[a];
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
// Phase-ordered ("flat") velocity passes.  The per-component functions above interleave loads and their use: compiled,
// the backward pass is ~60 dependent load -> wait round trips per thread (clusters of 1-13 loads).  Here all components
// go through the same expressions phase by phase: (A) every load that depends on the cell only (own values, -1 flags,
// the 9 face-velocity operands per component), (B) every gather that depends on the face velocities (8 trilinear
// corners and 2 x 8 clamp corners per component) in one batch, (C) the arithmetic.  A non-fluid cell computes its
// (unused) sample anyway -- addresses are clamped, so that is safe -- and takes the reference's pass-through by select.
// Same operations in the same order per value: same bits.  Measured: 2D 1024^2 advection 39 -> 36 us; 3D unchanged
// (1.06 ms at 512x512x64): there the scalar half with its line trace sets the pace.
// ---------------------------------------------------------------------------------------------------
template <bool IS3D>
__device__ __forceinline__ void sl_mac_bwd_clamp_cell_flat(const GridDims& g, const CellId& c, float dt, float half_s,
                                                           const float* __restrict__ orig,
                                                           const float* __restrict__ fwd,
                                                           const float* __restrict__ U,
                                                           const float* __restrict__ flags,
                                                           float* __restrict__ dst) {
  constexpr int NC = IS3D ? 3 : 2, NK = IS3D ? 8 : 4;
  const Field fo{orig + (size_t)c.b * NC * g.DHW}, fw{fwd + (size_t)c.b * NC * g.DHW},
      fu{U + (size_t)c.b * NC * g.DHW}, ff{flags + (size_t)c.b * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  float* d = dst + (size_t)c.b * NC * g.DHW + o;
  if (is_border<IS3D>(g, c.i, c.j, c.k)) {
#pragma unroll
    for (int a = 0; a < NC; ++a) d[(size_t)a * g.DHW] = 0.f;
    return;
  }
  const int i = c.i, j = c.j, k = c.k;
  // ---- (A) loads that depend on the cell only
  const float fcell = ff.p[o];
  float f[NC], og[NC], fmn[NC], v[NC][3];
  bool chk[NC];
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    f[a] = fw.p[(size_t)a * g.DHW + o];
    og[a] = fo.p[(size_t)a * g.DHW + o];
    const int idx = a == 0 ? i : (a == 1 ? j : k + g.zoff);
    chk[a] = idx > 0 && !(a == 2 && k == 0);
    fmn[a] = ff.p[o - (chk[a] ? (a == 0 ? 1 : (a == 1 ? g.W : g.HW)) : 0)];
  }
  get_at_mac<IS3D, false, 0>(g, fu, i, j, k, v[0]);
  get_at_mac<IS3D, false, 1>(g, fu, i, j, k, v[1]);
  if (IS3D) get_at_mac<IS3D, false, 2>(g, fu, i, j, k, v[2]);
  // ---- (B) gathers that depend on the face velocities
  float vd[NC][3];
  Lerp L[NC];
  float Iv[NC][NK], Cv[NC][2][NK];
  const float pos[3] = { (float)i, (float)j, (float)(k + g.zoff) };
#pragma unroll
  for (int a = 0; a < NC; ++a) {
#pragma unroll
    for (int q = 0; q < 3; ++q) vd[a][q] = v[a][q] * dt;
    L[a] = lerp_setup<IS3D>(g, ((float)i + 0.5f) + vd[a][0], ((float)j + 0.5f) + vd[a][1], ((float)(k + g.zoff) + 0.5f) + vd[a][2]);
    const float* q = fw.p + (size_t)a * g.DHW + (size_t)L[a].z0 * g.HW + L[a].y0 * g.W + L[a].x0;
    Iv[a][0] = q[0]; Iv[a][1] = q[g.W]; Iv[a][2] = q[1]; Iv[a][3] = q[g.W + 1];
    if (IS3D) { const float* r = q + g.HW; Iv[a][4] = r[0]; Iv[a][5] = r[g.W]; Iv[a][6] = r[1]; Iv[a][7] = r[g.W + 1]; }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int qx = (int)(l == 0 ? pos[0] - vd[a][0] : pos[0] + vd[a][0]);
      const int qy = (int)(l == 0 ? pos[1] - vd[a][1] : pos[1] + vd[a][1]);
      const int qz = (int)(l == 0 ? pos[2] - vd[a][2] : pos[2] + vd[a][2]);
      const int i0 = clampi(qx, 0, g.W - 2), j0 = clampi(qy, 0, g.H - 2);
      const int k0 = IS3D ? clampi(clampi(qz, 0, g.Dglob - 2) - g.zoff, 0, g.D - 2) : 0;
      const float* p = fo.p + (size_t)a * g.DHW + (size_t)k0 * g.HW + j0 * g.W + i0;
      Cv[a][l][0] = p[0]; Cv[a][l][1] = p[1]; Cv[a][l][2] = p[g.W]; Cv[a][l][3] = p[g.W + 1];
      if (IS3D) { const float* r = p + g.HW; Cv[a][l][4] = r[0]; Cv[a][l][5] = r[1]; Cv[a][l][6] = r[g.W]; Cv[a][l][7] = r[g.W + 1]; }
    }
  }
  // ---- (C) arithmetic (mac_bwd_correct_clamp)
  const bool fluid = fcell == FNX_FLUID;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const float lo = (Iv[a][0] * L[a].t0 + Iv[a][1] * L[a].t1) * L[a].s0 + (Iv[a][2] * L[a].t0 + Iv[a][3] * L[a].t1) * L[a].s1;
    float smp = lo;
    if (IS3D) {
      const float hi = (Iv[a][4] * L[a].t0 + Iv[a][5] * L[a].t1) * L[a].s0 + (Iv[a][6] * L[a].t0 + Iv[a][7] * L[a].t1) * L[a].s1;
      smp = lo * L[a].f0 + hi * L[a].f1;
    }
    const float bwd = fluid ? smp : (a == 0 ? f[1] : (a == 1 ? 0.f : f[a]));      // Q1 pass-through of SL(fwd)
    const bool skip = !fluid | (chk[a] & (fmn[a] != FNX_FLUID));
    const float corr = skip ? f[a] : f[a] + half_s * (og[a] - bwd);
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int q = 0; q < NK; ++q) { mn = fminf(mn, Cv[a][l][q]); mx = fmaxf(mx, Cv[a][l][q]); }
    d[(size_t)a * g.DHW] = fmaxf(fminf(corr, mx), mn);
  }
}

// forward pass (sl_mac_cell), the same way
template <bool IS3D>
__device__ __forceinline__ void sl_mac_cell_flat(const GridDims& g, const CellId& c, float dt, const float* __restrict__ src,
                                                 const float* __restrict__ U, const float* __restrict__ flags,
                                                 float* __restrict__ dst) {
  constexpr int NC = IS3D ? 3 : 2, NK = IS3D ? 8 : 4;
  const Field fs{src + (size_t)c.b * NC * g.DHW}, fu{U + (size_t)c.b * NC * g.DHW};
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  float* d = dst + (size_t)c.b * NC * g.DHW + o;
  if (is_border<IS3D>(g, c.i, c.j, c.k)) {
#pragma unroll
    for (int a = 0; a < NC; ++a) d[(size_t)a * g.DHW] = 0.f;
    return;
  }
  const int i = c.i, j = c.j, k = c.k;
  const float fcell = flags[(size_t)c.b * g.DHW + o];
  float own[NC], v[NC][3];
#pragma unroll
  for (int a = 0; a < NC; ++a) own[a] = fs.p[(size_t)a * g.DHW + o];
  get_at_mac<IS3D, false, 0>(g, fu, i, j, k, v[0]);
  get_at_mac<IS3D, false, 1>(g, fu, i, j, k, v[1]);
  if (IS3D) get_at_mac<IS3D, false, 2>(g, fu, i, j, k, v[2]);
  Lerp L[NC];
  float Iv[NC][NK];
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const float px = ((float)i + 0.5f) + v[a][0] * (-dt);
    const float py = ((float)j + 0.5f) + v[a][1] * (-dt);
    const float pz = ((float)(k + g.zoff) + 0.5f) + v[a][2] * (-dt);
    L[a] = lerp_setup<IS3D>(g, px, py, pz);
    const float* q = fs.p + (size_t)a * g.DHW + (size_t)L[a].z0 * g.HW + L[a].y0 * g.W + L[a].x0;
    Iv[a][0] = q[0]; Iv[a][1] = q[g.W]; Iv[a][2] = q[1]; Iv[a][3] = q[g.W + 1];
    if (IS3D) { const float* r = q + g.HW; Iv[a][4] = r[0]; Iv[a][5] = r[g.W]; Iv[a][6] = r[1]; Iv[a][7] = r[g.W + 1]; }
  }
  const bool fluid = fcell == FNX_FLUID;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const float lo = (Iv[a][0] * L[a].t0 + Iv[a][1] * L[a].t1) * L[a].s0 + (Iv[a][2] * L[a].t0 + Iv[a][3] * L[a].t1) * L[a].s1;
    float smp = lo;
    if (IS3D) {
      const float hi = (Iv[a][4] * L[a].t0 + Iv[a][5] * L[a].t1) * L[a].s0 + (Iv[a][6] * L[a].t0 + Iv[a][7] * L[a].t1) * L[a].s1;
      smp = lo * L[a].f0 + hi * L[a].f1;
    }
    // non-fluid cell: the reference writes src channel 1 into channel 0, 0 into channel 1, src channel 2 into channel 2 (:413-416)
    d[(size_t)a * g.DHW] = fluid ? smp : (a == 0 ? own[1] : (a == 1 ? 0.f : own[2]));
  }
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

// WHAT-dispatch of the templated tile launches: `K` is a generic lambda taking std::integral_constant<int, WHAT>
template <class K> static void with_what(int what, K k) {
  if (what == 3) k(AIC<3>{}); else if (what == 1) k(AIC<1>{}); else k(AIC<2>{});
}

static void launch_fwd_tile(const GridDims& g, int what, bool sample_outside, float dt, const float* rho, const float* U,
                            const float* flags, float* rho_fwd, int* cell, float* U_fwd, float* box, unsigned long long* fix_s,
                            unsigned long long* fix_v, hipStream_t s) {
  int ntx, nty, zchunk; unsigned G;
  tile_launch_geometry(g, ntx, nty, zchunk, G);
  const unsigned nfix = (unsigned)(((size_t)g.B * g.KN * g.H * ntx + 255) / 256);
  with_what(what, [&](auto W) {
    constexpr int WH = decltype(W)::value;
    if (sample_outside) {
      advect3d_fwd_tile_kernel<true, WH><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, (float2*)box, fix_s, fix_v, ntx, nty, zchunk);
      advect3d_fwd_fix_kernel<true><<<dim3(nfix), 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, fix_s, fix_v, ntx);
    } else {
      advect3d_fwd_tile_kernel<false, WH><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, (float2*)box, fix_s, fix_v, ntx, nty, zchunk);
      advect3d_fwd_fix_kernel<false><<<dim3(nfix), 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, fix_s, fix_v, ntx);
    }
  });
}

// which plan launch_advect_fused takes for this grid: 2 = the 2D LDS tiles, 3 = the 3D z-marching tiles, 0 = one thread per cell
// 3D default semantics: the z-marching LDS tile kernels (fnx_advect_march.h); quirks mode and plane ranges beyond the
// tile kernels' 32-bit offsets: one thread per cell
// 2D: LDS tile kernels (fnx_advect_tile2d.h) on grids large enough to pay for the two fix-up launches (measured, advection per
// step, tiles vs one thread per cell: 2048^2 105 vs 129 us, 1024^2 38.6 vs 37.2, 128^2 17.6 vs 12.6), wherever a row offset
// fits the tiles' 32-bit buffer offsets
int advect_tile_plan(const GridDims& g, const GridDims& gfwd, bool is3d, bool quirks, int plan) {
  const bool want_tiles = plan == 1 || plan == 3 || (plan == 0 && (is3d || (size_t)g.HW * g.B >= ((size_t)3 << 19)));
  if (!want_tiles) return 0;
  if (!is3d) return (size_t)g.HW < 0x3fffffffu ? 2 : 0;
  if (quirks) return 0;
  return ((size_t)(gfwd.KN + 2) * gfwd.HW < 0x3fffffffu ? 1 : 0) | ((size_t)(g.KN + 2) * g.HW < 0x3fffffffu ? 4 : 0);   // bit 0: forward, bit 2: backward
}

// MacCormack self-advection of U plus advection of rho by U, both by the OLD U (simulate.py:75-93), in two launches.
// `what`: 3 = both (one step's pair), 1 = the density only (stand-alone advectScalar: U_fwd / U_dst unused), 2 = the velocity only
// (stand-alone advectVel with orig == U: rho / rho_fwd / cell / box / rho_dst unused).  what != 3 needs a tile plan
// (advect_tile_plan(...) != 0 in 2D, both bits in 3D): the per-cell stand-alone launches are the callers' own.
void launch_advect_fused(const GridDims& g, const GridDims& gfwd, bool is3d, bool quirks, bool sample_outside, float dt,
                         float half_s, const float* rho, const float* U, const float* flags, float* rho_fwd, int* cell,
                         float* U_fwd, float* box, float* rho_dst, float* U_dst, unsigned long long* fix, hipStream_t s, int plan,
                         int what) {
  // fix-up bitmaps of the tile kernels: 4 x (one 64-bit word per 64-cell row segment): fwd density, fwd velocity,
  // bwd density, bwd velocity
  const size_t nwords = advect_fix_words(g);
  const dim3 block(BX, BY);
  const bool do_s = what & 1, do_v = what & 2;
  // forward passes and clamp bounds on `gfwd` (the compute window widened by what the backward pass reads)
  const int tp = advect_tile_plan(g, gfwd, is3d, quirks, plan);
  // The 3D backward pass of the pair is ONE march for density and velocity (advect3d_bwd_tile_kernel, round 6): 761-769 against 772-775 us
  // per pair of a developed 512 x 512 x 64 plume in alternating rounds on one box (profiles/r06/b_advect_ab_alternating.txt) -- 1 %: the
  // marches are VALU-issue bound and the fused one issues 96 % of their instructions; the traffic it saves was not the bound.
  // FNX_ADVECT_PLAN_TILES_SPLIT keeps the two separate marches (what the stand-alone advections run) for A/B timing.
  const bool bwd_fused = plan != 3;
  if (!is3d && tp) {
    const int ntx = (g.W + 63) / 64, nty = (g.H + T2R - 1) / T2R;
    const dim3 grid((unsigned)(ntx * nty * g.B));
    const unsigned nfix = (unsigned)(((size_t)g.B * g.H * ntx + 255) / 256);
    unsigned long long *ff_s = do_s ? fix : nullptr, *ff_v = do_v ? fix + nwords : nullptr;
    unsigned long long *fb_s = do_s ? fix + 2 * nwords : nullptr, *fb_v = do_v ? fix + 3 * nwords : nullptr;
    with_what(what, [&](auto W) {
      constexpr int WH = decltype(W)::value;
      if (sample_outside) {
        advect2d_fwd_tile_kernel<true, WH><<<grid, 64 * T2NW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx, nty);
        advect2d_fwd_fix_kernel<true><<<nfix, 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx);
        advect2d_bwd_tile_kernel<true, WH><<<grid, 64 * T2NW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx, nty);
        advect2d_bwd_fix_kernel<true><<<nfix, 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx);
      } else {
        advect2d_fwd_tile_kernel<false, WH><<<grid, 64 * T2NW, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx, nty);
        advect2d_fwd_fix_kernel<false><<<nfix, 256, 0, s>>>(g, dt, rho, U, flags, rho_fwd, cell, U_fwd, ff_s, ff_v, ntx);
        advect2d_bwd_tile_kernel<false, WH><<<grid, 64 * T2NW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx, nty);
        advect2d_bwd_fix_kernel<false><<<nfix, 256, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, rho_dst, U_dst, fb_s, fb_v, ntx);
      }
    });
    return;
  }
  // (the tile kernel also reduces the clamp bounds of the density step from the rho planes it streams)
  if (is3d && (tp & 1)) {
    launch_fwd_tile(gfwd, what, sample_outside, dt, rho, U, flags, rho_fwd, cell, U_fwd, box, do_s ? fix : nullptr, do_v ? fix + nwords : nullptr, s);
  } else {
    DISPATCH3(is3d, quirks, sample_outside, advect_fwd_kernel, <<<cell_grid(gfwd), block, 0, s>>>(gfwd, dt, rho, U, flags, rho_fwd, cell, U_fwd));
    if (is3d) launch_box_minmax(gfwd, sample_outside, rho, flags, box, s);
  }
  if (is3d && (tp & 4)) {
    int ntx, nty, zchunk; unsigned G;
    tile_launch_geometry(g, ntx, nty, zchunk, G);
    unsigned long long* fb_s = do_s ? fix + 2 * nwords : nullptr;
    unsigned long long* fb_v = do_v ? fix + 3 * nwords : nullptr;
    if (what == 3 && bwd_fused) {
      if (sample_outside) advect3d_bwd_tile_kernel<true><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, (const float2*)box, rho_dst, U_dst, fb_s, fb_v, ntx, nty, zchunk);
      else advect3d_bwd_tile_kernel<false><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, U_fwd, flags, (const float2*)box, rho_dst, U_dst, fb_s, fb_v, ntx, nty, zchunk);
    } else {
    if (do_s) {
      if (sample_outside) advect3d_bwd_scalar_tile_kernel<true><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, flags, (const float2*)box, rho_dst, fb_s, ntx, nty, zchunk);
      else advect3d_bwd_scalar_tile_kernel<false><<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, rho, rho_fwd, cell, U, flags, (const float2*)box, rho_dst, fb_s, ntx, nty, zchunk);
    }
    if (do_v) advect3d_bwd_vel_tile_kernel<<<dim3(G), 64 * ATNW, 0, s>>>(g, dt, half_s, U, U_fwd, flags, U_dst, fb_v, ntx, nty, zchunk);
    }
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
