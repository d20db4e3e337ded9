// Radius-1 stencil operators of the fluid step for gfx950: velocityDivergence, velocityUpdate,
// addBuoyancy, setWallBcs, setConstVals, flagsToOccupancy, emptyDomain.
//
// All are HBM-bound streaming kernels (arithmetic intensity < 1 FLOP/B): one thread per cell, x fastest,
// 64x4 blocks so each wave reads/writes 256 contiguous bytes per field row; the +-1 neighbours in x come
// out of the same cache lines, the +-1 rows out of L1/L2.  The reference implements these as 29-92 ATen
// ops each.
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

inline dim3 cell_grid(const GridDims& g) { return dim3((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B * g.KN); }

// velocityDivergence, lib/fluid/velocity_divergence.py:46-74
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void divergence_kernel(GridDims g, const float* __restrict__ U,
                                                            const float* __restrict__ flags, float* __restrict__ div) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* u = U + (size_t)c.b * NC * g.DHW + o;
  float d = 0.f;
  if (!is_border<IS3D>(g, c.i, c.j, c.k)) {
    d = ((u[0] - u[1]) + u[g.DHW]) - u[(size_t)g.DHW + g.W];
    if (IS3D) d = d + (u[(size_t)2 * g.DHW] - u[(size_t)2 * g.DHW + g.HW]);
  }
  if (flags[(size_t)c.b * g.DHW + o] == FNX_OBST) d = 0.f;
  div[(size_t)c.b * g.DHW + o] = d;
}

// velocityUpdate, lib/fluid/velocity_update.py:47-149 (2D); 3D: fluid-fluid faces only
// (solver_cpp/src/projection/update_vel.cpp:58-117 -- the reference's own 3D branch raises)
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void velocity_update_kernel(GridDims g, const float* __restrict__ p,
                                                                 float* __restrict__ U,
                                                                 const float* __restrict__ flags) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid || is_border<IS3D>(g, c.i, c.j, c.k)) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const float* pp = p + (size_t)c.b * g.DHW + o;
  float* u = U + (size_t)c.b * NC * g.DHW + o;
  const float fc = fl[0], P = pp[0];
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const float fm = *(fl - off), Pm = *(pp - off);
    const float uc = u[(size_t)a * g.DHW];
    const float m_ff = (fc == FNX_FLUID && fm == FNX_FLUID) ? 1.f : 0.f;
    float r;
    if (!IS3D) {
      const float m_fe = (fc == FNX_FLUID && fm == FNX_EMPTY) ? 1.f : 0.f;
      const float m_ef = (fc == FNX_EMPTY && fm == FNX_FLUID) ? 1.f : 0.f;
      const float m_nf = (fc == FNX_EMPTY && fm == FNX_EMPTY) ? 1.f : 0.f;
      r = ((m_ff * (uc - (P - Pm)) + m_fe * (uc - P)) + m_ef * (uc + Pm)) + m_nf * 0.f;
    } else {
      r = m_ff * (uc - (P - Pm));
    }
    u[(size_t)a * g.DHW] = r;
  }
}

// addBuoyancy, lib/fluid/source_terms.py:47-116
template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void add_buoyancy_kernel(GridDims g, float* __restrict__ U,
                                                              const float* __restrict__ flags,
                                                              const float* __restrict__ rho, float sx, float sy,
                                                              float sz, float rho_star) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid || is_border<IS3D>(g, c.i, c.j, c.k)) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  if (fl[0] != FNX_FLUID) return;
  const float* r = rho + (size_t)c.b * g.DHW + o;
  float* u = U + (size_t)c.b * NC * g.DHW + o;
  const float rc = r[0];
  if (*(fl - 1) == FNX_FLUID) u[0] = u[0] + sx * ((0.5f * (rc + *(r - 1))) - rho_star);
  if (*(fl - g.W) == FNX_FLUID) u[g.DHW] = u[g.DHW] + sy * ((0.5f * (rc + *(r - g.W))) - rho_star);
  if (IS3D) {
    if (*(fl - g.HW) == FNX_FLUID) {
      float* uz = u + (size_t)2 * g.DHW;
      if (!QUIRKS) *uz = *uz + sz * ((0.5f * (rc + *(r - g.HW))) - rho_star);
      else *uz = *uz + sz * (0.5f * (rc + (c.k + g.zoff <= 1 ? 0.f : *(r - g.HW))));      // source_terms.py:110-114
    }
  }
}

// addGravity, lib/fluid/source_terms.py:122-219
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void add_gravity_kernel(GridDims g, float* __restrict__ U,
                                                             const float* __restrict__ flags, float fx, float fy,
                                                             float fz) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid || is_border<IS3D>(g, c.i, c.j, c.k)) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const float fc = fl[0];
  if (fc != FNX_FLUID && fc != FNX_EMPTY) return;
  float* u = U + (size_t)c.b * NC * g.DHW + o;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const float fm = *(fl - (a == 0 ? 1 : (a == 1 ? g.W : g.HW)));
    if (fm == FNX_FLUID || (fm == FNX_EMPTY && fc == FNX_FLUID)) {
      float* q = u + (size_t)a * g.DHW;
      *q = *q + (a == 0 ? fx : (a == 1 ? fy : fz));
    }
  }
}

// correctScalar, lib/fluid/cpp/advection.py:9-12: src += (dt*0.5) * src * div on fluid cells (left to right: the scalar
// times src, times div, added to src)
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void correct_scalar_kernel(GridDims g, float half_dt, float* __restrict__ src,
                                                                const float* __restrict__ div,
                                                                const float* __restrict__ flags) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  const size_t o = (size_t)c.b * g.DHW + (size_t)c.k * g.HW + c.j * g.W + c.i;
  if (flags[o] != FNX_FLUID) return;
  const float x = src[o];
  const float t = half_dt * x;
  const float u = t * div[o];
  src[o] = x + u;
}

// addViscosity (2D), lib/fluid/viscosity.py:57-70: out = m * (u + coef * (u[i+1] + u[j+1] + u[i-1] + u[i-1,j-1] - 4u)),
// border cells copied; the fourth neighbour is the reference's (i-1, j-1).
__global__ __launch_bounds__(BX* BY) void add_viscosity_kernel(GridDims g, const float* __restrict__ Uin,
                                                               float* __restrict__ Uout,
                                                               const float* __restrict__ flags, float coef) {
  const CellId c = cell_id<false>(g);
  if (!c.valid) return;
  const size_t o = (size_t)c.j * g.W + c.i;
  const float* u = Uin + (size_t)c.b * 2 * g.DHW + o;
  float* w = Uout + (size_t)c.b * 2 * g.DHW + o;
  if (is_border<false>(g, c.i, c.j, c.k)) {
    w[0] = u[0];
    w[g.DHW] = u[g.DHW];
    return;
  }
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const bool fluid = fl[0] == FNX_FLUID;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float* q = u + (size_t)a * g.DHW;
    const float m = (fluid && *(fl - (a == 0 ? 1 : g.W)) == FNX_FLUID) ? 1.f : 0.f;
    float s = q[1] + q[g.W];
    s = s + *(q - 1);
    s = s + *(q - 1 - g.W);
    s = s - (4.f * q[0]);
    w[(size_t)a * g.DHW] = m * (q[0] + coef * s);
  }
}

// setWallBcs, lib/fluid/set_wall_bcs.py:45-84
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void set_wall_bcs_kernel(GridDims g, float* __restrict__ U,
                                                              const float* __restrict__ flags) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const float fc = fl[0];
  if (fc != FNX_FLUID && fc != FNX_OBST) return;
  float* u = U + (size_t)c.b * NC * g.DHW + o;
  const float fx = c.i > 0 ? *(fl - 1) : fc;          // i_l = max(i-1, 0): column 0 sees itself
  const float fy = c.j > 0 ? *(fl - g.W) : fc;
  if (fx == FNX_OBST || (fc == FNX_OBST && fx == FNX_FLUID)) u[0] = 0.f;
  if (fy == FNX_OBST || (fc == FNX_OBST && fy == FNX_FLUID)) u[g.DHW] = 0.f;
  if (IS3D && c.k + g.zoff > 0 && c.k > 0) {
    const float fz = *(fl - g.HW);
    if (fz == FNX_OBST || (fc == FNX_OBST && fz == FNX_FLUID)) u[(size_t)2 * g.DHW] = 0.f;
  }
}

// setWallBcsStick (2D), lib/fluid/set_wall_bcs_stick.py:57-156 with its three unbound names bound (see the header).
// Out of place: every neighbour velocity is the value after the slip phase (:57-77), which is a function of the INPUT
// field and the flags of the neighbour and its own -1 neighbour, so one pass suffices.
#define FNX_STICK 128.0f
__device__ __forceinline__ float stick_slip(const GridDims& g, const float* __restrict__ u, const float* __restrict__ fl,
                                            const float* __restrict__ st, int comp, int j, int i) {
  // u, fl, st: sample base pointers (u: channel 0); (:57) obstacle cells -> 0, (:61-62,73-74) -1 neighbour is an obstacle -> 0
  const size_t o = (size_t)j * g.W + i;
  const float fc = fl[o];
  if (fc == FNX_OBST) return 0.f;
  const bool cont = fc == FNX_FLUID || st[o] == FNX_STICK;
  const bool at0 = comp == 0 ? i <= 0 : j <= 0;
  if (cont && !at0 && fl[o - (comp == 0 ? 1 : g.W)] == FNX_OBST) return 0.f;
  return u[(size_t)comp * g.DHW + o];
}

__global__ __launch_bounds__(BX* BY) void set_wall_bcs_stick_kernel(GridDims g, const float* __restrict__ Uin,
                                                                    float* __restrict__ Uout,
                                                                    const float* __restrict__ flags,
                                                                    const float* __restrict__ stick) {
  const CellId c = cell_id<false>(g);
  if (!c.valid) return;
  const int i = c.i, j = c.j;
  const float* u = Uin + (size_t)c.b * 2 * g.DHW;
  const float* fl = flags + (size_t)c.b * g.DHW;
  const float* st = stick + (size_t)c.b * g.DHW;
  const size_t o = (size_t)j * g.W + i;
  const float fc = fl[o];
  const bool S = st[o] == FNX_STICK;
  const bool cont = fc == FNX_FLUID || fc == FNX_OBST || S;
  float uu = stick_slip(g, u, fl, st, 0, j, i);
  float vv = stick_slip(g, u, fl, st, 1, j, i);
  const int il = i > 0 ? i - 1 : 0, ir = i < g.W - 1 ? i + 1 : g.W - 1;
  const int jl = j > 0 ? j - 1 : 0, jr = j < g.H - 1 ? j + 1 : g.H - 1;
  if (S) {
    // (:97-116) vertical component mirrors the fluid neighbour left / right; the later rule wins, both -> mean
    const bool f_l = fl[(size_t)j * g.W + il] == FNX_FLUID, f_r = fl[(size_t)j * g.W + ir] == FNX_FLUID;
    const float v_l = i > 0 ? stick_slip(g, u, fl, st, 1, j, i - 1) : 0.f;
    const float v_r = i < g.W - 1 ? stick_slip(g, u, fl, st, 1, j, i + 1) : 0.f;
    if (f_l) vv = -v_l;
    if (f_r) vv = -v_r;
    if (f_l && f_r) vv = 0.5f * ((-v_l) - v_r);
    // (:118-136) horizontal component, neighbours below / above; the reference's "both" test reads the lower
    // neighbour twice (:131): the mean is taken whenever the lower neighbour is fluid
    const bool f_d = fl[(size_t)jl * g.W + i] == FNX_FLUID, f_u = fl[(size_t)jr * g.W + i] == FNX_FLUID;
    const float u_d = j > 0 ? stick_slip(g, u, fl, st, 0, j - 1, i) : 0.f;
    const float u_u = j < g.H - 1 ? stick_slip(g, u, fl, st, 0, j + 1, i) : 0.f;
    if (f_d) uu = -u_d;
    if (f_u) uu = -u_u;
    if (f_d) uu = 0.5f * ((-u_d) - u_u);
  }
  // (:138-156) corners; the doubled terms are the reference's sums
  const int ls = cont && st[(size_t)j * g.W + il] == FNX_STICK, rs = cont && st[(size_t)j * g.W + ir] == FNX_STICK;
  const int bs = cont && st[(size_t)jl * g.W + i] == FNX_STICK, us = cont && st[(size_t)jr * g.W + i] == FNX_STICK;
  if (2 * (int)S + 2 * ls + bs + us == 3) uu = 0.f;
  if (2 * (int)S + ls + 2 * bs + rs == 3) vv = 0.f;
  float* w = Uout + (size_t)c.b * 2 * g.DHW + o;
  w[0] = uu;
  w[g.DHW] = vv;
}

// setConstVals, lib/simulate.py:16-25: x = x*inv_mask + bc (two roundings)
__global__ __launch_bounds__(256) void set_const_vals_kernel(size_t n, float* __restrict__ x,
                                                             const float* __restrict__ bc,
                                                             const float* __restrict__ inv_mask) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const float t = x[q] * inv_mask[q];
    x[q] = t + bc[q];
  }
}

// flagsToOccupancy, lib/fluid/flags_to_occupancy.py:14-19
__global__ __launch_bounds__(256) void occupancy_kernel(size_t n, const float* __restrict__ flags,
                                                        float* __restrict__ occ) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const float f = flags[q];
    occ[q] = f == FNX_FLUID ? 0.f : (f == FNX_OBST ? 1.f : f);
  }
}

// emptyDomain, lib/fluid/util.py:5-47
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void empty_domain_kernel(GridDims g, float* __restrict__ flags, int bnd) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  bool border = (c.i < bnd) | (c.i > g.W - 1 - bnd) | (c.j < bnd) | (c.j > g.H - 1 - bnd);
  if (IS3D) border = border | (c.k + g.zoff < bnd) | (c.k + g.zoff > g.Dglob - 1 - bnd);
  flags[(size_t)c.b * g.DHW + (size_t)c.k * g.HW + c.j * g.W + c.i] = border ? FNX_OBST : FNX_FLUID;
}

// createCylinder / createBox2D, lib/fluid/geometry_utils.py:4-34, 36-63: obstacle cells written into flags on every z plane.
// Cylinder: the reference evaluates (X - cx)^2 + (Y - cy)^2 <= r*r on int64 index grids promoted to fp32 (the python
// scalars are cast to fp32 by the tensor iterator): float(i) - cx, squared, summed, compared with float(r*r).
template <bool BOX>
__global__ __launch_bounds__(BX* BY) void geometry_kernel(GridDims g, float* __restrict__ flags, float a0, float a1,
                                                          float a2, float a3) {
  const CellId c = cell_id<true>(g);
  if (!c.valid) return;
  bool inside;
  if (BOX) {
    // a0 <= x < a1, a2 <= y < a3 (the box the reference's docstring describes; its own body cannot run, :59-62)
    const float x = (float)c.i, y = (float)c.j;
    inside = (x >= a0) & (x < a1) & (y >= a2) & (y < a3);
  } else {
    const float dx = (float)c.i - a0, dy = (float)c.j - a1;
    inside = dx * dx + dy * dy <= a2;
  }
  if (inside) flags[(size_t)c.b * g.DHW + (size_t)c.k * g.HW + c.j * g.W + c.i] = FNX_OBST;
}

// getCentered, lib/fluid/grid.py:7-32: cell-centred velocity 0.5 (u_c + u_c(+1)) per component, 0 on the last
// column / row / plane; always 3 output channels (the z channel is 0 in 2D).
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void get_centered_kernel(GridDims g, const float* __restrict__ U,
                                                              float* __restrict__ out) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* u = U + (size_t)c.b * NC * g.DHW + o;
  float* r = out + (size_t)c.b * 3 * g.DHW + o;
  r[0] = c.i < g.W - 1 ? 0.5f * (u[0] + u[1]) : 0.f;
  r[g.DHW] = c.j < g.H - 1 ? 0.5f * (u[g.DHW] + u[(size_t)g.DHW + g.W]) : 0.f;
  float z = 0.f;
  if (IS3D) z = c.k < g.D - 1 ? 0.5f * (u[(size_t)2 * g.DHW] + u[(size_t)2 * g.DHW + g.HW]) : 0.f;
  r[(size_t)2 * g.DHW] = z;
}

// ---- adjoints of the linear stencil operators (the training graph differentiates through velocityUpdate -> setWallBcs ->
// velocityDivergence: lib/model.py:190-227, fluid_net_train.py:366; the reference gets them from autograd over its ATen
// chains).  setWallBcs zeroes a flag-dependent set of entries, so its adjoint is itself.

// d(c) = A(c) [ sum_a u_a(c) - u_a(c + e_a) ],  A(c) = interior and not an obstacle (velocity_divergence.py:46-74)
//   => dL/du_a(q) = A(q) g(q) - A(q - e_a) g(q - e_a)
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void divergence_bwd_kernel(GridDims g, const float* __restrict__ gdiv,
                                                                const float* __restrict__ flags, float* __restrict__ gU) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const float* gd = gdiv + (size_t)c.b * g.DHW + o;
  float* out = gU + (size_t)c.b * NC * g.DHW + o;
  auto active = [&](int i, int j, int k, const float* f) { return !is_border<IS3D>(g, i, j, k) && *f != FNX_OBST; };
  const float own = active(c.i, c.j, c.k, fl) ? gd[0] : 0.f;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const int coord = a == 0 ? c.i : (a == 1 ? c.j : c.k);
    float prev = 0.f;
    if (coord >= 1 && active(c.i - (a == 0), c.j - (a == 1), c.k - (a == 2), fl - off)) prev = *(gd - off);
    out[(size_t)a * g.DHW] = own - prev;
  }
}

// velocityUpdate (velocity_update.py:47-149): on interior cells u_a <- m_ff (u_a - (p - p_-)) + m_fe (u_a - p) + m_ef (u_a + p_-)
// (3D: the m_ff term only), border cells untouched.  With g = dL/du_out:
//   dL/du_a(c) = border ? g_a(c) : (m_ff + m_fe + m_ef)_a(c) g_a(c)
//   dL/dp(c)   = sum_a [ -(m_ff + m_fe)_a(c) g_a(c) ]_{c interior} + sum_a [ (m_ff + m_ef)_a(c + e_a) g_a(c + e_a) ]_{c + e_a interior}
template <bool IS3D>
__global__ __launch_bounds__(BX* BY) void velocity_update_bwd_kernel(GridDims g, const float* __restrict__ gout,
                                                                     const float* __restrict__ flags,
                                                                     float* __restrict__ gU, float* __restrict__ gp) {
  const CellId c = cell_id<IS3D>(g);
  if (!c.valid) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)c.k * g.HW + c.j * g.W + c.i;
  const float* fl = flags + (size_t)c.b * g.DHW + o;
  const float* go = gout + (size_t)c.b * NC * g.DHW + o;
  float* gu = gU + (size_t)c.b * NC * g.DHW + o;
  const bool inner = !is_border<IS3D>(g, c.i, c.j, c.k);
  const float fc = fl[0];
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const float ga = go[(size_t)a * g.DHW];
    float gua = ga;
    if (inner) {
      const float fm = *(fl - off);
      const bool ff = fc == FNX_FLUID && fm == FNX_FLUID;
      const bool fe = !IS3D && fc == FNX_FLUID && fm == FNX_EMPTY, ef = !IS3D && fc == FNX_EMPTY && fm == FNX_FLUID;
      gua = (ff | fe | ef) ? ga : 0.f;
      if (ff | fe) acc = acc - ga;
    }
    gu[(size_t)a * g.DHW] = gua;
  }
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const int ni = c.i + (a == 0), nj = c.j + (a == 1), nk = c.k + (a == 2);
    if (ni < g.W && nj < g.H && nk < g.D && !is_border<IS3D>(g, ni, nj, nk)) {
      const float fn = *(fl + off);                      // the neighbour's cell type; its -1 neighbour along a is this cell
      const bool ff = fn == FNX_FLUID && fc == FNX_FLUID, ef = !IS3D && fn == FNX_EMPTY && fc == FNX_FLUID;
      if (ff | ef) acc = acc + *(go + (size_t)a * g.DHW + off);
    }
  }
  gp[(size_t)c.b * g.DHW + o] = acc;
}

inline int stream_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

}  // namespace

namespace fnx {

void launch_divergence(const GridDims& g, bool is3d, const float* U, const float* flags, float* div, hipStream_t s) {
  if (is3d) divergence_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, div);
  else divergence_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, div);
}

void launch_velocity_update(const GridDims& g, bool is3d, const float* p, float* U, const float* flags, hipStream_t s) {
  if (is3d) velocity_update_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, p, U, flags);
  else velocity_update_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, p, U, flags);
}

void launch_add_buoyancy(const GridDims& g, bool is3d, bool quirks, float* U, const float* flags, const float* rho,
                         float sx, float sy, float sz, float rho_star, hipStream_t s) {
  if (is3d) {
    if (quirks) add_buoyancy_kernel<true, true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, rho, sx, sy, sz, rho_star);
    else add_buoyancy_kernel<true, false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, rho, sx, sy, sz, rho_star);
  } else {
    add_buoyancy_kernel<false, false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, rho, sx, sy, sz, rho_star);
  }
}

void launch_add_gravity(const GridDims& g, bool is3d, float* U, const float* flags, float fx, float fy, float fz,
                        hipStream_t s) {
  if (is3d) add_gravity_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, fx, fy, fz);
  else add_gravity_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags, fx, fy, fz);
}

void launch_correct_scalar(const GridDims& g, bool is3d, float half_dt, float* src, const float* div, const float* flags,
                           hipStream_t s) {
  if (is3d) correct_scalar_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, half_dt, src, div, flags);
  else correct_scalar_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, half_dt, src, div, flags);
}

void launch_add_viscosity(const GridDims& g, const float* Uin, float* Uout, const float* flags, float coef,
                          hipStream_t s) {
  add_viscosity_kernel<<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, Uin, Uout, flags, coef);
}

void launch_set_wall_bcs_stick(const GridDims& g, const float* Uin, float* Uout, const float* flags, const float* stick,
                               hipStream_t s) {
  set_wall_bcs_stick_kernel<<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, Uin, Uout, flags, stick);
}

void launch_set_wall_bcs(const GridDims& g, bool is3d, float* U, const float* flags, hipStream_t s) {
  if (is3d) set_wall_bcs_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags);
  else set_wall_bcs_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, flags);
}

void launch_set_const_vals(size_t n, float* x, const float* bc, const float* inv_mask, hipStream_t s) {
  set_const_vals_kernel<<<stream_blocks(n), 256, 0, s>>>(n, x, bc, inv_mask);
}

void launch_flags_to_occupancy(size_t n, const float* flags, float* occ, hipStream_t s) {
  occupancy_kernel<<<stream_blocks(n), 256, 0, s>>>(n, flags, occ);
}

// max |x| over n floats into *out (a non-negative float; its bit pattern orders like an unsigned integer).  The caller
// zeroes *out.  NaNs are ignored (v_max semantics).
__global__ __launch_bounds__(256) void max_abs_kernel(size_t n4, size_t n, const float* __restrict__ x, unsigned* __restrict__ out) {
  float m = 0.f;
  const float4* x4 = (const float4*)x;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  auto amax4 = [](const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  for (; q + 3 * stride < n4; q += 4 * stride) {          // four loads in flight per thread (a maximum does not care about the order)
    const float4 a = x4[q], b = x4[q + stride], c = x4[q + 2 * stride], d = x4[q + 3 * stride];
    m = fmaxf(m, fmaxf(fmaxf(amax4(a), amax4(b)), fmaxf(amax4(c), amax4(d))));
  }
  for (; q < n4; q += stride) m = fmaxf(m, amax4(x4[q]));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

void launch_max_abs(size_t n, const float* x, float* out, hipStream_t s) {
  size_t nb = (n / 4 + 256 * 8 - 1) / (256 * 8);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  max_abs_kernel<<<(unsigned)nb, 256, 0, s>>>(n / 4, n, x, (unsigned*)out);
}

void launch_empty_domain(const GridDims& g, bool is3d, float* flags, int bnd, hipStream_t s) {
  if (is3d) empty_domain_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, flags, bnd);
  else empty_domain_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, flags, bnd);
}

void launch_create_cylinder(const GridDims& g, float* flags, float cx, float cy, float r2, hipStream_t s) {
  geometry_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, flags, cx, cy, r2, 0.f);
}

void launch_create_box2d(const GridDims& g, float* flags, float x0, float x1, float y0, float y1, hipStream_t s) {
  geometry_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, flags, x0, x1, y0, y1);
}

void launch_get_centered(const GridDims& g, bool is3d, const float* U, float* out, hipStream_t s) {
  if (is3d) get_centered_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, out);
  else get_centered_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, U, out);
}

void launch_divergence_bwd(const GridDims& g, bool is3d, const float* gdiv, const float* flags, float* gU, hipStream_t s) {
  if (is3d) divergence_bwd_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, gdiv, flags, gU);
  else divergence_bwd_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, gdiv, flags, gU);
}

void launch_velocity_update_bwd(const GridDims& g, bool is3d, const float* gout, const float* flags, float* gU, float* gp,
                                hipStream_t s) {
  if (is3d) velocity_update_bwd_kernel<true><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, gout, flags, gU, gp);
  else velocity_update_bwd_kernel<false><<<cell_grid(g), dim3(BX, BY), 0, s>>>(g, gout, flags, gU, gp);
}

}  // namespace fnx
