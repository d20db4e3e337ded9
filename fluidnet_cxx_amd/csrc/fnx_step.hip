// Fused stages of one time step (lib/simulate.py:96-171) for gfx950.
//
// The reference runs setConstVals -> addBuoyancy -> setWallBcs -> setConstVals -> velocityDivergence as ~250
// ATen ops; as separate HIP kernels they are 9 launches that each stream the same 4-24 MiB fields.  Here:
//   stage2d_div_kernel    : U_adv, rho_adv (advection outputs) -> U, rho (BCs, buoyancy, wall BCs applied) and div; each
//                           thread produces its own cell and re-derives the +1 neighbours' velocity components the
//                           divergence needs (radius-1 recompute instead of a second pass over HBM)
//   stage3d_kernel        : the same in 3D (round 5; until then staging only, followed by a plain divergence pass: as a branchy
//                           kernel the recompute's 54 dependent loads per cell had measured slower)
//   post_projection_kernel: U -= grad p, wall BCs, BCs                      (simulate.py:154-168)
// Per-cell arithmetic is exactly the sequence of the separate operators (fnx_stencils.hip), so results are bit-identical
// to the unfused path.  The stages of one velocity component `a` of a cell, in order (what the comments below call "the
// stage sequence"): setConstVals (simulate.py:96), addBuoyancy on interior fluid cells whose -1 neighbour is fluid
// (source_terms.py:47-116), setWallBcs (set_wall_bcs.py:45-84; skipped before the convnet, simulate.py:120), setConstVals
// (simulate.py:133).
#include "fnx_device.h"
#include "fnx_kernels.h"

namespace {

constexpr int BX = 64, BY = 4;

struct StepPtrs {
  const float* U_adv; const float* rho_adv;          // advected fields (rho_adv may be null)
  const float* flags;
  const float* UBC; const float* UBCInvMask;         // may be null
  const float* rhoBC; const float* rhoBCInvMask;     // may be null
  float* U; float* rho; float* div;
  const unsigned char* cls;                          // optional BC class map (bit 0: velocity BCs are x*1+0, bit 1: density)
  int grav; float gx, gy, gz;                        // addGravity after the buoyancy (simulate.py:107-114); strengths = gravity * dt
  int bc2;                                           // 0: leave out the second setConstVals (simulate.py:133) -- the caller runs
                                                     // setWallBcsStick between the two (simulate.py:129-133)
};

// addGravity's condition for one component of a non-border cell (source_terms.py:122-219; add_gravity_kernel)
__device__ __forceinline__ bool gravity_applies(float fc, float fm) {
  return (fc == FNX_FLUID || fc == FNX_EMPTY) && (fm == FNX_FLUID || (fm == FNX_EMPTY && fc == FNX_FLUID));
}

// The stage sequence of component `a` of one 3D cell: u its advected value, (um, uc) its velocity-BC entry, fc / fm the flags of
// the cell and of its -1 neighbour along a (the cell's own where that neighbour does not exist), r0 / r1 their densities with
// the BC entries (rm0, rc0) / (rm1, rc1), kg = k + zoff the cell's global plane, k its local one.
template <bool QUIRKS, bool WALL>
__device__ __forceinline__ float stage_eval3d(int a, float v, bool ubc, float um, float uc, float fc, float fm, bool border, bool buoy,
                                              float r0, float r1, bool rbc, float rm0, float rc0, float rm1, float rc1, float s_a,
                                              float rho_star, bool grav, float g_a, bool bc2, int kg, int k) {
  if (ubc) { const float t = v * um; v = t + uc; }                                         // simulate.py:96
  if (buoy && !border && fc == FNX_FLUID && fm == FNX_FLUID) {                              // source_terms.py
    float q0 = r0, q1 = r1;
    if (rbc) { float t = q0 * rm0; q0 = t + rc0; t = q1 * rm1; q1 = t + rc1; }
    if (a == 2 && QUIRKS) v = v + s_a * (0.5f * (q0 + (kg <= 1 ? 0.f : q1)));
    else v = v + s_a * ((0.5f * (q0 + q1)) - rho_star);
  }
  if (grav && !border && gravity_applies(fc, fm)) v = v + g_a;                             // source_terms.py:122-219
  if (WALL && (fc == FNX_FLUID || fc == FNX_OBST)) {                                        // set_wall_bcs.py:45-84
    if (!(a == 2 && (kg == 0 || k == 0))) {
      if (fm == FNX_OBST || (fc == FNX_OBST && fm == FNX_FLUID)) v = 0.f;
    }
  }
  if (ubc && bc2) { const float t = v * um; v = t + uc; }                                  // simulate.py:133
  return v;
}

// The 3D staging pass as straight-line code: every load of a cell -- its three advected velocity components, the density
// and the flags of the cell and of its three -1 neighbours, and, unless the whole wave is in identity BC cells, the BC
// arrays of those cells -- is issued unconditionally up front (a -1 neighbour that does not exist reads the cell itself)
// and the stage conditions become selects.  (With the loads nested inside the conditions the kernel was a chain of load ->
// wait -> branch round trips: 58 waits, 257 us at 512x512x64 whether or not the BC loads were skipped; now 120 us.)
// DIV: the pass also writes velocityDivergence of the staged field (velocity_divergence.py:46-74; divergence_kernel's
// expression) for the planes below `kdiv`: the cell re-derives the staged component a of its +1 neighbour along a -- that
// neighbour's advected value, flags, density and BC entry are the nine to fifteen extra loads, all of them lines some
// neighbouring thread loads as its own -- instead of a second pass reading the staged field back (round 1 measured the fused
// form slower, 0.67 against 0.59 ms at 256^3: that was the branchy kernel with 54 dependent loads; in the straight-line form
// the divergence pass's 75 us at 512x512x64 become ~15).
template <bool QUIRKS, bool WALL, bool DIV>
__global__ __launch_bounds__(BX* BY) void stage3d_kernel(GridDims g, StepPtrs P, int buoy, float sx, float sy, float sz,
                                                         float rho_star, int kdiv) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int bk = blockIdx.z;
  const int b = bk / g.KN, k = g.K0 + (bk - b * g.KN);
  if (i >= g.W || j >= g.H) return;
  const size_t o = (size_t)k * g.HW + j * g.W + i, os = (size_t)b * g.DHW + o;
  const int off[3] = { i > 0 ? 1 : 0, j > 0 ? g.W : 0, k > 0 ? g.HW : 0 };     // 0: "the cell itself" (a missing neighbour counts as the cell's own type)
  const bool want_div = DIV && k < kdiv;                                        // (block-uniform)
  const int offp[3] = { i + 1 < g.W ? 1 : 0, j + 1 < g.H ? g.W : 0, k + 1 < g.D ? g.HW : 0 };   // +1 neighbours (DIV); 0 where there is none: never used
  const bool has_rho = P.rho_adv != nullptr, ubc = P.UBC != nullptr, rbc = P.rhoBC != nullptr;
  bool ident = false;
  if (P.cls) {
    bool mine = P.cls[os] == 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) mine = mine & ((P.cls[os - off[a]] & 2) != 0);
    if (want_div) {
#pragma unroll
      for (int a = 0; a < 3; ++a) mine = mine & (P.cls[os + offp[a]] == 3);
    }
    ident = __builtin_amdgcn_ballot_w64(!mine) == 0;
  }
  // ---- loads
  const float fc = P.flags[os];
  float fm[3], u[3], r1[3] = { 0.f, 0.f, 0.f }, r0 = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) { fm[a] = P.flags[os - off[a]]; u[a] = P.U_adv[((size_t)b * 3 + a) * g.DHW + o]; }
  if (has_rho) {
    r0 = P.rho_adv[os];
#pragma unroll
    for (int a = 0; a < 3; ++a) r1[a] = P.rho_adv[os - off[a]];
  }
  float fp[3] = { 0.f, 0.f, 0.f }, up[3] = { 0.f, 0.f, 0.f }, rp[3] = { 0.f, 0.f, 0.f };
  if (want_div) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { fp[a] = P.flags[os + offp[a]]; up[a] = P.U_adv[((size_t)b * 3 + a) * g.DHW + o + offp[a]]; }
    if (has_rho) {
#pragma unroll
      for (int a = 0; a < 3; ++a) rp[a] = P.rho_adv[os + offp[a]];
    }
  }
  float um[3] = { 1.f, 1.f, 1.f }, uc[3] = { 0.f, 0.f, 0.f }, rm0 = 1.f, rc0 = 0.f, rm1[3] = { 1.f, 1.f, 1.f }, rc1[3] = { 0.f, 0.f, 0.f };
  float ump[3] = { 1.f, 1.f, 1.f }, ucp[3] = { 0.f, 0.f, 0.f }, rmp[3] = { 1.f, 1.f, 1.f }, rcp[3] = { 0.f, 0.f, 0.f };
  if (!ident) {
    if (ubc) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { um[a] = P.UBCInvMask[((size_t)b * 3 + a) * g.DHW + o]; uc[a] = P.UBC[((size_t)b * 3 + a) * g.DHW + o]; }
      if (want_div) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { ump[a] = P.UBCInvMask[((size_t)b * 3 + a) * g.DHW + o + offp[a]]; ucp[a] = P.UBC[((size_t)b * 3 + a) * g.DHW + o + offp[a]]; }
      }
    }
    if (rbc && has_rho) {
      rm0 = P.rhoBCInvMask[os]; rc0 = P.rhoBC[os];
#pragma unroll
      for (int a = 0; a < 3; ++a) { rm1[a] = P.rhoBCInvMask[os - off[a]]; rc1[a] = P.rhoBC[os - off[a]]; }
      if (want_div) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { rmp[a] = P.rhoBCInvMask[os + offp[a]]; rcp[a] = P.rhoBC[os + offp[a]]; }
      }
    }
  }
  // ---- the stage sequence, per component
  const bool border = is_border<true>(g, i, j, k);
  const float sa[3] = { sx, sy, sz };
  const float ga[3] = { P.gx, P.gy, P.gz };
  const bool by = buoy && has_rho, gr = P.grav && has_rho, rb = rbc && has_rho;
  float un[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    un[a] = stage_eval3d<QUIRKS, WALL>(a, u[a], ubc, um[a], uc[a], fc, fm[a], border, by, r0, r1[a], rb, rm0, rc0, rm1[a], rc1[a], sa[a],
                                       rho_star, gr, ga[a], P.bc2 != 0, k + g.zoff, k);
  float rnew = r0;
  if (has_rho && rbc) {
    float t = rnew * rm0; rnew = t + rc0;       // simulate.py:96
    if (P.bc2) { t = rnew * rm0; rnew = t + rc0; }   // simulate.py:133
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) P.U[((size_t)b * 3 + a) * g.DHW + o] = un[a];
  if (has_rho) P.rho[os] = rnew;
  if (want_div) {
    float d = 0.f;
    if (!border) {
      // the staged component a of the +1 neighbour along a: its -1 neighbour along a is this cell
      float vp[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const bool bp = is_border<true>(g, i + (a == 0), j + (a == 1), k + (a == 2));
        vp[a] = stage_eval3d<QUIRKS, WALL>(a, up[a], ubc, ump[a], ucp[a], fp[a], fc, bp, by, rp[a], r0, rb, rmp[a], rcp[a], rm0, rc0, sa[a],
                                           rho_star, gr, ga[a], P.bc2 != 0, k + (a == 2) + g.zoff, k + (a == 2));
      }
      d = ((un[0] - vp[0]) + un[1]) - vp[1];
      d = d + (un[2] - vp[2]);
    }
    if (fc == FNX_OBST) d = 0.f;
    P.div[os] = d;
  }
}

// The 2D fused stage (BCs, buoyancy, wall BCs, BCs and -div in one pass) the same way: the cell needs the staged
// velocity of itself, of u_x at (i+1, j) and of u_y at (i, j+1); all the loads those four evaluations make -- flags and
// density at the cell and its four neighbours, four advected velocity values and, unless the whole wave is in identity
// BC cells, their BC entries -- are issued up front (clamped indices where a neighbour does not exist; such values are
// never used) and the stage conditions become selects.
template <bool WALL>
__device__ __forceinline__ float stage_eval2d(int a, float u, bool ubc, float um, float uc, float fc, float fm, bool border,
                                              bool buoy, float r0, float r1, bool rbc, float rm0, float rc0, float rm1,
                                              float rc1, float s_a, float rho_star, bool grav, float g_a, bool bc2) {
  if (ubc) { const float t = u * um; u = t + uc; }                                           // simulate.py:96
  if (buoy && !border && fc == FNX_FLUID && fm == FNX_FLUID) {                               // source_terms.py
    if (rbc) { float t = r0 * rm0; r0 = t + rc0; t = r1 * rm1; r1 = t + rc1; }
    u = u + s_a * ((0.5f * (r0 + r1)) - rho_star);
  }
  if (grav && !border && gravity_applies(fc, fm)) u = u + g_a;                               // source_terms.py:122-219
  if (WALL && (fc == FNX_FLUID || fc == FNX_OBST)) {                                         // set_wall_bcs.py:45-84
    if (fm == FNX_OBST || (fc == FNX_OBST && fm == FNX_FLUID)) u = 0.f;
  }
  if (ubc && bc2) { const float t = u * um; u = t + uc; }                                    // simulate.py:133
  (void)a;
  return u;
}

template <bool WALL>
__global__ __launch_bounds__(BX* BY) void stage2d_div_kernel(GridDims g, StepPtrs P, int buoy_, float sx, float sy,
                                                             float rho_star) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int b = blockIdx.z;
  if (i >= g.W || j >= g.H) return;
  const size_t o = (size_t)j * g.W + i, os = (size_t)b * g.DHW + o;
  const bool has_rho = P.rho_adv != nullptr, ubc = P.UBC != nullptr, rbc = P.rhoBC != nullptr && has_rho;
  const bool buoy = buoy_ != 0 && has_rho;
  const bool grav = P.grav != 0 && has_rho, bc2 = P.bc2 != 0;
  // neighbour offsets; 0 where the neighbour does not exist (a missing -1 neighbour counts as the cell's own type; the
  // +1 values of a border cell are never used)
  const int xm = i > 0 ? 1 : 0, ym = j > 0 ? g.W : 0, xp = i < g.W - 1 ? 1 : 0, yp = j < g.H - 1 ? g.W : 0;
  bool ident = false;
  if (P.cls) {
    const bool mine = (P.cls[os] == 3) & ((P.cls[os - xm] & 2) != 0) & ((P.cls[os - ym] & 2) != 0) & (P.cls[os + xp] == 3) &
                      (P.cls[os + yp] == 3);
    ident = __builtin_amdgcn_ballot_w64(!mine) == 0;
  }
  // ---- loads
  const float F00 = P.flags[os], Fm0 = P.flags[os - xm], F0m = P.flags[os - ym], Fp0 = P.flags[os + xp], F0p = P.flags[os + yp];
  const size_t o0 = ((size_t)b * 2 + 0) * g.DHW + o, o1 = ((size_t)b * 2 + 1) * g.DHW + o;
  const float u0 = P.U_adv[o0], u1 = P.U_adv[o1], u0p = P.U_adv[o0 + xp], u1p = P.U_adv[o1 + yp];
  float R00 = 0.f, Rm0 = 0.f, R0m = 0.f, Rp0 = 0.f, R0p = 0.f;
  if (has_rho) { R00 = P.rho_adv[os]; Rm0 = P.rho_adv[os - xm]; R0m = P.rho_adv[os - ym]; Rp0 = P.rho_adv[os + xp]; R0p = P.rho_adv[os + yp]; }
  float m0 = 1.f, c0 = 0.f, m1 = 1.f, c1 = 0.f, m0p = 1.f, c0p = 0.f, m1p = 1.f, c1p = 0.f;
  float rm00 = 1.f, rc00 = 0.f, rmm0 = 1.f, rcm0 = 0.f, rm0m = 1.f, rc0m = 0.f, rmp0 = 1.f, rcp0 = 0.f, rm0p = 1.f, rc0p = 0.f;
  if (!ident) {
    if (ubc) {
      m0 = P.UBCInvMask[o0]; c0 = P.UBC[o0]; m1 = P.UBCInvMask[o1]; c1 = P.UBC[o1];
      m0p = P.UBCInvMask[o0 + xp]; c0p = P.UBC[o0 + xp]; m1p = P.UBCInvMask[o1 + yp]; c1p = P.UBC[o1 + yp];
    }
    if (rbc) {
      rm00 = P.rhoBCInvMask[os]; rc00 = P.rhoBC[os]; rmm0 = P.rhoBCInvMask[os - xm]; rcm0 = P.rhoBC[os - xm];
      rm0m = P.rhoBCInvMask[os - ym]; rc0m = P.rhoBC[os - ym]; rmp0 = P.rhoBCInvMask[os + xp]; rcp0 = P.rhoBC[os + xp];
      rm0p = P.rhoBCInvMask[os + yp]; rc0p = P.rhoBC[os + yp];
    }
  }
  // ---- the four staged values
  const bool border = is_border<false>(g, i, j, 0);
  const float v0 = stage_eval2d<WALL>(0, u0, ubc, m0, c0, F00, Fm0, border, buoy, R00, Rm0, rbc, rm00, rc00, rmm0, rcm0, sx, rho_star, grav, P.gx, bc2);
  const float v1 = stage_eval2d<WALL>(1, u1, ubc, m1, c1, F00, F0m, border, buoy, R00, R0m, rbc, rm00, rc00, rm0m, rc0m, sy, rho_star, grav, P.gy, bc2);
  float rnew = R00;
  if (rbc) { float t = rnew * rm00; rnew = t + rc00; if (bc2) { t = rnew * rm00; rnew = t + rc00; } }    // simulate.py:96, :133
  float d = 0.f;
  if (P.div) {
    if (!border) {
      const float v0p = stage_eval2d<WALL>(0, u0p, ubc, m0p, c0p, Fp0, F00, is_border<false>(g, i + 1, j, 0), buoy, Rp0, R00, rbc,
                                           rmp0, rcp0, rm00, rc00, sx, rho_star, grav, P.gx, bc2);
      const float v1p = stage_eval2d<WALL>(1, u1p, ubc, m1p, c1p, F0p, F00, is_border<false>(g, i, j + 1, 0), buoy, R0p, R00, rbc,
                                           rm0p, rc0p, rm00, rc00, sy, rho_star, grav, P.gy, bc2);
      d = ((v0 - v0p) + v1) - v1p;
    }
    if (F00 == FNX_OBST) d = 0.f;
  }
  P.U[o0] = v0;
  P.U[o1] = v1;
  if (has_rho) P.rho[os] = rnew;
  if (P.div) P.div[os] = d;
}

// velocityUpdate + setWallBcs + setConstVals (simulate.py:154-168), in place on U (and rho for the BC re-imposition).
// With `scale` (the convnet branch: model.py:213-226 then simulate.py:168) the same pass is the tail of FluidNet.forward: U holds
// the unnormalised velocity and p the net's output for U / s, so u = U / s goes into the update, the updated u and the pressure are
// multiplied by s again (p_scaled receives p * s), then wall BCs and BCs -- the operators' own arithmetic in their own order.
template <bool IS3D, bool SCALE>
__global__ __launch_bounds__(BX* BY) void post_projection_kernel(GridDims g, const float* __restrict__ p,
                                                                 float* __restrict__ U, float* __restrict__ rho,
                                                                 const float* __restrict__ flags,
                                                                 const float* __restrict__ UBC,
                                                                 const float* __restrict__ UBCInvMask,
                                                                 const float* __restrict__ rhoBC,
                                                                 const float* __restrict__ rhoBCInvMask,
                                                                 const unsigned char* __restrict__ cls, int rho_done,
                                                                 const float* __restrict__ scale, float* __restrict__ p_scaled) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int bk = blockIdx.z;
  const int b = IS3D ? bk / g.KN : bk, k = IS3D ? g.K0 + (bk - b * g.KN) : 0;
  if (i >= g.W || j >= g.H) return;
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)k * g.HW + j * g.W + i, os = (size_t)b * g.DHW + o;
  // Straight-line like stage3d_kernel: every load is issued up front with an address that is valid for every lane (a missing -1
  // neighbour reads the cell itself; BC entries are loaded for the whole wave when any lane needs them) and the conditions become
  // selects.  (Nested in the conditions they were a chain of load -> wait -> branch round trips: 30 waits for 20 loads.)
  const float fc = flags[os], P = p[os];
  const unsigned cl = cls ? cls[os] : 0u;
  const bool border = is_border<IS3D>(g, i, j, k);
  const float sc = SCALE ? scale[b] : 1.f;
  float fm[NC], Pm[NC], u[NC], bm[NC], bc[NC];
  size_t ou[NC];
  const bool need_u = UBC != nullptr && !(cl & 1);
  const bool load_u = __builtin_amdgcn_ballot_w64(need_u) != 0;
  const bool do_rho = rho != nullptr && rhoBC != nullptr && !(rho_done && (cl & 2));
  const bool need_r = do_rho && !(cl & 2);
  const bool load_rho = __builtin_amdgcn_ballot_w64(do_rho) != 0, load_r = __builtin_amdgcn_ballot_w64(need_r) != 0;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    ou[a] = ((size_t)b * NC + a) * g.DHW + o;
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const int idx = a == 0 ? i : (a == 1 ? j : k);
    const int offs = idx > 0 ? off : 0;                    // (idx == 0 is a border cell: fm counts as fc, Pm is not used)
    fm[a] = flags[os - offs];
    Pm[a] = p[os - offs];
    u[a] = U[ou[a]];
    bm[a] = 1.f; bc[a] = 0.f;
  }
  if (load_u) {
#pragma unroll
    for (int a = 0; a < NC; ++a) { bm[a] = UBCInvMask[ou[a]]; bc[a] = UBC[ou[a]]; }
  }
  float r0 = 0.f, rm = 1.f, rc = 0.f;
  if (load_rho) r0 = rho[os];
  if (load_r) { rm = rhoBCInvMask[os]; rc = rhoBC[os]; }
  if (SCALE) p_scaled[os] = P * sc;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    float v = u[a];
    if (SCALE) v = v / sc;                                 // model.py:129-168: the net saw U / s
    {     // velocity_update.py:47-149 (border cells keep their value)
      const float m_ff = (fc == FNX_FLUID && fm[a] == FNX_FLUID) ? 1.f : 0.f;
      float w;
      if (!IS3D) {
        const float m_fe = (fc == FNX_FLUID && fm[a] == FNX_EMPTY) ? 1.f : 0.f;
        const float m_ef = (fc == FNX_EMPTY && fm[a] == FNX_FLUID) ? 1.f : 0.f;
        const float m_nf = (fc == FNX_EMPTY && fm[a] == FNX_EMPTY) ? 1.f : 0.f;
        w = ((m_ff * (v - (P - Pm[a])) + m_fe * (v - P)) + m_ef * (v + Pm[a])) + m_nf * 0.f;
      } else {
        w = m_ff * (v - (P - Pm[a]));
      }
      v = border ? v : w;
    }
    if (SCALE) v = v * sc;                                 // model.py:221-223
    {
      const bool zface = a == 2 && (k + g.zoff == 0 || k == 0);
      const bool wall = (fc == FNX_FLUID || fc == FNX_OBST) && !zface && (fm[a] == FNX_OBST || (fc == FNX_OBST && fm[a] == FNX_FLUID));
      v = wall ? 0.f : v;
    }
    if (UBC) {
      const float m = need_u ? bm[a] : 1.f, c = need_u ? bc[a] : 0.f;
      const float t = v * m; v = t + c;
    }
    U[ou[a]] = v;
  }
  // (rho_done: the density has been through this setConstVals before and an identity cell would get its own bits back)
  if (do_rho) {
    const float m = need_r ? rm : 1.f, c = need_r ? rc : 0.f;
    const float t = r0 * m; rho[os] = t + c;
  }
}

// bit 0: every velocity component has mask == 1 and bc == +0 (x*1 + 0 is then what setConstVals computes); bit 1: density
__global__ __launch_bounds__(256) void bc_classify_kernel(size_t n, size_t dhw, int nc, const float* __restrict__ UBC,
                                                          const float* __restrict__ UBCInvMask,
                                                          const float* __restrict__ rhoBC,
                                                          const float* __restrict__ rhoBCInvMask,
                                                          unsigned char* __restrict__ cls) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const size_t b = q / dhw, o = q - b * dhw;
    unsigned c = 0;
    if (UBC && UBCInvMask) {
      bool id = true;
      for (int a = 0; a < nc; ++a) {
        const size_t ou = (b * nc + a) * dhw + o;
        id = id & (UBCInvMask[ou] == 1.f) & (__float_as_uint(UBC[ou]) == 0u);
      }
      c |= id ? 1u : 0u;
    }
    if (rhoBC && rhoBCInvMask) c |= ((rhoBCInvMask[q] == 1.f) & (__float_as_uint(rhoBC[q]) == 0u)) ? 2u : 0u;
    cls[q] = (unsigned char)c;
  }
}

// The periodic patches of the Jacobi branch (simulate.py:121-128 before the projection, :157-164 after it):
//   U_temp = U.clone(); U = setWallBcs(U); U[:,1,:,:,1] = U_temp[:,1,:,:,W-1] (periodic-x); U[:,0,:,1] = U_temp[:,0,:,H-1]
//   (periodic-y); then setConstVals.
// Both sources are border cells (column W-1, row H-1).  Thread t of a (b, plane) pair handles row j = t of the x patch
// (t < H) or column i = t - H of the y patch.
//
// Before the projection the value of a border cell ahead of setWallBcs is setConstVals of the advected velocity (buoyancy
// and gravity leave border cells alone), so the patch is re-derived from the stage's inputs: src = U_adv*m + c at the source
// cell, then the second setConstVals at the destination.
__global__ __launch_bounds__(256) void periodic_pre_kernel(GridDims g, int nc, const float* __restrict__ U_adv,
                                                           const float* __restrict__ UBC, const float* __restrict__ UBCInvMask,
                                                           float* __restrict__ U, int px, int py) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int bk = blockIdx.y, b = bk / g.KN, k = g.K0 + (bk - b * g.KN);
  if (t >= g.H + g.W) return;
  const bool xp = t < g.H;
  if (xp ? !px : !py) return;
  const int a = xp ? 1 : 0;
  const size_t base = ((size_t)b * nc + a) * g.DHW + (size_t)k * g.HW;
  const size_t src = base + (xp ? (size_t)t * g.W + (g.W - 1) : (size_t)(g.H - 1) * g.W + (t - g.H));
  const size_t dst = base + (xp ? (size_t)t * g.W + 1 : (size_t)g.W + (t - g.H));
  float v = U_adv[src];
  if (UBC) { float q = v * UBCInvMask[src]; v = q + UBC[src]; q = v * UBCInvMask[dst]; v = q + UBC[dst]; }
  U[dst] = v;
}

// After the projection velocityUpdate leaves border cells alone, so U_temp's source row / column is what the state held
// before the in-place post-projection pass: saved by mode 0, written (with the destination's setConstVals) by mode 1.
__global__ __launch_bounds__(256) void periodic_post_kernel(GridDims g, int nc, float* __restrict__ U, float* __restrict__ save,
                                                            const float* __restrict__ UBC, const float* __restrict__ UBCInvMask,
                                                            int px, int py, int mode) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int bk = blockIdx.y, b = bk / g.KN, k = g.K0 + (bk - b * g.KN);
  if (t >= g.H + g.W) return;
  const bool xp = t < g.H;
  if (xp ? !px : !py) return;
  const int a = xp ? 1 : 0;
  const size_t base = ((size_t)b * nc + a) * g.DHW + (size_t)k * g.HW;
  const size_t src = base + (xp ? (size_t)t * g.W + (g.W - 1) : (size_t)(g.H - 1) * g.W + (t - g.H));
  const size_t dst = base + (xp ? (size_t)t * g.W + 1 : (size_t)g.W + (t - g.H));
  float* sv = save + (size_t)bk * (g.H + g.W) + t;
  if (mode == 0) { *sv = U[src]; return; }
  float v = *sv;
  if (UBC) { const float q = v * UBCInvMask[dst]; v = q + UBC[dst]; }
  U[dst] = v;
}

inline dim3 cell_grid(const GridDims& g) { return dim3((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B * g.KN); }

}  // namespace

namespace fnx {

void launch_pre_projection(const GridDims& g, bool is3d, bool quirks, const float* U_adv, const float* rho_adv,
                           const float* flags, const float* UBC, const float* UBCInvMask, const float* rhoBC,
                           const float* rhoBCInvMask, float* U, float* rho, float* div, bool buoyancy, float sx,
                           float sy, float sz, float rho_star, bool wall_bcs, hipStream_t s, const unsigned char* cls,
                           const float* gravity, bool second_bcs, int div_k_end) {
  StepPtrs P{U_adv, rho_adv, flags, UBC, UBCInvMask, rhoBC, rhoBCInvMask, U, rho, div, cls,
             gravity ? 1 : 0, gravity ? gravity[0] : 0.f, gravity ? gravity[1] : 0.f, gravity ? gravity[2] : 0.f, second_bcs ? 1 : 0};
  const dim3 grid = cell_grid(g), block(BX, BY);
  if (!is3d) {
    if (wall_bcs) stage2d_div_kernel<true><<<grid, block, 0, s>>>(g, P, buoyancy, sx, sy, rho_star);
    else stage2d_div_kernel<false><<<grid, block, 0, s>>>(g, P, buoyancy, sx, sy, rho_star);
    return;
  }
  // 3D: `div` non-null: the fused form, divergence for the planes below div_k_end (the caller stages one plane more than it wants
  // divergences of where there is one); null: staging only
  const int kdiv = div ? div_k_end : 0;
#define STAGE3D(Q, W_, D_) stage3d_kernel<Q, W_, D_><<<grid, block, 0, s>>>(g, P, buoyancy, sx, sy, sz, rho_star, kdiv)
  if (div) {
    if (quirks) { if (wall_bcs) STAGE3D(true, true, true); else STAGE3D(true, false, true); }
    else        { if (wall_bcs) STAGE3D(false, true, true); else STAGE3D(false, false, true); }
  } else {
    if (quirks) { if (wall_bcs) STAGE3D(true, true, false); else STAGE3D(true, false, false); }
    else        { if (wall_bcs) STAGE3D(false, true, false); else STAGE3D(false, false, false); }
  }
#undef STAGE3D
}

void launch_post_projection(const GridDims& g, bool is3d, const float* p, float* U, float* rho, const float* flags,
                            const float* UBC, const float* UBCInvMask, const float* rhoBC, const float* rhoBCInvMask,
                            hipStream_t s, const unsigned char* cls, bool rho_bc_applied, const float* scale, float* p_scaled) {
  const dim3 grid = cell_grid(g), block(BX, BY);
  const int rd = (cls && rho_bc_applied) ? 1 : 0;
  // (the scaled form is its own instantiation: as a run-time branch it cost the Jacobi step's pass 124 -> 163 us at 512x512x64)
  if (scale && p_scaled) {
    if (is3d) post_projection_kernel<true, true><<<grid, block, 0, s>>>(g, p, U, rho, flags, UBC, UBCInvMask, rhoBC, rhoBCInvMask, cls, rd, scale, p_scaled);
    else post_projection_kernel<false, true><<<grid, block, 0, s>>>(g, p, U, rho, flags, UBC, UBCInvMask, rhoBC, rhoBCInvMask, cls, rd, scale, p_scaled);
  } else {
    if (is3d) post_projection_kernel<true, false><<<grid, block, 0, s>>>(g, p, U, rho, flags, UBC, UBCInvMask, rhoBC, rhoBCInvMask, cls, rd, nullptr, nullptr);
    else post_projection_kernel<false, false><<<grid, block, 0, s>>>(g, p, U, rho, flags, UBC, UBCInvMask, rhoBC, rhoBCInvMask, cls, rd, nullptr, nullptr);
  }
}

void launch_periodic_pre(const GridDims& g, bool is3d, const float* U_adv, const float* UBC, const float* UBCInvMask, float* U,
                         bool px, bool py, hipStream_t s) {
  const dim3 grid((g.H + g.W + 255) / 256, g.B * g.KN);
  periodic_pre_kernel<<<grid, 256, 0, s>>>(g, is3d ? 3 : 2, U_adv, UBC, UBCInvMask, U, px, py);
}

size_t periodic_save_bytes(const GridDims& g) { return (size_t)g.B * g.KN * (g.H + g.W) * sizeof(float); }

void launch_periodic_post(const GridDims& g, bool is3d, float* U, float* save, const float* UBC, const float* UBCInvMask,
                          bool px, bool py, int mode, hipStream_t s) {
  const dim3 grid((g.H + g.W + 255) / 256, g.B * g.KN);
  periodic_post_kernel<<<grid, 256, 0, s>>>(g, is3d ? 3 : 2, U, save, UBC, UBCInvMask, px, py, mode);
}

void launch_bc_classify(const GridDims& g, bool is3d, const float* UBC, const float* UBCInvMask, const float* rhoBC,
                        const float* rhoBCInvMask, unsigned char* cls, hipStream_t s) {
  const size_t n = (size_t)g.B * g.DHW;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  bc_classify_kernel<<<(unsigned)blocks, 256, 0, s>>>(n, (size_t)g.DHW, is3d ? 3 : 2, UBC, UBCInvMask, rhoBC, rhoBCInvMask, cls);
}

}  // namespace fnx
