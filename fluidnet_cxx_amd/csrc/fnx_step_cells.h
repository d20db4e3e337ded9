// The per-cell functions of the fused stages of a time step (lib/simulate.py:96-171; see fnx_step.hip): what ONE cell computes in
// the 2D staging + divergence pass and in the post-projection pass.  Included INSIDE the anonymous namespace of the units that run
// them -- fnx_step.hip (one thread per cell) and fnx_small.hip (the single-launch step of small 2D grids) -- after fnx_device.h.
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
  unsigned long long* bar_reset;                     // optional: zeroed by the pass (the single-launch step's arrival counter,
                                                     // fnx_small.hip: the multi-launch step leaves it ready for the next step)
};

// addGravity's condition for one component of a non-border cell (source_terms.py:122-219; add_gravity_kernel)
__device__ __forceinline__ bool gravity_applies(float fc, float fm) {
  return (fc == FNX_FLUID || fc == FNX_EMPTY) && (fm == FNX_FLUID || (fm == FNX_EMPTY && fc == FNX_FLUID));
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

// (one call per cell of the grid, by every active lane of a wave: the identity-class test is a ballot)
template <bool WALL>
__device__ __forceinline__ void stage2d_div_cell(const GridDims& g, const StepPtrs& P, int i, int j, int b, int buoy_, float sx,
                                                 float sy, float rho_star) {
  const size_t o = (size_t)j * g.W + i, os = (size_t)b * g.DHW + o;
  if (P.bar_reset && os == 0) *P.bar_reset = 0ull;
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
__device__ __forceinline__ void post_projection_cell(const GridDims& g, int i, int j, int k, int b, const float* __restrict__ p,
                                                     float* __restrict__ U, float* __restrict__ rho,
                                                     const float* __restrict__ flags, const float* __restrict__ UBC,
                                                     const float* __restrict__ UBCInvMask, const float* __restrict__ rhoBC,
                                                     const float* __restrict__ rhoBCInvMask,
                                                     const unsigned char* __restrict__ cls, int rho_done,
                                                     const float* __restrict__ scale, float* __restrict__ p_scaled) {
  constexpr int NC = IS3D ? 3 : 2;
  const size_t o = (size_t)k * g.HW + j * g.W + i, os = (size_t)b * g.DHW + o;
  const float fc = flags[os], P = p[os];
  const unsigned cl = cls ? cls[os] : 0u;
  const bool border = is_border<IS3D>(g, i, j, k);
  const float sc = SCALE ? scale[b] : 1.f;
  if (SCALE) p_scaled[os] = P * sc;
#pragma unroll
  for (int a = 0; a < NC; ++a) {
    const size_t ou = ((size_t)b * NC + a) * g.DHW + o;
    const int off = a == 0 ? 1 : (a == 1 ? g.W : g.HW);
    const int idx = a == 0 ? i : (a == 1 ? j : k);
    const float fm = idx > 0 ? flags[os - off] : fc;
    float u = U[ou];
    if (SCALE) u = u / sc;                                 // model.py:129-168: the net saw U / s
    if (!border) {     // velocity_update.py:47-149
      const float Pm = p[os - off];
      const float m_ff = (fc == FNX_FLUID && fm == FNX_FLUID) ? 1.f : 0.f;
      if (!IS3D) {
        const float m_fe = (fc == FNX_FLUID && fm == FNX_EMPTY) ? 1.f : 0.f;
        const float m_ef = (fc == FNX_EMPTY && fm == FNX_FLUID) ? 1.f : 0.f;
        const float m_nf = (fc == FNX_EMPTY && fm == FNX_EMPTY) ? 1.f : 0.f;
        u = ((m_ff * (u - (P - Pm)) + m_fe * (u - P)) + m_ef * (u + Pm)) + m_nf * 0.f;
      } else {
        u = m_ff * (u - (P - Pm));
      }
    }
    if (SCALE) u = u * sc;                                 // model.py:221-223
    if (fc == FNX_FLUID || fc == FNX_OBST) {
      if (!(a == 2 && (k + g.zoff == 0 || k == 0))) {
        if (fm == FNX_OBST || (fc == FNX_OBST && fm == FNX_FLUID)) u = 0.f;
      }
    }
    if (UBC) {
      float m = 1.f, c = 0.f;
      if (!(cl & 1)) { m = UBCInvMask[ou]; c = UBC[ou]; }
      const float t = u * m; u = t + c;
    }
    U[ou] = u;
  }
  // (rho_done: the density has been through this setConstVals before and an identity cell would get its own bits back)
  if (rho && rhoBC && !(rho_done && (cl & 2))) {
    float m = 1.f, c = 0.f;
    if (!(cl & 2)) { m = rhoBCInvMask[os]; c = rhoBC[os]; }
    const float t = rho[os] * m; rho[os] = t + c;
  }
}

