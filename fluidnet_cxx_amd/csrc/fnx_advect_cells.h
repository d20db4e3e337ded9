// The per-cell functions of the MacCormack / semi-Lagrangian advection (advectScalar cpp/fluids_init.cpp:265-382, advectVel
// :656-807): what ONE cell computes in the forward pass and in the backward + correct + clamp pass.  Included INSIDE the
// anonymous namespace of the units that run them -- fnx_advect.hip (one thread per cell, the fix-up launches of the tile kernels)
// and fnx_small.hip (the single-launch step of small 2D grids) -- after fnx_device.h.  Bit-parity: see fnx_device.h.
struct CellId { int b, k, j, i; bool valid; };
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
        j0 = cell >> 16; i0 = cell & 0xffff;
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
