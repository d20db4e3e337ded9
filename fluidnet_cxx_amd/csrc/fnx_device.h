// Device-side building blocks shared by the gfx950 kernels: grid indexing, MAC sampling,
// bilinear/trilinear sampling and the obstacle-aware line trace.
//
// Numerical contract: every expression is evaluated in fp32 in the operand order of the reference's
// ATen code (cited per function), the translation units are compiled with -ffp-contract=off, and the
// only fused operation is the explicit fmaf chain of the 3-vector norm (see line_trace).  That is what
// makes the kernels bit-identical to the reference on the same inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define FNX_FLUID 1.0f
#define FNX_OBST 2.0f
#define FNX_EMPTY 4.0f
#define FNX_HIT_MARGIN 1e-5f  // cpp/calc_line_trace.cpp:7
#define FNX_EPSILON 1e-12f    // cpp/calc_line_trace.cpp:8

struct GridDims {
  int B, D, H, W;
  int HW;        // H*W
  int DHW;       // D*H*W (cells per sample; < 2^31 per sample is required)
  // z-slab decomposition: this array holds planes [zoff, zoff + D) of a domain that is Dglob planes deep.
  // Only the border test looks at them; single-GPU grids have zoff = 0, Dglob = D.
  int zoff, Dglob;
  // compute window: plane-parallel kernels launch B*KN plane slices and produce planes [K0, K0+KN) only
  int K0, KN;
};

__host__ __device__ inline GridDims make_dims(int B, int D, int H, int W, int zoff = 0, int Dglob = 0) {
  GridDims d; d.B = B; d.D = D; d.H = H; d.W = W; d.HW = H * W; d.DHW = D * H * W;
  d.zoff = zoff; d.Dglob = Dglob > 0 ? Dglob : D;
  d.K0 = 0; d.KN = D;
  return d;
}

// per-sample field views: pointer to channel 0 of sample b; channel stride = DHW
struct Field {
  const float* p;
  __device__ __forceinline__ float at(const GridDims& g, int c, int k, int j, int i) const {
    return p[(size_t)c * g.DHW + (size_t)k * g.HW + j * g.W + i];
  }
};

template <bool IS3D>
__device__ __forceinline__ bool is_border(const GridDims& g, int i, int j, int k) {
  bool r = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2);
  // a slab's first/last local plane has no z neighbour in this array: never evaluated (it is a stale ghost plane)
  if (IS3D) r = r | (k + g.zoff < 1) | (k + g.zoff > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
  return r;
}

__device__ __forceinline__ float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  return v;
}

// ---------------------------------------------------------------------------------------------------
// MAC sampling, cpp/grid.cpp:274-446 (getCentered, getAtMACX/Y/Z).  Interior cells only.
// ---------------------------------------------------------------------------------------------------
template <bool IS3D>
__device__ __forceinline__ void get_centered(const GridDims& g, const Field& U, int i, int j, int k, float out[3]) {
  out[0] = 0.5f * (U.at(g, 0, k, j, i) + U.at(g, 0, k, j, i + 1));
  out[1] = 0.5f * (U.at(g, 1, k, j, i) + U.at(g, 1, k, j + 1, i));
  out[2] = IS3D ? 0.5f * (U.at(g, 2, k, j, i) + U.at(g, 2, k + 1, j, i)) : 0.f;
}

// QUIRKS (3D only): the reference's z entries stay 0 (grid.cpp:349-354,394-399,441-443).
template <bool IS3D, bool QUIRKS, int COMP>
__device__ __forceinline__ void get_at_mac(const GridDims& g, const Field& U, int i, int j, int k, float v[3]) {
  constexpr bool ZOK = IS3D && !QUIRKS;
  if (COMP == 0) {
    v[0] = U.at(g, 0, k, j, i);
    v[1] = 0.25f * (((U.at(g, 1, k, j, i) + U.at(g, 1, k, j, i - 1)) + U.at(g, 1, k, j + 1, i)) + U.at(g, 1, k, j + 1, i - 1));
    if (ZOK) v[2] = 0.25f * (((U.at(g, 2, k, j, i) + U.at(g, 2, k, j, i - 1)) + U.at(g, 2, k + 1, j, i)) + U.at(g, 2, k + 1, j, i - 1));
    else v[2] = 0.f;
  } else if (COMP == 1) {
    v[0] = 0.25f * (((U.at(g, 0, k, j, i) + U.at(g, 0, k, j - 1, i)) + U.at(g, 0, k, j, i + 1)) + U.at(g, 0, k, j - 1, i + 1));
    v[1] = U.at(g, 1, k, j, i);
    if (ZOK) v[2] = 0.25f * (((U.at(g, 2, k, j, i) + U.at(g, 2, k, j - 1, i)) + U.at(g, 2, k + 1, j, i)) + U.at(g, 2, k + 1, j - 1, i));
    else v[2] = 0.f;
  } else {
    v[0] = 0.25f * (((U.at(g, 0, k, j, i) + U.at(g, 0, k - 1, j, i)) + U.at(g, 0, k, j, i + 1)) + U.at(g, 0, k - 1, j, i + 1));
    v[1] = 0.25f * (((U.at(g, 1, k, j, i) + U.at(g, 1, k - 1, j, i)) + U.at(g, 1, k, j + 1, i)) + U.at(g, 1, k - 1, j + 1, i));
    v[2] = ZOK ? U.at(g, 2, k, j, i) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------
// Interpolation, cpp/grid.cpp:13-76 (interpol), :118-269 (interpolWithFluid), :448-511 (interpolComponent)
// ---------------------------------------------------------------------------------------------------
struct Lerp { int x0, y0, z0; float s0, s1, t0, t1, f0, f1; };

template <bool IS3D>
__device__ __forceinline__ Lerp lerp_setup(const GridDims& g, float px, float py, float pz) {
  Lerp L;
  px = px - 0.5f; py = py - 0.5f; pz = pz - 0.5f;
  const int qx = (int)px, qy = (int)py, qz = (int)pz;      // trunc toward zero
  const float s1 = px - (float)qx, t1 = py - (float)qy, f1 = pz - (float)qz;
  const float s0 = 1.f - s1, t0 = 1.f - t1, f0 = 1.f - f1;
  L.x0 = clampi(qx, 0, g.W - 2);
  L.y0 = clampi(qy, 0, g.H - 2);
  // positions are GLOBAL z; the plane index is clamped in the global domain (reference semantics) and then moved
  // into this array (and clamped again only so that a stale ghost cell can never read out of bounds)
  L.z0 = IS3D ? clampi(clampi(qz, 0, g.Dglob - 2) - g.zoff, 0, g.D - 2) : 0;   // 2D: plane 0
  L.s1 = clamp01(s1); L.t1 = clamp01(t1); L.f1 = clamp01(f1);
  L.s0 = clamp01(s0); L.t0 = clamp01(t0); L.f0 = clamp01(f0);
  return L;
}

// plain bi/trilinear sample of channel `c` of field f
template <bool IS3D>
__device__ __forceinline__ float interpol(const GridDims& g, const Field& f, int c, float px, float py, float pz) {
  const Lerp L = lerp_setup<IS3D>(g, px, py, pz);
  const float* q = f.p + (size_t)c * g.DHW + (size_t)L.z0 * g.HW + L.y0 * g.W + L.x0;
  const float Ia = q[0], Ib = q[g.W], Ic = q[1], Id = q[g.W + 1];
  const float lo = (Ia * L.t0 + Ib * L.t1) * L.s0 + (Ic * L.t0 + Id * L.t1) * L.s1;
  if (!IS3D) return lo;
  const float* r = q + g.HW;
  const float Ie = r[0], If = r[g.W], Ig = r[1], Ih = r[g.W + 1];
  const float hi = (Ie * L.t0 + If * L.t1) * L.s0 + (Ig * L.t0 + Ih * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}

__device__ __forceinline__ void lerp1d_fluid(float a, bool fa, float b, bool fb, float ta, float tb, float& v, bool& fl) {
  // cpp/grid.cpp:78-96
  if (!fa && !fb) { v = 0.f; fl = false; }
  else if (!fa)   { v = b; fl = true; }
  else if (!fb)   { v = a; fl = true; }
  else            { v = a * ta + b * tb; fl = true; }
}

template <bool IS3D, bool QUIRKS>
__device__ __forceinline__ float interpol_with_fluid(const GridDims& g, const Field& f, const Field& flags,
                                                     float px, float py, float pz) {
  const Lerp L = lerp_setup<IS3D>(g, px, py, pz);
  const size_t o = (size_t)L.z0 * g.HW + L.y0 * g.W + L.x0;
  const float* q = f.p + o;
  const float* m = flags.p + o;
  const float Ia = q[0], Ib = q[g.W], Ic = q[1], Id = q[g.W + 1];
  const bool fa = m[0] == FNX_FLUID, fb = m[g.W] == FNX_FLUID, fc = m[1] == FNX_FLUID, fd = m[g.W + 1] == FNX_FLUID;
  float vab, vcd, v; bool fab, fcd, fl;
  lerp1d_fluid(Ia, fa, Ib, fb, L.t0, L.t1, vab, fab);
  lerp1d_fluid(Ic, fc, Id, fd, L.t0, L.t1, vcd, fcd);
  lerp1d_fluid(vab, fab, vcd, fcd, L.s0, L.s1, v, fl);
  float plain = (Ia * L.t0 + Ib * L.t1) * L.s0 + (Ic * L.t0 + Id * L.t1) * L.s1;
  if (IS3D) {
    const float* r = q + g.HW;
    const float* n = m + g.HW;
    const float Ie = r[0], If = r[g.W], Ig = r[1], Ih = r[g.W + 1];
    const bool fe = n[0] == FNX_FLUID, ff = n[g.W] == FNX_FLUID;
    // Q15: grid.cpp:204-205 reads the g/h flags at x0 instead of x0+1
    const bool fg = QUIRKS ? fe : (n[1] == FNX_FLUID), fh = QUIRKS ? ff : (n[g.W + 1] == FNX_FLUID);
    float vef, vgh, vhi; bool fef, fgh, fhi;
    lerp1d_fluid(Ie, fe, If, ff, L.t0, L.t1, vef, fef);
    lerp1d_fluid(Ig, fg, Ih, fh, L.t0, L.t1, vgh, fgh);
    lerp1d_fluid(vef, fef, vgh, fgh, L.s0, L.s1, vhi, fhi);
    const float vlo = v; const bool flo = fl;
    lerp1d_fluid(vlo, flo, vhi, fhi, L.f0, L.f1, v, fl);
    const float hi = (Ie * L.t0 + If * L.t1) * L.s0 + (Ig * L.t0 + Ih * L.t1) * L.s1;
    plain = plain * L.f0 + hi * L.f1;
  }
  return fl ? v : plain;   // no fluid corner at all: plain interpol (grid.cpp:227-229, 265-267)
}

// ---------------------------------------------------------------------------------------------------
// Line trace, cpp/calc_line_trace.cpp:259-424 -- one ray per thread, bounded loops.
// The reference's AT_ASSERTM sanity checks of the march (:185-190, :346, :365, :391) are device asserts in a debug build
// (FNX_EXTRA_HIPCC_FLAGS=-DFNX_DEBUG_ASSERTS python -m fluidnet_cxx_amd.build --force); the release build has none.
// ---------------------------------------------------------------------------------------------------
#ifdef FNX_DEBUG_ASSERTS
#include <cassert>
#define FNX_DASSERT(cond, msg) assert((cond) && msg)
#else
#define FNX_DASSERT(cond, msg) ((void)0)
#endif
__device__ __forceinline__ bool out_of_domain(const GridDims& g, const float p[3]) {   // :16-27
  return (p[0] <= 0.f) | (p[0] >= (float)g.W) | (p[1] <= 0.f) | (p[1] >= (float)g.H) | (p[2] <= 0.f) | (p[2] >= (float)g.Dglob);
}
__device__ __forceinline__ bool blocked_cell(const GridDims& g, const Field& flags, const float p[3]) {  // :33-64
  if (out_of_domain(g, p)) return false;
  return flags.at(g, 0, clampi((int)p[2] - g.zoff, 0, g.D - 1), (int)p[1], (int)p[0]) != FNX_FLUID;
}

// HitBoundingBox, :73-149, batched-reference semantics (including the origin-inside-box case)
__device__ inline bool ray_box(const float origin[3], const float dir[3], const float ctr[3], float coord[3]) {
  float minB[3], maxB[3], cand[3], maxT[3];
  int quad[3];
  bool inside = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    minB[c] = (ctr[c] - 0.5f) - FNX_HIT_MARGIN;
    maxB[c] = (ctr[c] + 0.5f) + FNX_HIT_MARGIN;
    cand[c] = 0.f; quad[c] = 2;
    if (origin[c] < minB[c]) { quad[c] = 1; cand[c] = minB[c]; inside = false; }
    else if (origin[c] > maxB[c]) { quad[c] = 0; cand[c] = maxB[c]; inside = false; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!inside && quad[c] != 2 && dir[c] != 0.f) maxT[c] = (cand[c] - origin[c]) / dir[c];
    else if ((!inside && quad[c] == 2) || dir[c] == 0.f) maxT[c] = -1.f;
    else maxT[c] = 0.f;
  }
  int which = 0;
  if (maxT[1] > maxT[which]) which = 1;
  if (maxT[2] > maxT[which]) which = 2;
  const float fin = which == 0 ? maxT[0] : (which == 1 ? maxT[1] : maxT[2]);
  bool ret = !(fin < 0.f && !inside);
#pragma unroll
  for (int c = 0; c < 3; ++c) coord[c] = (which == c) ? cand[c] : origin[c] + fin * dir[c];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (which != c && (coord[c] < minB[c] - 1e-6f || coord[c] > maxB[c] + 1e-6f)) ret = false;
  return ret;
}

__device__ inline void line_trace(const GridDims& g, const Field& flags, const float pos[3], const float delta[3],
                                  float out[3]) {
  out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2];
  if (out_of_domain(g, pos)) return;
  if (blocked_cell(g, flags, pos)) return;
  // delta.norm(2,1): ATen's reduction is acc = fma(d,d,acc) (matches the reference build bit-for-bit)
  const float length = sqrtf(fmaf(delta[2], delta[2], fmaf(delta[1], delta[1], delta[0] * delta[0])));
  if (length <= FNX_EPSILON) return;
  const float dir[3] = { delta[0] / length, delta[1] / length, delta[2] / length };
  const float size[3] = { (float)g.W, (float)g.H, (float)g.Dglob };
  float cur = 0.f, next[3];
  // unit steps: a ray leaves the domain within W+H+D steps; the cap only guards against NaN inputs
  const int max_steps = g.W + g.H + g.Dglob + 8;
  for (int it = 0; it < max_steps; ++it) {
    if (cur >= length - FNX_HIT_MARGIN) return;
    const float step = fminf(length - cur, 1.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) next[c] = out[c] + dir[c] * step;
    if (out_of_domain(g, next)) {
      // Case 1: calcRayBorderIntersection from the ORIGINAL pos (:327, :175-257)
      FNX_DASSERT(!out_of_domain(g, pos), "Error: source location is already outside the domain!");             // :185-186
      float min_step = INFINITY, ipos[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (next[c] <= FNX_HIT_MARGIN) {
          const float d = next[c] - pos[c];
          if (fabsf(d) >= FNX_EPSILON) min_step = fminf(min_step, (FNX_HIT_MARGIN - pos[c]) / d);
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float lim = size[c] - FNX_HIT_MARGIN;
        if (next[c] >= lim) {
          const float d = next[c] - pos[c];
          if (fabsf(d) >= FNX_EPSILON) min_step = fminf(min_step, (lim - pos[c]) / d);
        }
      }
      if (min_step >= 0.f && min_step < INFINITY) {
#pragma unroll
        for (int c = 0; c < 3; ++c) ipos[c] = min_step * (next[c] - pos[c]) + pos[c];
      } else {
        // the reference aborts here (its clampToDomain is a no-op); per-cell intent: clamp for real
#pragma unroll
        for (int c = 0; c < 3; ++c) ipos[c] = fminf(fmaxf(next[c], FNX_HIT_MARGIN), size[c] - FNX_HIT_MARGIN);
      }
      FNX_DASSERT(!out_of_domain(g, ipos), "Error: case 1 exited bounds!");                                      // :346
      if (!blocked_cell(g, flags, ipos)) { out[0] = ipos[0]; out[1] = ipos[1]; out[2] = ipos[2]; return; }
      next[0] = ipos[0]; next[1] = ipos[1]; next[2] = ipos[2];
    }
    if (blocked_cell(g, flags, next)) {
      // Case 2 (:362-411): back off to the blocker's face, at most 4 times
      FNX_DASSERT(!blocked_cell(g, flags, out), "Error: Ray source is already in a blocked cell!");              // :365
      bool cont = true;
      for (int count = 0; count <= 4; ++count) {
        if (!blocked_cell(g, flags, next)) break;
        if (count == 4) { cont = false; break; }      // the reference raises
        float ctr[3], ipos[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) ctr[c] = (float)(int)next[c] + 0.5f;
        FNX_DASSERT(!out_of_domain(g, ctr), "Error: Center of blocker cell is out of the domain!");              // :391
        if (!ray_box(out, dir, ctr, ipos)) { cont = false; break; }
        next[0] = ipos[0]; next[1] = ipos[1]; next[2] = ipos[2];
      }
      if (cont) { out[0] = next[0]; out[1] = next[1]; out[2] = next[2]; }
      return;
    }
    out[0] = next[0]; out[1] = next[1]; out[2] = next[2];
    cur += step;
  }
}
