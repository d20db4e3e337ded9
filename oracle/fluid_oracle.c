/*
 * fluid_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, per-cell CPU restatement of the fluidnet_cxx fluid time-step operators.
 * It exists to check the HIP kernels (tests/, __graft_entry__.smoke()) and to be timed as
 * the "port" CPU baseline by bench.py.  Nothing under fluidnet_cxx_amd/ may include, link
 * or call it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (CFL<1) against golden
 * vectors captured from the reference's own ATen implementation (tools/make_golden.py ->
 * tests/golden/ *.npz, test_oracle_golden.py).
 *
 * Layout: every field is fp32, contiguous (B,C,D,H,W), x fastest; 2D == D=1 with a
 * 2-channel velocity.  flags is fp32 holding Manta cell types (1 fluid, 2 obstacle, 4 empty;
 * reference pytorch/lib/fluid/cpp/cell_type.h:7-18).
 *
 * All arithmetic is fp32 in the reference's expression order; build with -ffp-contract=off.
 *
 * `quirks` != 0 reproduces the reference's 3D defects so that 3D op-level
 * parity against the reference can be shown; `quirks` == 0 gives the intended 3D semantics
 * (solver_cpp/src/fluidnet_implementation).  2D results do not depend on `quirks`.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* zoff/Dglob: the arrays hold planes [zoff, zoff+D) of a Dglob-deep domain (z-slab tests); 0/0 = whole domain */
typedef struct { int B, D, H, W, is3D, zoff, Dglob; } OraGrid;
#define DG(g) ((g)->Dglob > 0 ? (g)->Dglob : (g)->D)

#define HIT_MARGIN 1e-5f   /* calc_line_trace.cpp:7 */
#define EPSILON    1e-12f  /* calc_line_trace.cpp:8 */
#define T_FLUID 1.0f
#define T_OBST  2.0f
#define T_EMPTY 4.0f
#define T_STICK 128.0f   /* cell_type.py:13, only in the separate flags_stick grid */

#define IDX(g, nc, b, c, k, j, i) \
  ((((((size_t)(b)) * (nc) + (c)) * (g)->D + (k)) * (g)->H + (j)) * (size_t)(g)->W + (i))

static inline int is_border(const OraGrid* g, int i, int j, int k, int bnd) {
  /* fluids_init.cpp:313-320 */
  int r = (i < bnd) || (i > g->W - 1 - bnd) || (j < bnd) || (j > g->H - 1 - bnd);
  if (g->is3D) r = r || (k + g->zoff < bnd) || (k + g->zoff > DG(g) - 1 - bnd) || (bnd == 1 && (k < 1 || k > g->D - 2));
  return r;
}

static inline long clampl(long v, long lo, long hi) {
  /* torch.clamp semantics: min applied first then max (min>max -> max). */
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  return v;
}
static inline float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

/* ------------------------------------------------------------------------------------------
 * MAC sampling  (reference grid.cpp:274-446; intended 3D: solver_cpp/.../grid.cpp:379-450)
 * All callers guarantee an interior cell.
 * ---------------------------------------------------------------------------------------- */
static void get_centered(const OraGrid* g, const float* U, int nc, int b, int i, int j, int k, float out[3]) {
  out[0] = 0.5f * (U[IDX(g, nc, b, 0, k, j, i)] + U[IDX(g, nc, b, 0, k, j, i + 1)]);
  out[1] = 0.5f * (U[IDX(g, nc, b, 1, k, j, i)] + U[IDX(g, nc, b, 1, k, j + 1, i)]);
  out[2] = g->is3D ? 0.5f * (U[IDX(g, nc, b, 2, k, j, i)] + U[IDX(g, nc, b, 2, k + 1, j, i)]) : 0.f;
}

static void get_at_mac(const OraGrid* g, const float* U, int nc, int b, int i, int j, int k,
                       int comp, int quirks, float v[3]) {
#define UU(c, kk, jj, ii) U[IDX(g, nc, b, c, kk, jj, ii)]
  const int zok = g->is3D && !quirks;   /* Q11: reference leaves z = 0 in 3D */
  if (comp == 0) {          /* grid.cpp:341-347 */
    v[0] = UU(0, k, j, i);
    v[1] = 0.25f * (((UU(1, k, j, i) + UU(1, k, j, i - 1)) + UU(1, k, j + 1, i)) + UU(1, k, j + 1, i - 1));
    v[2] = zok ? 0.25f * (((UU(2, k, j, i) + UU(2, k, j, i - 1)) + UU(2, k + 1, j, i)) + UU(2, k + 1, j, i - 1)) : 0.f;
  } else if (comp == 1) {   /* grid.cpp:385-392 */
    v[0] = 0.25f * (((UU(0, k, j, i) + UU(0, k, j - 1, i)) + UU(0, k, j, i + 1)) + UU(0, k, j - 1, i + 1));
    v[1] = UU(1, k, j, i);
    v[2] = zok ? 0.25f * (((UU(2, k, j, i) + UU(2, k, j - 1, i)) + UU(2, k + 1, j, i)) + UU(2, k + 1, j - 1, i)) : 0.f;
  } else {                  /* grid.cpp:430-443 (3D only) */
    v[0] = 0.25f * (((UU(0, k, j, i) + UU(0, k - 1, j, i)) + UU(0, k, j, i + 1)) + UU(0, k - 1, j, i + 1));
    v[1] = 0.25f * (((UU(1, k, j, i) + UU(1, k - 1, j, i)) + UU(1, k, j + 1, i)) + UU(1, k - 1, j + 1, i));
    v[2] = zok ? UU(2, k, j, i) : 0.f;
  }
#undef UU
}

/* ------------------------------------------------------------------------------------------
 * Interpolation (grid.cpp:13-76, 118-269, 448-511)
 * f points at channel 0 of batch b; `chan` selects the channel.
 * ---------------------------------------------------------------------------------------- */
typedef struct { long x0, y0, z0; float s0, s1, t0, t1, f0, f1; } Lerp;

static void lerp_setup(const OraGrid* g, const float pos[3], Lerp* L) {
  float px = pos[0] - 0.5f, py = pos[1] - 0.5f, pz = pos[2] - 0.5f;
  long qx = (long)px, qy = (long)py, qz = (long)pz;
  float s1 = px - (float)qx, t1 = py - (float)qy, f1 = pz - (float)qz;
  float s0 = 1.f - s1, t0 = 1.f - t1, f0 = 1.f - f1;
  L->x0 = clampl(qx, 0, g->W - 2);
  L->y0 = clampl(qy, 0, g->H - 2);
  if (g->is3D) L->z0 = clampl(clampl(qz, 0, DG(g) - 2) - g->zoff, 0, g->D - 2);   /* global plane -> this array */
  else L->z0 = 0;                    /* 2D: clamp(.,0,-1) = -1, which wraps to plane 0 */
  L->s1 = clamp01(s1); L->t1 = clamp01(t1); L->f1 = clamp01(f1);
  L->s0 = clamp01(s0); L->t0 = clamp01(t0); L->f0 = clamp01(f0);
}

static float interpol_chan(const OraGrid* g, const float* f, int nc, int b, int chan, const float pos[3]) {
  Lerp L; lerp_setup(g, pos, &L);
#define FF(kk, jj, ii) f[IDX(g, nc, b, chan, kk, jj, ii)]
  float Ia = FF(L.z0, L.y0, L.x0), Ib = FF(L.z0, L.y0 + 1, L.x0);
  float Ic = FF(L.z0, L.y0, L.x0 + 1), Id = FF(L.z0, L.y0 + 1, L.x0 + 1);
  float lo = (Ia * L.t0 + Ib * L.t1) * L.s0 + (Ic * L.t0 + Id * L.t1) * L.s1;
  if (!g->is3D) return lo;
  float Ie = FF(L.z0 + 1, L.y0, L.x0), If = FF(L.z0 + 1, L.y0 + 1, L.x0);
  float Ig = FF(L.z0 + 1, L.y0, L.x0 + 1), Ih = FF(L.z0 + 1, L.y0 + 1, L.x0 + 1);
  float hi = (Ie * L.t0 + If * L.t1) * L.s0 + (Ig * L.t0 + Ih * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
#undef FF
}

/* grid.cpp:78-96 */
static inline void lerp1d_fluid(float a, int fa, float b, int fb, float ta, float tb, float* v, int* fl) {
  if (!fa && !fb) { *v = 0.f; *fl = 0; }
  else if (!fa)   { *v = b;   *fl = 1; }
  else if (!fb)   { *v = a;   *fl = 1; }
  else            { *v = a * ta + b * tb; *fl = 1; }
}

static float interpol_with_fluid(const OraGrid* g, const float* f, const float* flags, int b,
                                 const float pos[3], int quirks) {
  Lerp L; lerp_setup(g, pos, &L);
#define FF(kk, jj, ii) f[IDX(g, 1, b, 0, kk, jj, ii)]
#define FL(kk, jj, ii) (flags[IDX(g, 1, b, 0, kk, jj, ii)] == T_FLUID)
  float vab, vcd, v; int fab, fcd, fl;
  lerp1d_fluid(FF(L.z0, L.y0, L.x0), FL(L.z0, L.y0, L.x0), FF(L.z0, L.y0 + 1, L.x0), FL(L.z0, L.y0 + 1, L.x0),
               L.t0, L.t1, &vab, &fab);
  lerp1d_fluid(FF(L.z0, L.y0, L.x0 + 1), FL(L.z0, L.y0, L.x0 + 1), FF(L.z0, L.y0 + 1, L.x0 + 1),
               FL(L.z0, L.y0 + 1, L.x0 + 1), L.t0, L.t1, &vcd, &fcd);
  lerp1d_fluid(vab, fab, vcd, fcd, L.s0, L.s1, &v, &fl);
  if (g->is3D) {
    float vef, vgh, vhi; int fef, fgh, fhi;
    long z1 = L.z0 + 1;
    long xg = quirks ? L.x0 : L.x0 + 1;    /* Q15: grid.cpp:204-205 read x0 for the g/h flags */
    lerp1d_fluid(FF(z1, L.y0, L.x0), FL(z1, L.y0, L.x0), FF(z1, L.y0 + 1, L.x0), FL(z1, L.y0 + 1, L.x0),
                 L.t0, L.t1, &vef, &fef);
    lerp1d_fluid(FF(z1, L.y0, L.x0 + 1), FL(z1, L.y0, xg), FF(z1, L.y0 + 1, L.x0 + 1), FL(z1, L.y0 + 1, xg),
                 L.t0, L.t1, &vgh, &fgh);
    lerp1d_fluid(vef, fef, vgh, fgh, L.s0, L.s1, &vhi, &fhi);
    float vlo = v; int flo = fl;
    lerp1d_fluid(vlo, flo, vhi, fhi, L.f0, L.f1, &v, &fl);
  }
#undef FF
#undef FL
  if (!fl) return interpol_chan(g, f, 1, b, 0, pos);
  return v;
}

/* ------------------------------------------------------------------------------------------
 * Line trace (calc_line_trace.cpp:259-424), one cell at a time, batched-reference semantics.
 * ---------------------------------------------------------------------------------------- */
static inline int out_of_domain(const OraGrid* g, const float p[3]) {   /* :16-27 */
  return (p[0] <= 0.f) || (p[0] >= (float)g->W) || (p[1] <= 0.f) || (p[1] >= (float)g->H) ||
         (p[2] <= 0.f) || (p[2] >= (float)DG(g));
}
static inline int blocked_cell(const OraGrid* g, const float* flags, int b, const float p[3]) {   /* :33-64 */
  if (out_of_domain(g, p)) return 0;
  long ix = (long)p[0], iy = (long)p[1], iz = clampl((long)p[2] - g->zoff, 0, g->D - 1);
  return flags[IDX(g, 1, b, 0, iz, iy, ix)] != T_FLUID;
}

/* HitBoundingBox, calc_line_trace.cpp:73-149 (incl. the inside-box behaviour, Q7). */
static int ray_box(const float origin[3], const float dir[3], const float ctr[3], float coord[3]) {
  float minB[3], maxB[3], cand[3], maxT[3];
  int quad[3]; int inside = 1;
  for (int c = 0; c < 3; ++c) {
    minB[c] = (ctr[c] - 0.5f) - HIT_MARGIN;
    maxB[c] = (ctr[c] + 0.5f) + HIT_MARGIN;
    cand[c] = 0.f; quad[c] = 2;
    if (origin[c] < minB[c]) { quad[c] = 1; cand[c] = minB[c]; inside = 0; }
    else if (origin[c] > maxB[c]) { quad[c] = 0; cand[c] = maxB[c]; inside = 0; }
  }
  for (int c = 0; c < 3; ++c) {
    if (!inside && quad[c] != 2 && dir[c] != 0.f) maxT[c] = (cand[c] - origin[c]) / dir[c];
    else if ((!inside && quad[c] == 2) || dir[c] == 0.f) maxT[c] = -1.f;
    else maxT[c] = 0.f;
  }
  int which = 0;
  if (maxT[1] > maxT[which]) which = 1;
  if (maxT[2] > maxT[which]) which = 2;
  float fin = maxT[which];
  int ret = 1;
  if (fin < 0.f && !inside) ret = 0;
  for (int c = 0; c < 3; ++c) coord[c] = (which == c) ? cand[c] : origin[c] + fin * dir[c];
  for (int c = 0; c < 3; ++c)
    if (which != c && (coord[c] < minB[c] - 1e-6f || coord[c] > maxB[c] + 1e-6f)) ret = 0;
  return ret;
}

static void line_trace(const OraGrid* g, const float* flags, int b, const float pos[3], const float delta[3],
                       float out[3]) {
  out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2];
  if (out_of_domain(g, pos)) return;
  if (blocked_cell(g, flags, b, pos)) return;
  /* delta.norm(2,1): ATen's reduction accumulates acc = fma(d,d,acc) on FMA-capable hosts (verified
   * bit-for-bit against the reference build here), so the squares are fused, not rounded. */
  const float length = sqrtf(fmaf(delta[2], delta[2], fmaf(delta[1], delta[1], delta[0] * delta[0])));
  if (length <= EPSILON) return;
  const float dir[3] = { delta[0] / length, delta[1] / length, delta[2] / length };
  const float size[3] = { (float)g->W, (float)g->H, (float)DG(g) };
  float cur = 0.f, next[3];
  const int max_steps = g->W + g->H + DG(g) + 8;   /* NaN guard only: a unit-step ray exits the domain sooner */
  for (int it = 0; it < max_steps; ++it) {
    if (cur >= length - HIT_MARGIN) return;
    const float step = fminf(length - cur, 1.f);
    for (int c = 0; c < 3; ++c) next[c] = out[c] + dir[c] * step;
    if (out_of_domain(g, next)) {
      /* Case 1. calcRayBorderIntersection(pos, next) -- from the ORIGINAL pos (Q9). */
      float min_step = INFINITY, ipos[3];
      for (int c = 0; c < 3; ++c) {
        if (next[c] <= HIT_MARGIN) {
          float d = next[c] - pos[c];
          if (fabsf(d) >= EPSILON) min_step = fminf(min_step, (HIT_MARGIN - pos[c]) / d);
        }
      }
      for (int c = 0; c < 3; ++c) {
        const float lim = size[c] - HIT_MARGIN;
        if (next[c] >= lim) {
          float d = next[c] - pos[c];
          if (fabsf(d) >= EPSILON) min_step = fminf(min_step, (lim - pos[c]) / d);
        }
      }
      if (min_step >= 0.f && min_step < INFINITY) {
        for (int c = 0; c < 3; ++c) ipos[c] = min_step * (next[c] - pos[c]) + pos[c];
      } else {
        /* reference aborts here (Q8); per-cell intent: clamp for real. */
        for (int c = 0; c < 3; ++c) ipos[c] = fminf(fmaxf(next[c], HIT_MARGIN), size[c] - HIT_MARGIN);
      }
      if (!blocked_cell(g, flags, b, ipos)) { out[0] = ipos[0]; out[1] = ipos[1]; out[2] = ipos[2]; return; }
      next[0] = ipos[0]; next[1] = ipos[1]; next[2] = ipos[2];
    }
    if (blocked_cell(g, flags, b, next)) {
      /* Case 2. */
      int cont = 1;
      for (int count = 0; count <= 4; ++count) {
        if (!blocked_cell(g, flags, b, next)) break;
        if (count == 4) { cont = 0; break; }    /* reference raises */
        float ctr[3], ipos[3];
        for (int c = 0; c < 3; ++c) ctr[c] = (float)(long)next[c] + 0.5f;
        if (!ray_box(out, dir, ctr, ipos)) { cont = 0; break; }
        next[0] = ipos[0]; next[1] = ipos[1]; next[2] = ipos[2];
      }
      if (cont) { out[0] = next[0]; out[1] = next[1]; out[2] = next[2]; }
      return;
    }
    out[0] = next[0]; out[1] = next[1]; out[2] = next[2];
    cur += step;
  }
}

/* ------------------------------------------------------------------------------------------
 * advectScalar (fluids_init.cpp:265-382)
 * ---------------------------------------------------------------------------------------- */
/* One semi-Lagrangian pass (SemiLagrangeEulerFluidNetSavePos, :69-133). pos_out may be NULL. */
static void sl_scalar_pass(const OraGrid* g, float dt, const float* src, const float* U, const float* flags,
                           int sample_outside, int quirks, float* dst, float* pos_out) {
  const int nc = g->is3D ? 3 : 2;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          const float ctr[3] = { (float)i + 0.5f, (float)j + 0.5f, (float)(k + g->zoff) + 0.5f };
          float val = 0.f, p[3] = { ctr[0], ctr[1], ctr[2] };
          if (!is_border(g, i, j, k, 1)) {
            if (flags[c] != T_FLUID) {
              val = src[c];
            } else {
              float cen[3], disp[3], back[3];
              get_centered(g, U, nc, b, i, j, k, cen);
              for (int a = 0; a < 3; ++a) disp[a] = (-dt) * cen[a];
              line_trace(g, flags, b, ctr, disp, back);
              val = sample_outside ? interpol_chan(g, src, 1, b, 0, back)
                                   : interpol_with_fluid(g, src, flags, b, back, quirks);
              p[0] = back[0]; p[1] = back[1]; p[2] = back[2];
            }
          }
          dst[c] = val;
          if (pos_out) {
            pos_out[IDX(g, 3, b, 0, k, j, i)] = p[0];
            pos_out[IDX(g, 3, b, 1, k, j, i)] = p[1];
            pos_out[IDX(g, 3, b, 2, k, j, i)] = p[2];
          }
        }
}

/* method: 0 = eulerFluidNet, 1 = maccormackFluidNet (advect_type.cpp:5-16) */
int ora_advect_scalar(const OraGrid* g, float dt, const float* src, const float* U, const float* flags,
                      float* dst, int method, int bnd, int sample_outside, float strength, int quirks) {
  if (bnd != 1) return 1;   /* Q4: only bnd = 1 is meaningful in the reference */
  const size_t n = (size_t)g->B * g->D * g->H * g->W;
  if (method == 0) { sl_scalar_pass(g, dt, src, U, flags, sample_outside, quirks, dst, NULL); return 0; }
  float* fwd = (float*)malloc(n * sizeof(float));
  float* bwd = (float*)malloc(n * sizeof(float));
  float* fpos = (float*)malloc(3 * n * sizeof(float));
  sl_scalar_pass(g, dt, src, U, flags, sample_outside, quirks, fwd, fpos);
  sl_scalar_pass(g, -dt, fwd, U, flags, sample_outside, quirks, bwd, NULL);
  const float half_s = strength * 0.5f;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          float d = fwd[c];
          if (flags[c] == T_FLUID) d = fwd[c] + half_s * (src[c] - bwd[c]);   /* :135-148, all cells (Q3) */
          if (!is_border(g, i, j, k, 1)) {
            /* getClampBounds :154-222 */
            long i0 = clampl((long)fpos[IDX(g, 3, b, 0, k, j, i)], 0, g->W - 1);
            long j0 = clampl((long)fpos[IDX(g, 3, b, 1, k, j, i)], 0, g->H - 1);
            long k0 = (g->is3D && !quirks) ? clampl((long)fpos[IDX(g, 3, b, 2, k, j, i)], 0, DG(g) - 1) - g->zoff : -g->zoff;  /* Q10 */
            float mn = INFINITY, mx = -INFINITY; int cnt = 0;
            for (int dk = -1; dk <= 1; ++dk)
              for (int dj = -1; dj <= 1; ++dj)
                for (int di = -1; di <= 1; ++di) {
                  long kk = k0 + dk, jj = j0 + dj, ii = i0 + di;
                  if (kk + g->zoff < 0 || kk + g->zoff >= DG(g)) continue;
                  if (kk < 0 || kk >= g->D || jj < 0 || jj >= g->H || ii < 0 || ii >= g->W) continue;
                  const size_t q = IDX(g, 1, b, 0, kk, jj, ii);
                  if (flags[q] == T_FLUID || sample_outside) {
                    mn = fminf(mn, src[q]); mx = fmaxf(mx, src[q]); ++cnt;
                  }
                }
            d = cnt ? fmaxf(mn, fminf(mx, d)) : fwd[c];    /* :262 */
          }
          dst[c] = d;
        }
  free(fwd); free(bwd); free(fpos);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * advectVel (fluids_init.cpp:656-807)
 * ---------------------------------------------------------------------------------------- */
static void sl_mac_pass(const OraGrid* g, float dt, const float* src, const float* U, const float* flags,
                        int quirks, float* dst) {
  const int nc = g->is3D ? 3 : 2;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          float r[3] = { 0.f, 0.f, 0.f };
          if (!is_border(g, i, j, k, 1)) {
            if (flags[IDX(g, 1, b, 0, k, j, i)] != T_FLUID) {
              r[0] = src[IDX(g, nc, b, 1, k, j, i)];                       /* Q1, :413-416 */
              if (g->is3D) r[2] = src[IDX(g, nc, b, 2, k, j, i)];
            } else {
              const float ctr[3] = { (float)i + 0.5f, (float)j + 0.5f, (float)(k + g->zoff) + 0.5f };
              for (int c = 0; c < nc; ++c) {
                if (c == 2 && quirks) { r[2] = 0.f; break; }               /* Q12, :441-447 */
                float v[3], p[3];
                get_at_mac(g, U, nc, b, i, j, k, c, quirks, v);
                for (int a = 0; a < 3; ++a) p[a] = ctr[a] + v[a] * (-dt);
                r[c] = interpol_chan(g, src, nc, b, c, p);
              }
            }
          }
          for (int c = 0; c < nc; ++c) dst[IDX(g, nc, b, c, k, j, i)] = r[c];
        }
}

int ora_advect_vel(const OraGrid* g, float dt, const float* orig, const float* U, const float* flags,
                   float* dst, int method, int bnd, float strength, int quirks) {
  if (bnd != 1) return 1;
  const int nc = g->is3D ? 3 : 2;
  const size_t n = (size_t)g->B * nc * g->D * g->H * g->W;
  if (method == 0) { sl_mac_pass(g, dt, orig, U, flags, quirks, dst); return 0; }
  float* fwd = (float*)malloc(n * sizeof(float));
  float* bwd = (float*)malloc(n * sizeof(float));
  sl_mac_pass(g, dt, orig, U, flags, quirks, fwd);
  sl_mac_pass(g, -dt, fwd, U, flags, quirks, bwd);
  const float half_s = strength * 0.5f;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          if (is_border(g, i, j, k, 1)) {
            for (int c = 0; c < nc; ++c) dst[IDX(g, nc, b, c, k, j, i)] = 0.f;
            continue;
          }
          const int fl = flags[IDX(g, 1, b, 0, k, j, i)] == T_FLUID;
          const int idx[3] = { i, j, k + g->zoff };
          for (int c = 0; c < nc; ++c) {
            const size_t q = IDX(g, nc, b, c, k, j, i);
            /* MacCormackCorrectMAC :453-498 */
            int skip = !fl;
            if (idx[c] > 0 && !(c == 2 && k == 0)) {
              const size_t qm = IDX(g, 1, b, 0, k - (c == 2), j - (c == 1), i - (c == 0));
              if (flags[qm] != T_FLUID) skip = 1;
            }
            float d = skip ? fwd[q] : fwd[q] + half_s * (orig[q] - bwd[q]);
            /* doClampComponentMAC :500-614 */
            float v[3];
            get_at_mac(g, U, nc, b, i, j, k, c, quirks, v);
            for (int a = 0; a < 3; ++a) v[a] = v[a] * dt;
            const float pos[3] = { (float)i, (float)j, (float)(k + g->zoff) };
            float mn = INFINITY, mx = -INFINITY;
            for (int l = 0; l < 2; ++l) {
              int q0[3];
              for (int a = 0; a < 3; ++a) q0[a] = (int)(l == 0 ? pos[a] - v[a] : pos[a] + v[a]);
              long i0 = clampl(q0[0], 0, g->W - 2), j0 = clampl(q0[1], 0, g->H - 2);
              long k0 = g->is3D ? clampl(clampl(q0[2], 0, DG(g) - 2) - g->zoff, 0, g->D - 2) : 0;
              long k1 = g->is3D ? k0 + 1 : k0;
              for (long kk = k0; kk <= k1; ++kk)
                for (long jj = j0; jj <= j0 + 1; ++jj)
                  for (long ii = i0; ii <= i0 + 1; ++ii) {
                    float o = orig[IDX(g, nc, b, c, kk, jj, ii)];
                    mn = fminf(mn, o); mx = fmaxf(mx, o);
                  }
            }
            dst[q] = fmaxf(fminf(d, mx), mn);
          }
        }
  free(fwd); free(bwd);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * velocityDivergence (velocity_divergence.py:46-74)
 * ---------------------------------------------------------------------------------------- */
int ora_velocity_divergence(const OraGrid* g, const float* U, const float* flags, float* div) {
  const int nc = g->is3D ? 3 : 2;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          float d = 0.f;
          if (!is_border(g, i, j, k, 1)) {
            d = ((U[IDX(g, nc, b, 0, k, j, i)] - U[IDX(g, nc, b, 0, k, j, i + 1)]) + U[IDX(g, nc, b, 1, k, j, i)])
                - U[IDX(g, nc, b, 1, k, j + 1, i)];
            if (g->is3D) d = d + (U[IDX(g, nc, b, 2, k, j, i)] - U[IDX(g, nc, b, 2, k + 1, j, i)]);
          }
          if (flags[c] == T_OBST) d = 0.f;
          div[c] = d;
        }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * solveLinearSystemJacobi (fluids_init.cpp:809-1004)
 * p must hold B*D*H*W floats; residual receives max_b ||p - p_prev||_2 of the last sweep.
 * ---------------------------------------------------------------------------------------- */
int ora_jacobi(const OraGrid* g, const float* flags, const float* div, float* p, float* residual,
               float p_tol, int max_iter, int quirks, int* iters_done) {
  if (max_iter < 1) return 2;
  const size_t nb = (size_t)g->D * g->H * g->W, n = nb * g->B;
  float* bufA = p;
  float* bufB = (float*)calloc(n, sizeof(float));
  memset(bufA, 0, n * sizeof(float));
  float* cur = bufA; float* prev = bufB;
  const float denom = g->is3D ? 6.f : 4.f;
  double* rsum = (double*)calloc(g->B, sizeof(double));
  int iter = 0; float res = 0.f;
  for (;;) {
    for (int b = 0; b < g->B; ++b) rsum[b] = 0.0;
    for (int b = 0; b < g->B; ++b) {
      double acc = 0.0;
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : acc)
      for (int k = 0; k < g->D; ++k)
        for (int j = 0; j < g->H; ++j)
          for (int i = 0; i < g->W; ++i) {
            const size_t c = IDX(g, 1, b, 0, k, j, i);
            float v = 0.f;
            if (!is_border(g, i, j, k, 1) && flags[c] != T_OBST) {
              const float pc = prev[c];
#define NB(q, sub) ((sub) && flags[q] == T_OBST ? pc : prev[q])
              const size_t xl = c - 1, xr = c + 1, yl = c - g->W, yr = c + g->W;
              float s = NB(xl, 1) + NB(xr, 1);
              s = s + NB(yl, 1);
              s = s + NB(yr, 1);
              if (g->is3D) {
                const size_t zl = c - (size_t)g->H * g->W, zr = c + (size_t)g->H * g->W;
                s = s + NB(zl, !quirks);      /* Q13: no Neumann substitution in z */
                s = s + NB(zr, !quirks);
              } else {
                s = s + 0.f; s = s + 0.f;
              }
#undef NB
              v = (s + div[c]) / denom;
            }
            cur[c] = v;
            const double dlt = (double)v - (double)prev[c];
            acc += dlt * dlt;
          }
      rsum[b] = acc;
    }
    double mx = 0.0;
    for (int b = 0; b < g->B; ++b) { double r = sqrt(rsum[b]); if (r > mx) mx = r; }
    res = (float)mx;
    if (res < p_tol) break;
    ++iter;
    if (iter >= max_iter) break;
    float* t = cur; cur = prev; prev = t;
  }
  if (cur != p) memcpy(p, cur, n * sizeof(float));
  if (residual) *residual = res;
  if (iters_done) *iters_done = iter;
  free(bufB); free(rsum);
  return 0;
}

/* `nsweeps` more sweeps on an existing pressure field, in place (same per-cell update as ora_jacobi). */
int ora_jacobi_sweeps(const OraGrid* g, const float* flags, const float* div, float* p, int nsweeps, int quirks) {
  const size_t n = (size_t)g->B * g->D * g->H * g->W;
  float* prev = (float*)malloc(n * sizeof(float));
  const float denom = g->is3D ? 6.f : 4.f;
  for (int it = 0; it < nsweeps; ++it) {
    memcpy(prev, p, n * sizeof(float));
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < g->B; ++b)
      for (int k = 0; k < g->D; ++k)
        for (int j = 0; j < g->H; ++j)
          for (int i = 0; i < g->W; ++i) {
            const size_t c = IDX(g, 1, b, 0, k, j, i);
            float v = 0.f;
            if (!is_border(g, i, j, k, 1) && flags[c] != T_OBST) {
              const float pc = prev[c];
#define NB(q, sub) ((sub) && flags[q] == T_OBST ? pc : prev[q])
              float s = NB(c - 1, 1) + NB(c + 1, 1);
              s = s + NB(c - g->W, 1);
              s = s + NB(c + g->W, 1);
              if (g->is3D) {
                s = s + NB(c - (size_t)g->H * g->W, !quirks);
                s = s + NB(c + (size_t)g->H * g->W, !quirks);
              } else {
                s = s + 0.f; s = s + 0.f;
              }
#undef NB
              v = (s + div[c]) / denom;
            }
            p[c] = v;
          }
  }
  free(prev);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * velocityUpdate (velocity_update.py:47-149; 3D intent solver_cpp/src/projection/update_vel.cpp:58-117)
 * ---------------------------------------------------------------------------------------- */
int ora_velocity_update(const OraGrid* g, const float* p, float* U, const float* flags) {
  const int nc = g->is3D ? 3 : 2;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          if (is_border(g, i, j, k, 1)) continue;
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          const float fc = flags[c], P = p[c];
          for (int a = 0; a < nc; ++a) {
            const size_t qm = IDX(g, 1, b, 0, k - (a == 2), j - (a == 1), i - (a == 0));
            const float fm = flags[qm], Pm = p[qm];
            const size_t q = IDX(g, nc, b, a, k, j, i);
            const float u = U[q];
            if (!g->is3D) {
              const float m_ff = (fc == T_FLUID && fm == T_FLUID) ? 1.f : 0.f;
              const float m_fe = (fc == T_FLUID && fm == T_EMPTY) ? 1.f : 0.f;
              const float m_ef = (fc == T_EMPTY && fm == T_FLUID) ? 1.f : 0.f;
              const float m_nf = (fc == T_EMPTY && fm == T_EMPTY) ? 1.f : 0.f;
              U[q] = ((m_ff * (u - (P - Pm)) + m_fe * (u - P)) + m_ef * (u + Pm)) + m_nf * 0.f;
            } else {
              const float m_ff = (fc == T_FLUID && fm == T_FLUID) ? 1.f : 0.f;
              U[q] = m_ff * (u - (P - Pm));
            }
          }
        }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * addBuoyancy (source_terms.py:47-116)
 * ---------------------------------------------------------------------------------------- */
int ora_add_buoyancy(const OraGrid* g, float* U, const float* flags, const float* rho, const float gravity[3],
                     float rho_star, float dt, int quirks) {
  const int nc = g->is3D ? 3 : 2;
  const float st[3] = { gravity[0] * dt, gravity[1] * dt, gravity[2] * dt };
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          if (is_border(g, i, j, k, 1)) continue;
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          if (flags[c] != T_FLUID) continue;
          const size_t xl = c - 1, yl = c - g->W;
          if (flags[xl] == T_FLUID) {
            const size_t q = IDX(g, nc, b, 0, k, j, i);
            U[q] = U[q] + st[0] * ((0.5f * (rho[c] + rho[xl])) - rho_star);
          }
          if (flags[yl] == T_FLUID) {
            const size_t q = IDX(g, nc, b, 1, k, j, i);
            U[q] = U[q] + st[1] * ((0.5f * (rho[c] + rho[yl])) - rho_star);
          }
          if (g->is3D) {
            const size_t zl = c - (size_t)g->H * g->W;
            const size_t q = IDX(g, nc, b, 2, k, j, i);
            if (!quirks) {
              if (flags[zl] == T_FLUID) U[q] = U[q] + st[2] * ((0.5f * (rho[c] + rho[zl])) - rho_star);
            } else {
              /* Q14, source_terms.py:110-114: tests j<=0 (never true inside), no rho_star, zero for k<=1 */
              if (flags[zl] == T_FLUID) U[q] = U[q] + st[2] * (0.5f * (rho[c] + (k + g->zoff <= 1 ? 0.f : rho[zl])));
            }
          }
        }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * addGravity (source_terms.py:122-219): U_c += gravity_c*dt on interior fluid/empty cells whose -1 neighbour is
 * fluid, or is empty while the cell itself is fluid.
 * ---------------------------------------------------------------------------------------- */
int ora_add_gravity(const OraGrid* g, float* U, const float* flags, const float gravity[3], float dt) {
  const int nc = g->is3D ? 3 : 2;
  const float f[3] = { gravity[0] * dt, gravity[1] * dt, gravity[2] * dt };
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          if (is_border(g, i, j, k, 1)) continue;
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          const float fc = flags[c];
          if (fc != T_FLUID && fc != T_EMPTY) continue;
          for (int a = 0; a < nc; ++a) {
            const float fm = flags[IDX(g, 1, b, 0, k - (a == 2), j - (a == 1), i - (a == 0))];
            if (fm == T_FLUID || (fm == T_EMPTY && fc == T_FLUID)) {
              const size_t q = IDX(g, nc, b, a, k, j, i);
              U[q] = U[q] + f[a];
            }
          }
        }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * addViscosity (viscosity.py:7-70, 2D only): U = m * (U + (dt*nu) * (U[i+1] + U[j+1] + U[i-1] + U[i-1,j-1] - 4U))
 * on interior cells, m_c = fluid(cell) && fluid(cell - e_c); the fourth neighbour is (i-1,j-1) as in the reference
 * (:68).  Reads the old field throughout (the reference evaluates the right-hand side before assigning).
 * ---------------------------------------------------------------------------------------- */
int ora_add_viscosity(const OraGrid* g, float dt, float* U, const float* flags, float viscosity) {
  if (g->is3D) return 1;
  const size_t n = (size_t)g->B * 2 * g->H * g->W;
  float* old = (float*)malloc(n * sizeof(float));
  memcpy(old, U, n * sizeof(float));
  const float coef = (float)((double)dt * (double)viscosity);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int j = 1; j < g->H - 1; ++j)
      for (int i = 1; i < g->W - 1; ++i) {
        const size_t c = IDX(g, 1, b, 0, 0, j, i);
        for (int a = 0; a < 2; ++a) {
          const float m = (flags[c] == T_FLUID && flags[c - (a == 0 ? 1 : g->W)] == T_FLUID) ? 1.f : 0.f;
          const size_t q = IDX(g, 2, b, a, 0, j, i);
          float s = old[q + 1] + old[q + g->W];
          s = s + old[q - 1];
          s = s + old[q - 1 - g->W];
          s = s - (4.f * old[q]);
          U[q] = m * (old[q] + coef * s);
        }
      }
  free(old);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * setWallBcs (set_wall_bcs.py:45-84)
 * ---------------------------------------------------------------------------------------- */
int ora_set_wall_bcs(const OraGrid* g, float* U, const float* flags) {
  const int nc = g->is3D ? 3 : 2;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const size_t c = IDX(g, 1, b, 0, k, j, i);
          const float fc = flags[c];
          if (fc != T_FLUID && fc != T_OBST) continue;
          const int il = i > 0 ? i - 1 : 0, jl = j > 0 ? j - 1 : 0;
          const float fx = flags[IDX(g, 1, b, 0, k, j, il)];
          const float fy = flags[IDX(g, 1, b, 0, k, jl, i)];
          if (fx == T_OBST || (fc == T_OBST && fx == T_FLUID)) U[IDX(g, nc, b, 0, k, j, i)] = 0.f;
          if (fy == T_OBST || (fc == T_OBST && fy == T_FLUID)) U[IDX(g, nc, b, 1, k, j, i)] = 0.f;
          if (g->is3D && k > 0 && k + g->zoff > 0) {
            const float fz = flags[IDX(g, 1, b, 0, k - 1, j, i)];
            if (fz == T_OBST || (fc == T_OBST && fz == T_FLUID)) U[IDX(g, nc, b, 2, k, j, i)] = 0.f;
          }
        }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * setWallBcsStick (set_wall_bcs_stick.py:5-157), 2D.  The reference file raises NameError as shipped (bare
 * TypeObstacle / TypeFluid / TypeStick from :62 on); with those three names bound it runs as written in 2D, and that
 * body is what is restated here, quirks included:
 *   phase 1 (:57-77)   U = 0 in obstacle cells; U_c = 0 where the -1 neighbour along c is an obstacle (cells of type
 *                      fluid / obstacle / stick only);
 *   phase 2 (:97-136)  in stick cells the tangential component mirrors the fluid neighbour across the wall:
 *                      v = -v(i-1) if the left neighbour is fluid, -v(i+1) if the right one is (the later rule wins),
 *                      0.5*((-v(i-1)) - v(i+1)) if both; u likewise with the neighbours below / above -- except that
 *                      the reference's "both" test checks the lower neighbour twice (:131), so the mean is taken
 *                      whenever the lower neighbour is fluid;
 *   phase 3 (:138-156) corners: u = 0 where 2*cur + 2*left + below + above == 3, v = 0 where
 *                      2*cur + left + 2*below + right == 3 (stick indicators; the doubled terms are the reference's).
 * All neighbour velocities are the values after phase 1 (the reference gathers them before it scatters).  Neighbour
 * indices clamp to the cell itself at the domain edge (:60,:72,:94-95); the gathered velocity is 0 there (:103-104).
 * U_in and U_out are distinct buffers.
 * ---------------------------------------------------------------------------------------- */
static inline float stick_phase1(const OraGrid* g, const float* U, const float* flags, const float* stick, int b, int c,
                                 int j, int i) {
  const float fc = flags[IDX(g, 1, b, 0, 0, j, i)];
  float u = U[IDX(g, 2, b, c, 0, j, i)];
  if (fc == T_OBST) return 0.f;
  const int cont = fc == T_FLUID || fc == T_OBST || stick[IDX(g, 1, b, 0, 0, j, i)] == T_STICK;
  if (c == 0 && i > 0 && cont && flags[IDX(g, 1, b, 0, 0, j, i - 1)] == T_OBST) u = 0.f;
  if (c == 1 && j > 0 && cont && flags[IDX(g, 1, b, 0, 0, j - 1, i)] == T_OBST) u = 0.f;
  return u;
}

int ora_set_wall_bcs_stick(const OraGrid* g, const float* U_in, float* U_out, const float* flags, const float* stick) {
  if (g->is3D || g->D != 1) return -1;
  const int H = g->H, W = g->W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < g->B; ++b)
    for (int j = 0; j < H; ++j)
      for (int i = 0; i < W; ++i) {
        const size_t c = IDX(g, 1, b, 0, 0, j, i);
        const float fc = flags[c];
        const int S = stick[c] == T_STICK;
        const int cont = fc == T_FLUID || fc == T_OBST || S;
        float u = stick_phase1(g, U_in, flags, stick, b, 0, j, i);
        float v = stick_phase1(g, U_in, flags, stick, b, 1, j, i);
        const int il = i > 0 ? i - 1 : 0, ir = i < W - 1 ? i + 1 : W - 1;
        const int jl = j > 0 ? j - 1 : 0, jr = j < H - 1 ? j + 1 : H - 1;
        if (S && cont) {
          const int f_l = flags[IDX(g, 1, b, 0, 0, j, il)] == T_FLUID, f_r = flags[IDX(g, 1, b, 0, 0, j, ir)] == T_FLUID;
          const float v_l = i > 0 ? stick_phase1(g, U_in, flags, stick, b, 1, j, i - 1) : 0.f;
          const float v_r = i < W - 1 ? stick_phase1(g, U_in, flags, stick, b, 1, j, i + 1) : 0.f;
          if (f_l) v = -v_l;
          if (f_r) v = -v_r;
          if (f_l && f_r) v = 0.5f * ((-v_l) - v_r);
          const int f_d = flags[IDX(g, 1, b, 0, 0, jl, i)] == T_FLUID, f_u = flags[IDX(g, 1, b, 0, 0, jr, i)] == T_FLUID;
          const float u_d = j > 0 ? stick_phase1(g, U_in, flags, stick, b, 0, j - 1, i) : 0.f;
          const float u_u = j < H - 1 ? stick_phase1(g, U_in, flags, stick, b, 0, j + 1, i) : 0.f;
          if (f_d) u = -u_d;
          if (f_u) u = -u_u;
          if (f_d) u = 0.5f * ((-u_d) - u_u);                       /* :131 tests the lower neighbour twice */
        }
        const int ls = cont && stick[IDX(g, 1, b, 0, 0, j, il)] == T_STICK, rs = cont && stick[IDX(g, 1, b, 0, 0, j, ir)] == T_STICK;
        const int bs = cont && stick[IDX(g, 1, b, 0, 0, jl, i)] == T_STICK, us = cont && stick[IDX(g, 1, b, 0, 0, jr, i)] == T_STICK;
        if (2 * S + 2 * ls + bs + us == 3) u = 0.f;
        if (2 * S + ls + 2 * bs + rs == 3) v = 0.f;
        U_out[IDX(g, 2, b, 0, 0, j, i)] = u;
        U_out[IDX(g, 2, b, 1, 0, j, i)] = v;
      }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * setConstVals (simulate.py:4-26): U = U*mask + bc, rho = rho*mask + bc.  NULL skips a field.
 * ---------------------------------------------------------------------------------------- */
int ora_set_const_vals(const OraGrid* g, float* U, const float* UBC, const float* UBCInvMask, float* rho,
                       const float* rhoBC, const float* rhoBCInvMask) {
  const int nc = g->is3D ? 3 : 2;
  const size_t n1 = (size_t)g->B * g->D * g->H * g->W;
  if (U && UBC && UBCInvMask) {
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n1 * nc; ++q) { float t = U[q] * UBCInvMask[q]; U[q] = t + UBC[q]; }
  }
  if (rho && rhoBC && rhoBCInvMask) {
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n1; ++q) { float t = rho[q] * rhoBCInvMask[q]; rho[q] = t + rhoBC[q]; }
  }
  return 0;
}

/* flagsToOccupancy (flags_to_occupancy.py:6-19) */
int ora_flags_to_occupancy(const OraGrid* g, const float* flags, float* occ) {
  const size_t n1 = (size_t)g->B * g->D * g->H * g->W;
  for (size_t q = 0; q < n1; ++q) occ[q] = flags[q] == T_FLUID ? 0.f : (flags[q] == T_OBST ? 1.f : flags[q]);
  return 0;
}

/* emptyDomain (util.py:5-47): border of width bnd = obstacle, interior = fluid */
int ora_empty_domain(const OraGrid* g, float* flags, int bnd) {
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) flags[IDX(g, 1, b, 0, k, j, i)] = is_border(g, i, j, k, bnd) ? T_OBST : T_FLUID;
  return 0;
}

/* createCylinder (geometry_utils.py:26-33): (X - cx)^2 + (Y - cy)^2 <= r*r with int64 index grids promoted to fp32 and
 * the python scalars rounded to fp32 (r*r is squared in double first); every z plane. */
int ora_create_cylinder(const OraGrid* g, float* flags, double cx, double cy, double radius) {
  const float fcx = (float)cx, fcy = (float)cy, r2 = (float)(radius * radius);
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const float dx = (float)i - fcx, dy = (float)j - fcy;
          const float a = dx * dx, c = dy * dy;
          if (a + c <= r2) flags[IDX(g, 1, b, 0, k, j, i)] = T_OBST;
        }
  return 0;
}

/* createBox2D as its docstring describes it (geometry_utils.py:36-46; the body :59-62 cannot run): x0 <= x < x1, y0 <= y < y1 */
int ora_create_box2d(const OraGrid* g, float* flags, double x0, double x1, double y0, double y1) {
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i)
          if ((float)i >= (float)x0 && (float)i < (float)x1 && (float)j >= (float)y0 && (float)j < (float)y1)
            flags[IDX(g, 1, b, 0, k, j, i)] = T_OBST;
  return 0;
}

/* getCentered (grid.py:7-32): U (B,2|3,D,H,W) -> (B,3,D,H,W) */
int ora_get_centered(const OraGrid* g, const float* U, float* out) {
  const int nc = g->is3D ? 3 : 2;
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          out[IDX(g, 3, b, 0, k, j, i)] = i < g->W - 1 ? 0.5f * (U[IDX(g, nc, b, 0, k, j, i)] + U[IDX(g, nc, b, 0, k, j, i + 1)]) : 0.f;
          out[IDX(g, 3, b, 1, k, j, i)] = j < g->H - 1 ? 0.5f * (U[IDX(g, nc, b, 1, k, j, i)] + U[IDX(g, nc, b, 1, k, j + 1, i)]) : 0.f;
          out[IDX(g, 3, b, 2, k, j, i)] = (g->is3D && k < g->D - 1) ? 0.5f * (U[IDX(g, nc, b, 2, k, j, i)] + U[IDX(g, nc, b, 2, k + 1, j, i)]) : 0.f;
        }
  return 0;
}

/* Adjoints of velocityDivergence and velocityUpdate (what autograd computes over the reference's ATen chains,
 * velocity_divergence.py:46-74 and velocity_update.py:47-149), per cell. */
static int ora_active(const OraGrid* g, const float* flags, int b, int k, int j, int i) {
  return !is_border(g, i, j, k, 1) && flags[IDX(g, 1, b, 0, k, j, i)] != T_OBST;
}
int ora_velocity_divergence_backward(const OraGrid* g, const float* gdiv, const float* flags, float* gU) {
  const int nc = g->is3D ? 3 : 2;
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const float own = ora_active(g, flags, b, k, j, i) ? gdiv[IDX(g, 1, b, 0, k, j, i)] : 0.f;
          for (int a = 0; a < nc; ++a) {
            const int pi = i - (a == 0), pj = j - (a == 1), pk = k - (a == 2);
            float prev = 0.f;
            if (pi >= 0 && pj >= 0 && pk >= 0 && ora_active(g, flags, b, pk, pj, pi)) prev = gdiv[IDX(g, 1, b, 0, pk, pj, pi)];
            gU[IDX(g, nc, b, a, k, j, i)] = own - prev;
          }
        }
  return 0;
}
int ora_velocity_update_backward(const OraGrid* g, const float* gout, const float* flags, float* gU, float* gp) {
  const int nc = g->is3D ? 3 : 2;
  for (int b = 0; b < g->B; ++b)
    for (int k = 0; k < g->D; ++k)
      for (int j = 0; j < g->H; ++j)
        for (int i = 0; i < g->W; ++i) {
          const int inner = !is_border(g, i, j, k, 1);
          const float fc = flags[IDX(g, 1, b, 0, k, j, i)];
          float acc = 0.f;
          for (int a = 0; a < nc; ++a) {
            const float ga = gout[IDX(g, nc, b, a, k, j, i)];
            float gua = ga;
            if (inner) {
              const float fm = flags[IDX(g, 1, b, 0, k - (a == 2), j - (a == 1), i - (a == 0))];
              const int ff = fc == T_FLUID && fm == T_FLUID;
              const int fe = !g->is3D && fc == T_FLUID && fm == T_EMPTY, ef = !g->is3D && fc == T_EMPTY && fm == T_FLUID;
              gua = (ff || fe || ef) ? ga : 0.f;
              if (ff || fe) acc = acc - ga;
            }
            gU[IDX(g, nc, b, a, k, j, i)] = gua;
          }
          for (int a = 0; a < nc; ++a) {
            const int ni = i + (a == 0), nj = j + (a == 1), nk = k + (a == 2);
            if (ni < g->W && nj < g->H && nk < g->D && !is_border(g, ni, nj, nk, 1)) {
              const float fn = flags[IDX(g, 1, b, 0, nk, nj, ni)];
              const int ff = fn == T_FLUID && fc == T_FLUID, ef = !g->is3D && fn == T_EMPTY && fc == T_FLUID;
              if (ff || ef) acc = acc + gout[IDX(g, nc, b, a, nk, nj, ni)];
            }
          }
          gp[IDX(g, 1, b, 0, k, j, i)] = acc;
        }
  return 0;
}
