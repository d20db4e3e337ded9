/*
 * cnn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's CNN pressure projection:
 *   MultiScaleNet.forward   pytorch/lib/multi_scale_net.py:118-127  (channel plan :111-116)
 *   FluidNet.forward        pytorch/lib/model.py:76-227 (ScaleNet variant; _ScaleNet :8-23)
 * The convolution / bilinear / std arithmetic itself lives in PyTorch (torch.nn.Conv2d,
 * F.upsample(mode='bilinear') == F.interpolate(align_corners=False), torch.std), a third-party
 * dependency whose version the reference does not pin (README says "Pytorch 0.4").  This file
 * restates the published algorithms (direct cross-correlation with zero padding k/2; half-pixel
 * bilinear with edge clamp; Bessel-corrected std) and is pinned against golden vectors produced
 * by torch 2.10 CPU on hash-seeded weights (tests/golden/cnn.npz) to <= 2e-5 relative: parity for
 * the CNN is a floating-point-tolerance statement, not a bit-exact one.
 * The 3D variant (Conv3d, trilinear) has no reference at all ("parity unpinned" by the reference;
 * the oracle is the same code with D > 1).
 *
 * Weight blob (shared convention with the product): for the 17 convs in the order
 *   convN_4[0..3], convN_2[0..5], convN_1[0..5], final
 * weight (Cout,Cin,kd,kh,kw) then bias (Cout), all fp32, torch memory order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int B, D, H, W, is3D, zoff, Dglob; } OraGrid;

int ora_velocity_divergence(const OraGrid* g, const float* U, const float* flags, float* div);
int ora_velocity_update(const OraGrid* g, const float* p, float* U, const float* flags);
int ora_set_wall_bcs(const OraGrid* g, float* U, const float* flags);

typedef struct { int cin, cout, k, relu; } Layer;
static const Layer T4[4] = { {2, 32, 3, 1}, {32, 64, 3, 1}, {64, 32, 3, 0}, {32, 1, 3, 0} };
static const Layer T2[6] = { {3, 32, 5, 1}, {32, 64, 3, 1}, {64, 128, 3, 1}, {128, 64, 3, 1}, {64, 32, 3, 0}, {32, 1, 3, 0} };
static const Layer T1[6] = { {3, 32, 5, 1}, {32, 64, 3, 1}, {64, 128, 3, 1}, {128, 64, 3, 1}, {64, 32, 3, 0}, {32, 8, 5, 0} };
static const Layer TF[1] = { {8, 1, 1, 0} };

/* Direct cross-correlation, zero padding k/2, NC(D)HW, double accumulation. */
static void conv_nd(const float* x, float* y, const float* w, const float* bias, int B, int cin, int cout,
                    int D, int H, int W, int k, int relu) {
  const int kd = D > 1 ? k : 1, pad = k / 2, pd = D > 1 ? pad : 0;
  const size_t plane = (size_t)H * W, vol = plane * D;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co)
      for (int z = 0; z < D; ++z)
        for (int yy = 0; yy < H; ++yy)
          for (int xx = 0; xx < W; ++xx) {
            double acc = bias[co];
            for (int ci = 0; ci < cin; ++ci) {
              const float* xin = x + ((size_t)b * cin + ci) * vol;
              const float* wk = w + (((size_t)co * cin + ci) * kd) * k * k;
              for (int a = 0; a < kd; ++a) {
                const int zz = z + a - pd;
                if (zz < 0 || zz >= D) continue;
                for (int r = 0; r < k; ++r) {
                  const int yr = yy + r - pad;
                  if (yr < 0 || yr >= H) continue;
                  for (int s = 0; s < k; ++s) {
                    const int xs = xx + s - pad;
                    if (xs < 0 || xs >= W) continue;
                    acc += (double)wk[(a * k + r) * k + s] * (double)xin[zz * plane + (size_t)yr * W + xs];
                  }
                }
              }
            }
            float v = (float)acc;
            if (relu && v < 0.f) v = 0.f;
            y[((size_t)b * cout + co) * vol + z * plane + (size_t)yy * W + xx] = v;
          }
}

/* torch upsample_{bi,tri}linear, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0. */
static inline void src_index(int dst, int in, int out, int* i0, int* i1, float* l0, float* l1) {
  const float scale = (float)in / (float)out;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  *i0 = (int)s;
  if (*i0 > in - 1) *i0 = in - 1;
  *i1 = *i0 + ((*i0 < in - 1) ? 1 : 0);
  *l1 = s - (float)*i0;
  *l0 = 1.f - *l1;
}

/* x (B*C, Di,Hi,Wi) -> y (B*C, Do,Ho,Wo) written at channel offset c_off of a (B, Ctot, ...) tensor.
 * zw (ora_multiscale_forward_crop; NULL = whole tensors): either tensor may hold a WINDOW of planes of a deeper notional tensor --
 * the interpolation runs in the notional depths zw[0] -> zw[2]; x holds the notional planes [zw[1], zw[1] + Di), y the notional
 * planes [zw[3], zw[3] + Do); a source plane outside x's window is clamped into it. */
static void resize_linear_win(const float* x, float* y, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                              int Ctot, int c_off, const int* zw) {
  const int full_in = zw ? zw[0] : Di, in_off = zw ? zw[1] : 0, full_out = zw ? zw[2] : Do, out_off = zw ? zw[3] : 0;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* xi = x + ((size_t)b * C + c) * Di * Hi * Wi;
      float* yo = y + ((size_t)b * Ctot + c_off + c) * Do * Ho * Wo;
      for (int z = 0; z < Do; ++z) {
        int z0, z1; float f0, f1; src_index(z + out_off, full_in, full_out, &z0, &z1, &f0, &f1);
        z0 -= in_off; z1 -= in_off;
        z0 = z0 < 0 ? 0 : (z0 > Di - 1 ? Di - 1 : z0);
        z1 = z1 < 0 ? 0 : (z1 > Di - 1 ? Di - 1 : z1);
        for (int j = 0; j < Ho; ++j) {
          int y0, y1; float t0, t1; src_index(j, Hi, Ho, &y0, &y1, &t0, &t1);
          for (int i = 0; i < Wo; ++i) {
            int x0, x1; float s0, s1; src_index(i, Wi, Wo, &x0, &x1, &s0, &s1);
#define XI(zz, yy, xx) xi[((size_t)(zz) * Hi + (yy)) * Wi + (xx)]
            float lo = t0 * (s0 * XI(z0, y0, x0) + s1 * XI(z0, y0, x1)) + t1 * (s0 * XI(z0, y1, x0) + s1 * XI(z0, y1, x1));
            float v = lo;
            if (full_in > 1 || full_out > 1) {
              float hi = t0 * (s0 * XI(z1, y0, x0) + s1 * XI(z1, y0, x1)) + t1 * (s0 * XI(z1, y1, x0) + s1 * XI(z1, y1, x1));
              v = f0 * lo + f1 * hi;
            }
#undef XI
            yo[((size_t)z * Ho + j) * Wo + i] = v;
          }
        }
      }
    }
}
static void resize_linear(const float* x, float* y, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                          int Ctot, int c_off) {
  resize_linear_win(x, y, B, C, Di, Hi, Wi, Do, Ho, Wo, Ctot, c_off, NULL);
}

static size_t layer_floats(const Layer* L, int is3D) {
  size_t kk = (size_t)L->k * L->k * (is3D ? L->k : 1);
  return (size_t)L->cout * L->cin * kk + L->cout;
}

static const float* run_tower(const Layer* T, int n, const float* wts, float** buf, int B, int D, int H, int W,
                              int is3D, float* in, float** out) {
  float* cur = in;
  for (int l = 0; l < n; ++l) {
    size_t kk = (size_t)T[l].k * T[l].k * (is3D ? T[l].k : 1);
    const float* w = wts; const float* bias = wts + (size_t)T[l].cout * T[l].cin * kk;
    float* o = (float*)malloc((size_t)B * T[l].cout * D * H * W * sizeof(float));
    conv_nd(cur, o, w, bias, B, T[l].cin, T[l].cout, D, H, W, T[l].k, T[l].relu);
    if (cur != in) free(cur);
    cur = o;
    wts += layer_floats(&T[l], is3D);
  }
  (void)buf;
  *out = cur;
  return wts;
}

size_t ora_scalenet_weight_floats(int is3D) {
  size_t n = 0;
  for (int l = 0; l < 4; ++l) n += layer_floats(&T4[l], is3D);
  for (int l = 0; l < 6; ++l) n += layer_floats(&T2[l], is3D);
  for (int l = 0; l < 6; ++l) n += layer_floats(&T1[l], is3D);
  n += layer_floats(&TF[0], is3D);
  return n;
}

/* MultiScaleNet.forward (multi_scale_net.py:118-127). x: (B,2,D,H,W) -> p: (B,1,D,H,W) */
int ora_multiscale_forward(const OraGrid* g, const float* wts, const float* x, float* p) {
  const int B = g->B, D = g->D, H = g->H, W = g->W, is3D = g->is3D;
  const int Dq = is3D ? (int)(D * 0.25) : 1, Hq = (int)(H * 0.25), Wq = (int)(W * 0.25);
  const int Dh = is3D ? (int)(D * 0.5) : 1, Hh = (int)(H * 0.5), Wh = (int)(W * 0.5);
  float* xq = (float*)malloc((size_t)B * 2 * Dq * Hq * Wq * sizeof(float));
  resize_linear(x, xq, B, 2, D, H, W, Dq, Hq, Wq, 2, 0);
  float* c4; wts = run_tower(T4, 4, wts, NULL, B, Dq, Hq, Wq, is3D, xq, &c4);
  float* in2 = (float*)malloc((size_t)B * 3 * Dh * Hh * Wh * sizeof(float));
  resize_linear(x, in2, B, 2, D, H, W, Dh, Hh, Wh, 3, 0);
  resize_linear(c4, in2, B, 1, Dq, Hq, Wq, Dh, Hh, Wh, 3, 2);
  float* c2; wts = run_tower(T2, 6, wts, NULL, B, Dh, Hh, Wh, is3D, in2, &c2);
  float* in1 = (float*)malloc((size_t)B * 3 * D * H * W * sizeof(float));
  resize_linear(x, in1, B, 2, D, H, W, D, H, W, 3, 0);
  resize_linear(c2, in1, B, 1, Dh, Hh, Wh, D, H, W, 3, 2);
  float* c1; wts = run_tower(T1, 6, wts, NULL, B, D, H, W, is3D, in1, &c1);
  float* fin; wts = run_tower(TF, 1, wts, NULL, B, D, H, W, is3D, c1, &fin);
  memcpy(p, fin, (size_t)B * D * H * W * sizeof(float));
  free(xq); free(c4); free(in2); free(c2); free(in1); free(c1); free(fin);
  return 0;
}

/* The forward pass on NESTED z-crops: not a reference function -- the checker of fnx_multiscale_forward_crop (include/fluidnet_hip.h),
 * which the z-slab driver's CNN projection runs on a rank's window.  The same towers and resampling as ora_multiscale_forward
 * (multi_scale_net.py:118-127), the quarter-resolution tower on all D planes of x, the half-resolution tower on the planes
 * [trim[2], D - trim[3]), the full-resolution tower on [trim[0], D - trim[1]) (full-resolution plane counts, multiples of 4 like D);
 * the resampling runs in the untrimmed grids' coordinates.  p: (B,1,D - trim[0] - trim[1],H,W). */
int ora_multiscale_forward_crop(const OraGrid* g, const float* wts, const float* x, const int* trim, float* p) {
  const int B = g->B, D = g->D, H = g->H, W = g->W;
  if (!g->is3D || D % 4 || trim[0] % 4 || trim[1] % 4 || trim[2] % 4 || trim[3] % 4 || trim[2] > trim[0] || trim[3] > trim[1]) return 1;
  const int Dq = (int)(D * 0.25), Hq = (int)(H * 0.25), Wq = (int)(W * 0.25);
  const int Dh = (int)(D * 0.5), Hh = (int)(H * 0.5), Wh = (int)(W * 0.5);
  const int f_lo = trim[0], f_n = D - trim[0] - trim[1], h_lo = trim[2] / 2, h_n = Dh - trim[2] / 2 - trim[3] / 2;
  float* xq = (float*)malloc((size_t)B * 2 * Dq * Hq * Wq * sizeof(float));
  resize_linear(x, xq, B, 2, D, H, W, Dq, Hq, Wq, 2, 0);
  float* c4; wts = run_tower(T4, 4, wts, NULL, B, Dq, Hq, Wq, 1, xq, &c4);
  float* in2 = (float*)malloc((size_t)B * 3 * h_n * Hh * Wh * sizeof(float));
  const int x_to_h[4] = { D, 0, Dh, h_lo }, q_to_h[4] = { Dq, 0, Dh, h_lo };
  resize_linear_win(x, in2, B, 2, D, H, W, h_n, Hh, Wh, 3, 0, x_to_h);
  resize_linear_win(c4, in2, B, 1, Dq, Hq, Wq, h_n, Hh, Wh, 3, 2, q_to_h);
  float* c2; wts = run_tower(T2, 6, wts, NULL, B, h_n, Hh, Wh, 1, in2, &c2);
  float* in1 = (float*)malloc((size_t)B * 3 * f_n * H * W * sizeof(float));
  const int x_to_f[4] = { D, 0, D, f_lo }, h_to_f[4] = { Dh, h_lo, D, f_lo };
  resize_linear_win(x, in1, B, 2, D, H, W, f_n, H, W, 3, 0, x_to_f);
  resize_linear_win(c2, in1, B, 1, h_n, Hh, Wh, f_n, H, W, 3, 2, h_to_f);
  float* c1; wts = run_tower(T1, 6, wts, NULL, B, f_n, H, W, 1, in1, &c1);
  float* fin; wts = run_tower(TF, 1, wts, NULL, B, f_n, H, W, 1, c1, &fin);
  memcpy(p, fin, (size_t)B * f_n * H * W * sizeof(float));
  free(xq); free(c4); free(in2); free(c2); free(in1); free(c1); free(fin);
  return 0;
}

/* _ScaleNet (model.py:8-23): clamp(std(UDiv.view(B,-1), unbiased), thr, inf) per sample */
int ora_scale_std(const OraGrid* g, const float* U, float thr, float* s) {
  const int nc = g->is3D ? 3 : 2;
  const size_t n = (size_t)nc * g->D * g->H * g->W;
  for (int b = 0; b < g->B; ++b) {
    const float* u = U + b * n;
    double sum = 0, sq = 0;
    for (size_t q = 0; q < n; ++q) sum += u[q];
    const double mean = sum / (double)n;
    for (size_t q = 0; q < n; ++q) { double d = u[q] - mean; sq += d * d; }
    float sd = (float)sqrt(sq / (double)(n - 1));
    s[b] = sd < thr ? thr : sd;
  }
  return 0;
}

/* FluidNet.forward (model.py:76-227, mconf: inputChannels={div}, normalizeInputChan=UDiv, ScaleNet).
 * in: (B, 5|6, D,H,W) = [p, U.., flags, rho];  p_out (B,1,..), U_out (B,2|3,..) */
int ora_fluidnet_forward(const OraGrid* g, const float* wts, const float* in, float thr, float* p_out, float* U_out) {
  const int nc = g->is3D ? 3 : 2, cin = nc + 3;
  const size_t n1 = (size_t)g->D * g->H * g->W;
  float* flags = (float*)malloc(g->B * n1 * sizeof(float));
  float* div = (float*)malloc(g->B * n1 * sizeof(float));
  float* x = (float*)malloc(g->B * 2 * n1 * sizeof(float));
  float* s = (float*)malloc(g->B * sizeof(float));
  for (int b = 0; b < g->B; ++b) {
    memcpy(U_out + (size_t)b * nc * n1, in + ((size_t)b * cin + 1) * n1, nc * n1 * sizeof(float));
    memcpy(flags + b * n1, in + ((size_t)b * cin + 1 + nc) * n1, n1 * sizeof(float));
  }
  ora_velocity_divergence(g, U_out, flags, div);
  ora_scale_std(g, U_out, thr, s);
  for (int b = 0; b < g->B; ++b) {
    for (size_t q = 0; q < nc * n1; ++q) U_out[(size_t)b * nc * n1 + q] = U_out[(size_t)b * nc * n1 + q] / s[b];
    for (size_t q = 0; q < n1; ++q) {
      x[((size_t)b * 2 + 0) * n1 + q] = div[b * n1 + q] / s[b];
      const float f = flags[b * n1 + q];
      x[((size_t)b * 2 + 1) * n1 + q] = f == 1.f ? 0.f : (f == 2.f ? 1.f : f);
    }
  }
  ora_multiscale_forward(g, wts, x, p_out);
  ora_velocity_update(g, p_out, U_out, flags);
  for (int b = 0; b < g->B; ++b) {
    for (size_t q = 0; q < n1; ++q) p_out[b * n1 + q] = p_out[b * n1 + q] * s[b];
    for (size_t q = 0; q < nc * n1; ++q) U_out[(size_t)b * nc * n1 + q] = U_out[(size_t)b * nc * n1 + q] * s[b];
  }
  ora_set_wall_bcs(g, U_out, flags);
  free(flags); free(div); free(x); free(s);
  return 0;
}
