// 2D advection as LDS tile kernels (included by fnx_advect.hip inside its anonymous namespace, after fnx_advect_march.h,
// whose helpers -- alerp, atile_trace, alerp1d_fluid, the buffer resource -- it shares).
//
// The per-cell 2D kernels gather: ~35 global loads per cell in the forward pass, ~60 in the backward pass (two 9-cell clamp
// walks over two fields, two 4-corner samples, two 2 x 4-corner clamp boxes), and a vector-memory instruction costs the CU
// ~14 cycles whatever it hits (tools/ubench/ta_bench.hip).  At CFL < 1 everything a cell samples lies within two cells of
// it (the traced cell within one, its 3x3 clamp box within two), so a workgroup stages a 64 x 16-cell tile with a halo of
// two of every field in LDS with 16-byte loads and the samples become ds_reads -- what fnx_advect_march.h does along z,
// without the march.  A lane whose displacement is not below one cell, whose trace ends in a non-fluid cell, or (backward
// pass) whose forward pass traced further than one cell is not computed here: it is recorded in a bitmap (one 64-bit
// word per 64-cell row segment) and a fix-up launch runs the per-cell function on exactly those cells.  Every expression
// is the per-cell function's own, operand for operand (fnx_device.h, the functions above): same bits for every input.

constexpr int T2R = 16;                     // tile rows
constexpr int T2RPW = 2;                    // rows per wave
constexpr int T2NW = T2R / T2RPW;           // 8 waves
constexpr int T2HALO = 2;
constexpr int T2RR = T2R + 2 * T2HALO;      // rows held: j0-2 .. j0+17
constexpr int T2P = 68;                     // row pitch in floats: columns x0-2 .. x0+65 = 17 chunks of 16 bytes
constexpr int T2SZ = T2RR * T2P;            // floats per field
constexpr int T2CH = T2RR * (T2P / 4);      // 16-byte chunks per field: 340

struct T2Tile {
  int lane, w, x, j0, b, bx, ntx;
  int xs;                                    // LDS column of my cell = lane + xs (the tile's first LDS column is x0 - xs)
  unsigned off;                              // byte offset of this thread's chunk inside a field (threads 0 .. 339)
  int ldsoff;                                // float index of that chunk in a field's LDS image
  bool loader;
  __device__ __forceinline__ size_t word(const GridDims& g, int j) const { return ((size_t)b * g.H + j) * ntx + bx; }
};

__device__ __forceinline__ void t2_setup(T2Tile& m, const GridDims& g, int ntx, int nty) {
  m.lane = threadIdx.x & 63;
  m.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int t = blockIdx.x;
  m.bx = t % ntx; t /= ntx;
  const int by = t % nty;
  m.b = t / nty;
  m.ntx = ntx;
  m.x = m.bx * 64 + m.lane;
  m.j0 = by * T2R;
  // (as in the z-marching tiles: no chunk may start before column 0, so the first tile column holds columns 0 .. 67)
  m.xs = m.bx == 0 ? 0 : T2HALO;
  const int c = threadIdx.x;
  m.loader = c < T2CH;
  const int row = c / (T2P / 4), cq = c - row * (T2P / 4);
  int jr = m.j0 - T2HALO + row;
  jr = jr < 0 ? 0 : (jr > g.H - 1 ? g.H - 1 : jr);          // rows clamped into the grid; columns past W-1 read on (border cells only)
  m.off = (unsigned)(jr * g.W + (m.bx * 64 - m.xs) + 4 * cq) * 4u;
  m.ldsoff = row * T2P + 4 * cq;
}

// One field (a channel of a sample) -> its LDS image.  `cells_after`: cells between the channel's end and the tensor's (a chunk
// hanging over the channel end reads real memory; past the tensor the range check returns 0).
__device__ __forceinline__ void t2_load(const T2Tile& m, const GridDims& g, const float* chan, size_t cells_after, float* img) {
  const size_t left = (size_t)g.HW + cells_after;
  const ABuf rs = amake_rsrc(chan, (left > 0x3fffffffu ? 0x3fffffffu : (unsigned)left) * 4u);
  if (m.loader) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, m.off, 0, 0));
    *(f4*)(img + m.ldsoff) = v;
  }
}

// 3 x 3 fluid bits around (row, col) of a flags image: bit 3 (dy + 1) + (dx + 1)
__device__ __forceinline__ unsigned t2_fluid9(const float* fimg, int rc) {
  unsigned c = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) c |= (fimg[rc + (dy - 1) * T2P + dx - 1] == FNX_FLUID ? 1u : 0u) << (3 * dy + dx);
  return c;
}

// the 4 corners of a sample at the per-lane base cell (base -1 where n? is set): c[0] (x0,y0), c[1] (x1,y0), c[2] (x0,y1), c[3] (x1,y1)
__device__ __forceinline__ void t2_corners(const float* img, int rc0, bool nx, bool ny, float (&c)[4]) {
  const float* z = img + rc0 - (ny ? T2P : 0) - (nx ? 1 : 0);
  c[0] = z[0]; c[1] = z[1]; c[2] = z[T2P]; c[3] = z[T2P + 1];
}
__device__ __forceinline__ float t2_bilin(const float (&c)[4], const ALerp& L) {       // interpol<false>
  return (c[0] * L.t0 + c[2] * L.t1) * L.s0 + (c[1] * L.t0 + c[3] * L.t1) * L.s1;
}
// interpol_with_fluid<false, .>: corner fluid bits out of the 9 neighbourhood bits
__device__ __forceinline__ float t2_bilin_fluid(const float (&c)[4], unsigned nb9, const ALerp& L) {
  const unsigned q = nb9 >> ((L.ny ? 0u : 3u) + (L.nx ? 0u : 1u));
  const bool f0 = q & 1u, f1 = q & 2u, f2 = q & 8u, f3 = q & 16u;
  float vab, vcd, v; bool fab, fcd, fl;
  alerp1d_fluid(c[0], f0, c[2], f2, L.t0, L.t1, vab, fab);
  alerp1d_fluid(c[1], f1, c[3], f3, L.t0, L.t1, vcd, fcd);
  alerp1d_fluid(vab, fab, vcd, fcd, L.s0, L.s1, v, fl);
  const float plain = (c[0] * L.t0 + c[2] * L.t1) * L.s0 + (c[1] * L.t0 + c[3] * L.t1) * L.s1;
  return fl ? v : plain;
}

// ---------------------------------------------------------------------------------------------------
// Forward pass: sl_scalar_cell<false> (density) + sl_mac_cell_flat<false> (velocity).  LDS fields: rho, Ux, Uy, flags.
// WHAT: bit 0 = the density part, bit 1 = the velocity part (3: both advections of a step; 1 / 2: stand-alone advectScalar / advectVel).
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE, int WHAT>
__global__ __launch_bounds__(64 * T2NW) void advect2d_fwd_tile_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                                      const float* __restrict__ U,
                                                                      const float* __restrict__ flags,
                                                                      float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                                      float* __restrict__ U_fwd,
                                                                      unsigned long long* __restrict__ fix_s,
                                                                      unsigned long long* __restrict__ fix_v, int ntx, int nty) {
  __shared__ __attribute__((aligned(16))) float tl[4][T2SZ];
  T2Tile m;
  t2_setup(m, g, ntx, nty);
  const size_t sb1 = (size_t)m.b * g.DHW, sb2 = (size_t)m.b * 2 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after2 = 2 * after1;
  constexpr bool DO_S = (WHAT & 1) != 0, DO_V = (WHAT & 2) != 0;
  if constexpr (DO_S) t2_load(m, g, rho + sb1, after1, tl[0]);
  t2_load(m, g, U + sb2, after2 + g.DHW, tl[1]);
  t2_load(m, g, U + sb2 + g.DHW, after2, tl[2]);
  t2_load(m, g, flags + sb1, after1, tl[3]);
  __syncthreads();
  const int lane = m.lane, w = m.w, i = m.x, col = lane + m.xs;
  const bool xin = i < g.W;
  const float ndt = -dt;
#pragma unroll
  for (int r = 0; r < T2RPW; ++r) {
    const int j = m.j0 + T2RPW * w + r;
    const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2);
    const bool live = xin & (j < g.H);
    const float fi = (float)i, fj = (float)j;
    const float ctrx = fi + 0.5f, ctry = fj + 0.5f;
    const int rc0 = (T2HALO + T2RPW * w + r) * T2P + col;
    const unsigned nb9 = t2_fluid9(tl[3], rc0);
    const bool fluid = (nb9 >> 4) & 1u;
    const float x_c = tl[1][rc0], y_c = tl[2][rc0], x_r = tl[1][rc0 + 1], y_u = tl[2][rc0 + T2P];
    // ================= density: sl_scalar_cell =================
    float o_rho = 0.f; int o_cell = 0; unsigned long long ws = 0ull;
    if constexpr (DO_S) {
      const float cen0 = 0.5f * (x_c + x_r), cen1 = 0.5f * (y_c + y_u), cen2 = 0.f;          // get_centered<false>
      float p0, p1, p2;
      const bool traced = atile_trace(ndt * cen0, ndt * cen1, ndt * cen2, ctrx, ctry, 0.5f, i, j, 0, nb9 << 9, p0, p1, p2);
      const ALerp Ls = alerp(p0, p1, 0.5f, fi, fj, 0.f);
      float cs[4];
      t2_corners(tl[0], rc0, Ls.nx, Ls.ny, cs);
      // (wave-uniform shortcut as in the 3D tiles: with an all-fluid neighbourhood interpol_with_fluid IS the plain expression)
      const bool allfluid = SAMPLE_OUTSIDE || __builtin_amdgcn_ballot_w64(!border & fluid & (nb9 != 0x1ffu)) == 0;
      const float smp = allfluid ? t2_bilin(cs, Ls) : t2_bilin_fluid(cs, nb9, Ls);
      const float rho_c = tl[0][rc0];
      o_rho = border ? 0.f : (fluid ? smp : rho_c);
      const bool keep = border | !fluid;                   // p = ctr
      const float q0 = keep ? ctrx : p0, q1 = keep ? ctry : p1;
      const int ci = clampi((int)q0, 0, g.W - 1), cj = clampi((int)q1, 0, g.H - 1);
      o_cell = (cj << 16) | ci;
      ws = __builtin_amdgcn_ballot_w64(live & !keep & (!traced | !Ls.ok));
    }
    // ================= velocity: sl_mac_cell_flat =================
    float o_u[2] = {0.f, 0.f};
    bool okv = true;
    if constexpr (DO_V) {
    {
      const float v0 = x_c, v1 = 0.25f * (((y_c + tl[2][rc0 - 1]) + y_u) + tl[2][rc0 + T2P - 1]);       // get_at_mac<false, ., 0>
      const ALerp L = alerp(ctrx + v0 * ndt, ctry + v1 * ndt, 0.5f, fi, fj, 0.f);
      float c[4]; t2_corners(tl[1], rc0, L.nx, L.ny, c);
      o_u[0] = fluid ? t2_bilin(c, L) : y_c;              // non-fluid cell: channel 1 into channel 0 (:413-416)
      okv &= L.ok;
    }
    {
      const float v0 = 0.25f * (((x_c + tl[1][rc0 - T2P]) + x_r) + tl[1][rc0 - T2P + 1]), v1 = y_c;     // get_at_mac<false, ., 1>
      const ALerp L = alerp(ctrx + v0 * ndt, ctry + v1 * ndt, 0.5f, fi, fj, 0.f);
      float c[4]; t2_corners(tl[2], rc0, L.nx, L.ny, c);
      o_u[1] = fluid ? t2_bilin(c, L) : 0.f;
      okv &= L.ok;
    }
    }
    const unsigned long long wv = DO_V ? __builtin_amdgcn_ballot_w64(live & !border & fluid & !okv) : 0ull;
    if (border) { o_u[0] = 0.f; o_u[1] = 0.f; }
    if (live) {
      const size_t o = (size_t)j * g.W + i;
      if constexpr (DO_S) {
        rho_fwd[sb1 + o] = o_rho;
        cell_out[sb1 + o] = o_cell;
      }
      if constexpr (DO_V) {
        U_fwd[sb2 + o] = o_u[0];
        U_fwd[sb2 + g.DHW + o] = o_u[1];
      }
    }
    if (lane == 0 && j < g.H) {
      const size_t wi = m.word(g, j);
      if constexpr (DO_S) fix_s[wi] = ws;
      if constexpr (DO_V) fix_v[wi] = wv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Backward pass: sl_scalar_bwd_clamp_cell<false> + sl_mac_bwd_clamp_cell_flat<false> (self-advection: orig == U).
// LDS fields: rho_fwd, Ux, Uy, flags, rho (the clamp walks the 3x3 box of the traced cell: halo 2), U_fwd x, y.
// WHAT as in the forward kernel.
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE, int WHAT>
__global__ __launch_bounds__(64 * T2NW) void advect2d_bwd_tile_kernel(GridDims g, float dt, float half_s,
                                                                      const float* __restrict__ rho,
                                                                      const float* __restrict__ rho_fwd,
                                                                      const int* __restrict__ cell_in,
                                                                      const float* __restrict__ U,
                                                                      const float* __restrict__ U_fwd,
                                                                      const float* __restrict__ flags,
                                                                      float* __restrict__ rho_dst, float* __restrict__ U_dst,
                                                                      unsigned long long* __restrict__ fix_s,
                                                                      unsigned long long* __restrict__ fix_v, int ntx, int nty) {
  __shared__ __attribute__((aligned(16))) float tl[7][T2SZ];
  T2Tile m;
  t2_setup(m, g, ntx, nty);
  const size_t sb1 = (size_t)m.b * g.DHW, sb2 = (size_t)m.b * 2 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after2 = 2 * after1;
  constexpr bool DO_S = (WHAT & 1) != 0, DO_V = (WHAT & 2) != 0;
  if constexpr (DO_S) t2_load(m, g, rho_fwd + sb1, after1, tl[0]);
  t2_load(m, g, U + sb2, after2 + g.DHW, tl[1]);
  t2_load(m, g, U + sb2 + g.DHW, after2, tl[2]);
  t2_load(m, g, flags + sb1, after1, tl[3]);
  if constexpr (DO_S) t2_load(m, g, rho + sb1, after1, tl[4]);
  if constexpr (DO_V) {
    t2_load(m, g, U_fwd + sb2, after2 + g.DHW, tl[5]);
    t2_load(m, g, U_fwd + sb2 + g.DHW, after2, tl[6]);
  }
  const int lane = m.lane, w = m.w, i = m.x, col = lane + m.xs;
  const bool xin = i < g.W;
  // the traced cells of my rows (per-cell global operands), issued before the barrier
  int cell[T2RPW];
#pragma unroll
  for (int r = 0; r < T2RPW; ++r) {
    const int j = m.j0 + T2RPW * w + r;
    cell[r] = DO_S ? cell_in[sb1 + (size_t)(j < g.H ? j : g.H - 1) * g.W + (xin ? i : g.W - 1)] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < T2RPW; ++r) {
    const int j = m.j0 + T2RPW * w + r;
    const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2);
    const bool live = xin & (j < g.H);
    const float fi = (float)i, fj = (float)j;
    const float ctrx = fi + 0.5f, ctry = fj + 0.5f;
    const int rc0 = (T2HALO + T2RPW * w + r) * T2P + col;
    const unsigned nb9 = t2_fluid9(tl[3], rc0);
    const bool fluid = (nb9 >> 4) & 1u;
    const float x_c = tl[1][rc0], y_c = tl[2][rc0], x_r = tl[1][rc0 + 1], y_u = tl[2][rc0 + T2P];
    // ================= density: sl_scalar_bwd_clamp_cell =================
    float o_d = 0.f; unsigned long long ws = 0ull;
    if constexpr (DO_S) {
    const float f = tl[0][rc0];
    const float cen0 = 0.5f * (x_c + x_r), cen1 = 0.5f * (y_c + y_u), cen2 = 0.f;
    float p0, p1, p2;                                       // displacement (-ndt) * cen with ndt = -dt
    const bool traced = atile_trace(dt * cen0, dt * cen1, dt * cen2, ctrx, ctry, 0.5f, i, j, 0, nb9 << 9, p0, p1, p2);
    const ALerp Ls = alerp(p0, p1, 0.5f, fi, fj, 0.f);
    float cs[4];
    t2_corners(tl[0], rc0, Ls.nx, Ls.ny, cs);
    const bool allfluid = SAMPLE_OUTSIDE || __builtin_amdgcn_ballot_w64(!border & fluid & (nb9 != 0x1ffu)) == 0;
    const float smp = allfluid ? t2_bilin(cs, Ls) : t2_bilin_fluid(cs, nb9, Ls);
    const float bwd = border ? 0.f : (fluid ? smp : f);
    float d = f;
    if (fluid) d = f + half_s * (tl[4][rc0] - bwd);        // applied on border cells too (reference :371)
    // clamp: the 3x3 box of the traced cell (j0 << 16 | i0), members inside the grid and -- unless SAMPLE_OUTSIDE -- fluid
    const int tj = (int)((unsigned)cell[r] >> 16), ti = cell[r] & 0xffff;   // (unsigned: rows >= 32768 set the sign bit)
    const int di = ti - i, dj = tj - j;
    const bool nearc = ((unsigned)(di + 1) <= 2u) & ((unsigned)(dj + 1) <= 2u);
    const int rct = rc0 + (nearc ? dj * T2P + di : 0);
    float mn = INFINITY, mx = -INFINITY; bool any = false;
#pragma unroll
    for (int ej = -1; ej <= 1; ++ej)
#pragma unroll
      for (int ei = -1; ei <= 1; ++ei) {
        const int ii = ti + ei, jj = tj + ej;
        const bool vi = (jj >= 0) & (jj < g.H) & (ii >= 0) & (ii < g.W);
        const float s = tl[4][rct + ej * T2P + ei];
        const bool ok = vi & (SAMPLE_OUTSIDE || tl[3][rct + ej * T2P + ei] == FNX_FLUID);
        mn = ok ? fminf(mn, s) : mn;
        mx = ok ? fmaxf(mx, s) : mx;
        any = any | ok;
      }
    const float dc = any ? fmaxf(mn, fminf(mx, d)) : f;
    o_d = border ? d : dc;
    ws = __builtin_amdgcn_ballot_w64(live & !border & ((fluid & (!traced | !Ls.ok)) | !nearc));
    }
    // ================= velocity: sl_mac_bwd_clamp_cell_flat =================
    float o_u[2] = {0.f, 0.f};
    bool ok = true;
    if constexpr (DO_V) {
    const bool fmx = (nb9 >> 3) & 1u, fmy = (nb9 >> 1) & 1u;             // flags of the -1 neighbours along x, y
    float v[2][2];
    v[0][0] = x_c; v[0][1] = 0.25f * (((y_c + tl[2][rc0 - 1]) + y_u) + tl[2][rc0 + T2P - 1]);
    v[1][0] = 0.25f * (((x_c + tl[1][rc0 - T2P]) + x_r) + tl[1][rc0 - T2P + 1]); v[1][1] = y_c;
    const float fwd0 = tl[5][rc0], fwd1 = tl[6][rc0];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float vd0 = v[a][0] * dt, vd1 = v[a][1] * dt;
      const ALerp L = alerp(ctrx + vd0, ctry + vd1, 0.5f, fi, fj, 0.f);
      float c[4];
      t2_corners(tl[5 + a], rc0, L.nx, L.ny, c);
      const float smpv = t2_bilin(c, L);
      float bmn = INFINITY, bmx = -INFINITY;
      bool okc = true;
#pragma unroll
      for (int l = 0; l < 2; ++l) {                        // doClampComponentMAC: the boxes at trunc(pos -/+ vd)
        const int qx = (int)(l == 0 ? fi - vd0 : fi + vd0);
        const int qy = (int)(l == 0 ? fj - vd1 : fj + vd1);
        const int rx = qx - i, ry = qy - j;
        okc &= ((unsigned)(rx + 1) <= 1u) & ((unsigned)(ry + 1) <= 1u);
        float e[4];
        t2_corners(tl[1 + a], rc0, rx == -1, ry == -1, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { bmn = fminf(bmn, e[q]); bmx = fmaxf(bmx, e[q]); }
      }
      const float fa = a == 0 ? fwd0 : fwd1;
      const float og = a == 0 ? x_c : y_c;
      const float bw = fluid ? smpv : (a == 0 ? fwd1 : 0.f);               // Q1 pass-through of SL(fwd)
      const bool skip = !fluid | !(a == 0 ? fmx : fmy);
      const float corr = skip ? fa : fa + half_s * (og - bw);
      o_u[a] = border ? 0.f : fmaxf(fminf(corr, bmx), bmn);
      ok &= okc & (L.ok | !fluid);
    }
    }
    const unsigned long long wv = DO_V ? __builtin_amdgcn_ballot_w64(live & !border & !ok) : 0ull;
    if (live) {
      const size_t o = (size_t)j * g.W + i;
      if constexpr (DO_S) rho_dst[sb1 + o] = o_d;
      if constexpr (DO_V) {
        U_dst[sb2 + o] = o_u[0];
        U_dst[sb2 + g.DHW + o] = o_u[1];
      }
    }
    if (lane == 0 && j < g.H) {
      const size_t wi = m.word(g, j);
      if constexpr (DO_S) fix_s[wi] = ws;
      if constexpr (DO_V) fix_v[wi] = wv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Fix-up launches: the per-cell functions on the cells the tile kernels flagged (one thread per bitmap word).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool t2fix_decode(const GridDims& g, int ntx, size_t t, CellId& c, size_t& wi) {
  const size_t nrow = (size_t)g.B * g.H;
  if (t >= nrow * ntx) return false;
  const int bx = (int)(t % ntx); const size_t q = t / ntx;
  c.j = (int)(q % g.H); c.b = (int)(q / g.H); c.k = 0;
  c.i = bx * 64; c.valid = true;
  wi = ((size_t)c.b * g.H + c.j) * ntx + bx;
  return true;
}

template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256) void advect2d_fwd_fix_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                               const float* __restrict__ U, const float* __restrict__ flags,
                                                               float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                               float* __restrict__ U_fwd,
                                                               const unsigned long long* __restrict__ fix_s,
                                                               const unsigned long long* __restrict__ fix_v, int ntx) {
  CellId c; size_t wi;
  if (!t2fix_decode(g, ntx, (size_t)blockIdx.x * 256 + threadIdx.x, c, wi)) return;
  const unsigned long long ws = fix_s ? fix_s[wi] : 0ull, wv = fix_v ? fix_v[wi] : 0ull;   // (a stand-alone advection has one bitmap)
  const int i0 = c.i;
  for (unsigned long long a = ws | wv; a != 0; a &= a - 1) {
    const int bit = __builtin_ctzll(a);
    c.i = i0 + bit;
    if ((ws >> bit) & 1ull) sl_scalar_cell<false, false, SAMPLE_OUTSIDE>(g, c, dt, rho, U, flags, rho_fwd, cell_out);
    if ((wv >> bit) & 1ull) sl_mac_cell_flat<false>(g, c, dt, U, U, flags, U_fwd);
  }
}

template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256) void advect2d_bwd_fix_kernel(GridDims g, float dt, float half_s, const float* __restrict__ rho,
                                                               const float* __restrict__ rho_fwd, const int* __restrict__ cell_in,
                                                               const float* __restrict__ U, const float* __restrict__ U_fwd,
                                                               const float* __restrict__ flags, float* __restrict__ rho_dst,
                                                               float* __restrict__ U_dst,
                                                               const unsigned long long* __restrict__ fix_s,
                                                               const unsigned long long* __restrict__ fix_v, int ntx) {
  CellId c; size_t wi;
  if (!t2fix_decode(g, ntx, (size_t)blockIdx.x * 256 + threadIdx.x, c, wi)) return;
  const unsigned long long ws = fix_s ? fix_s[wi] : 0ull, wv = fix_v ? fix_v[wi] : 0ull;   // (a stand-alone advection has one bitmap)
  const int i0 = c.i;
  for (unsigned long long a = ws | wv; a != 0; a &= a - 1) {
    const int bit = __builtin_ctzll(a);
    c.i = i0 + bit;
    if ((ws >> bit) & 1ull) sl_scalar_bwd_clamp_cell<false, false, SAMPLE_OUTSIDE>(g, c, dt, half_s, rho, rho_fwd, cell_in, U, flags, nullptr, rho_dst);
    if ((wv >> bit) & 1ull) sl_mac_bwd_clamp_cell_flat<false>(g, c, dt, half_s, U, U_fwd, U, flags, U_dst);
  }
}
