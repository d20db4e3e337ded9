// 3D advection as z-marching tile kernels (included by fnx_advect.hip inside its anonymous namespace, after the
// per-cell functions it falls back to).
//
// Why.  The per-cell kernels above issue ~135 global gathers per cell, and the texture addresser charges ~14 cycles per
// wave-level VMEM instruction whatever its width (tools/ubench/ta_bench.hip): 135 x 14 cycles x 1024 waves per CU is the
// measured 1.06 ms at 16.8 M cells (TA busy 95 %, profiles/r01_pmc).  But at CFL < 1 -- every shipped configuration, and
// the bound the z-slab decomposition rests on anyway -- everything a cell samples lies in its own 3x3x3 neighbourhood:
// the trace end point is within one cell, so the 8 interpolation corners are {b, b+1}^3 with b in {-1, 0} per axis, and
// so are the 2 x 8 clamp corners of the velocity pass.  So the fields are streamed through LDS like a stencil:
//
//   * a workgroup (4 waves) owns a 64-column x 8-row tile and marches along z; the planes z-1, z, z+1 (+ the one in
//     flight) of rows j0-1 .. j0+8, columns x0-1 .. x0+66 of every field live in an LDS ring;
//   * a plane arrives as buffer_load_dwordx4 ... lds: one instruction moves three 272-byte rows (lane l: row l/17, 16-byte
//     chunk l%17) straight into the ring, no VGPRs, 20 instructions per plane for the whole tile instead of ~135 per
//     cell-wave; one barrier per plane hands it to the four waves;
//   * a sample's 8 corners are ds_reads at a per-lane LDS address (a ds_read costs the CU 2 cycles, a global gather 14);
//     the fixed-offset neighbours (face velocities) are ds_reads at immediate offsets;
//   * the flags become 3 bits x 4 rows per plane (fluid?) in registers, so the fluid-aware interpolation and the "is the
//     traced cell blocked" test never touch memory.
// A lane whose displacement is not below one cell, or whose trace ends in a non-fluid cell, takes the per-cell function
// above under its exec mask (same arithmetic, global gathers): the kernels are bit-identical to the per-cell ones for
// EVERY input, the fast path only has to be the common case.  Every expression below is the per-cell function's own,
// operand for operand (fnx_device.h); only where the operands come from differs.
// (A first version kept the planes in registers and picked the corners with v_cndmask networks -- 38 selects + 18 DPP
// moves per sample: 1540 VALU per two rows, slower than the gather kernel it replaced; the LDS does that selection for
// the price of an address.)

constexpr int ATR = 8;               // tile rows (2 per wave)
constexpr int ATRR = ATR + 2;        // rows held: j0-1 .. j0+8
constexpr int ATP = 68;              // row pitch in floats: columns x0-1 .. x0+66 = 17 chunks of 16 bytes
constexpr int ATNQ = (ATRR + 2) / 3; // DMA instructions per field-plane (3 rows each): 4

typedef __amdgpu_buffer_rsrc_t ABuf;
typedef __attribute__((address_space(3))) void* ALds;
__device__ __forceinline__ ABuf amake_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
template <int N> struct AIC { static constexpr int value = N; };

// lerp_setup (fnx_device.h) for a position whose base cell lands on {c-1, c} per axis.  The weights are lerp_setup's own
// expressions minus its clamps: for a non-negative coordinate s1 = p - trunc(p) is in [0, 1) and s0 = 1 - s1 in (0, 1], and
// clamp01 returns such a value unchanged (bit for bit).  n? = the base is c-1; ok = all three bases are in {c-1, c} (the
// only case the tile path covers; then the coordinates are positive and lerp_setup's index clamps are no-ops too).
struct ALerp { float s0, s1, t0, t1, f0, f1; bool nx, ny, nz, ok; };
__device__ __forceinline__ ALerp alerp(float px, float py, float pz, int i, int j, int kg) {
  ALerp L;
  px = px - 0.5f; py = py - 0.5f; pz = pz - 0.5f;
  const int qx = (int)px, qy = (int)py, qz = (int)pz;
  L.s1 = px - (float)qx; L.t1 = py - (float)qy; L.f1 = pz - (float)qz;
  L.s0 = 1.f - L.s1; L.t0 = 1.f - L.t1; L.f0 = 1.f - L.f1;
  const int rx = qx - i, ry = qy - j, rz = qz - kg;
  L.nx = rx == -1; L.ny = ry == -1; L.nz = rz == -1;
  // (a coordinate in (-1, 0) also truncates to 0 = c-1 for c = 1, but lerp_setup then clamps its negative weight: not this path)
  L.ok = ((unsigned)(rx + 1) <= 1u) & ((unsigned)(ry + 1) <= 1u) & ((unsigned)(rz + 1) <= 1u) & (px >= 0.f) & (py >= 0.f) & (pz >= 0.f);
  return L;
}

// interpol<true> on 8 corners c[(cz*2 + cy)*2 + cx]  (Ia = c[0] (x0,y0), Ib = c[2] (x0,y1), Ic = c[1] (x1,y0), Id = c[3])
__device__ __forceinline__ float atrilin(const float (&c)[8], const ALerp& L) {
  const float lo = (c[0] * L.t0 + c[2] * L.t1) * L.s0 + (c[1] * L.t0 + c[3] * L.t1) * L.s1;
  const float hi = (c[4] * L.t0 + c[6] * L.t1) * L.s0 + (c[5] * L.t0 + c[7] * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}

// lerp1d_fluid (fnx_device.h / grid.cpp:78-96) as selects
__device__ __forceinline__ void alerp1d_fluid(float a, bool fa, float b, bool fb, float ta, float tb, float& v, bool& fl) {
  const float mix = a * ta + b * tb;
  const float t = fb ? mix : a, u = fb ? b : 0.f;
  v = fa ? t : u;
  fl = fa | fb;
}

// interpol_with_fluid<true, false> on 8 corners and their fluid bits (bit n of fb <-> corner c[n])
__device__ __forceinline__ float atrilin_fluid(const float (&c)[8], unsigned fb, const ALerp& L) {
  const bool f0 = fb & 1u, f1 = fb & 2u, f2 = fb & 4u, f3 = fb & 8u, f4 = fb & 16u, f5 = fb & 32u, f6 = fb & 64u, f7 = fb & 128u;
  float vab, vcd, v; bool fab, fcd, fl;
  alerp1d_fluid(c[0], f0, c[2], f2, L.t0, L.t1, vab, fab);
  alerp1d_fluid(c[1], f1, c[3], f3, L.t0, L.t1, vcd, fcd);
  alerp1d_fluid(vab, fab, vcd, fcd, L.s0, L.s1, v, fl);
  float plain = (c[0] * L.t0 + c[2] * L.t1) * L.s0 + (c[1] * L.t0 + c[3] * L.t1) * L.s1;
  float vef, vgh, vhi; bool fef, fgh, fhi;
  alerp1d_fluid(c[4], f4, c[6], f6, L.t0, L.t1, vef, fef);
  alerp1d_fluid(c[5], f5, c[7], f7, L.t0, L.t1, vgh, fgh);
  alerp1d_fluid(vef, fef, vgh, fgh, L.s0, L.s1, vhi, fhi);
  const float vlo = v; const bool flo = fl;
  alerp1d_fluid(vlo, flo, vhi, fhi, L.f0, L.f1, v, fl);
  const float hi = (c[4] * L.t0 + c[6] * L.t1) * L.s0 + (c[5] * L.t0 + c[7] * L.t1) * L.s1;
  plain = plain * L.f0 + hi * L.f1;
  return fl ? v : plain;
}

// ---------------------------------------------------------------------------------------------------
// Tile frame shared by the forward and backward kernels.
// ---------------------------------------------------------------------------------------------------
struct ATile {
  int lane, w, x, j0, b, k_lo, k_hi, k0;
  int xs;                        // LDS column of my cell = lane + xs (the tile's first LDS column is x0 - xs)
  unsigned voff[ATNQ];           // per-lane byte offset of "my" row + chunk inside a plane, per DMA instruction
  bool dma_lane[ATNQ];
  unsigned hw;
  __device__ __forceinline__ unsigned planeoff(const GridDims& g, int k) const {   // bytes, relative to plane k0
    const int kc = k < 0 ? 0 : (k > g.D - 1 ? g.D - 1 : k);
    return (unsigned)(kc - k0) * hw * 4u;
  }
};

// blockIdx.x -> (tile, z chunk), renumbered like the Jacobi march (XCD q gets a band of neighbouring tiles).
__device__ __forceinline__ bool atile_setup(ATile& m, const GridDims& g, int ntx, int nty, int zchunk) {
  m.lane = threadIdx.x & 63;
  m.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x;
  const int gid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int ntiles = ntx * nty * g.B;
  const int zc = gid / ntiles, tl = gid - zc * ntiles;
  if (zc * zchunk >= g.KN) return false;
  const int bx = tl % ntx, l1 = tl / ntx;
  const int by = l1 % nty;
  m.b = l1 / nty;
  m.x = bx * 64 + m.lane;
  m.j0 = by * ATR;
  m.k_lo = g.K0 + zc * zchunk;
  m.k_hi = min(m.k_lo + zchunk, g.K0 + g.KN);
  m.hw = (unsigned)g.HW;
  m.k0 = m.k_lo - 1 < 0 ? 0 : m.k_lo - 1;
  // DMA instruction q moves held rows 3q .. 3q+2: lane l fetches the 16-byte chunk l%17 of row 3q + l/17 (51 lanes).
  // Rows are clamped into the grid; columns are not (a chunk is 4 columns): right of column W-1 it reads the next row's
  // cells (or 0 past the end of the tensor: the buffer range check) -- only border cells ever see those.  On the LEFT no
  // chunk may start before column 0: a negative offset fails the range check for the whole 16 bytes, columns 0..2
  // included, so the tiles of the first tile column hold columns 0 .. 67 (no left halo: column 0 is a border column and
  // never looks left) and every other tile holds x0-1 .. x0+66.
  m.xs = bx == 0 ? 0 : 1;
  const int rsub = m.lane / 17, cq = m.lane - rsub * 17;
#pragma unroll
  for (int q = 0; q < ATNQ; ++q) {
    const int hr = 3 * q + rsub;
    int jr = m.j0 - 1 + hr;
    jr = jr < 0 ? 0 : (jr > g.H - 1 ? g.H - 1 : jr);
    m.voff[q] = (unsigned)(jr * g.W + (bx * 64 - m.xs) + 4 * cq) * 4u;
    m.dma_lane[q] = (m.lane < 51) & (hr < ATRR);
  }
  return true;
}

// Buffer resource of one channel of the sample, based at plane k0 and ending with the TENSOR (`cells_after` = cells
// between the channel's end and the tensor's: a chunk hanging over the channel end reads real memory there); the plane
// offset travels in the VGPR offset because the range check does not see an SGPR offset.
__device__ __forceinline__ ABuf atile_rsrc(const ATile& m, const GridDims& g, const float* chan, size_t cells_after) {
  const size_t left = (size_t)(g.D - m.k0) * g.HW + cells_after;
  const unsigned ncell = left > 0x3fffffffu ? 0x3fffffffu : (unsigned)left;
  return amake_rsrc(chan + (size_t)m.k0 * g.HW, ncell * 4u);
}

// one field-plane (ATRR rows) -> LDS at `dst` ([ATRR][ATP] floats); the ATNQ instructions are dealt round-robin to
// the four waves starting with wave `first`
__device__ __forceinline__ void atile_dma(const ATile& m, const ABuf& rs, float* dst, unsigned plane_bytes, int first) {
#pragma unroll
  for (int q = 0; q < ATNQ; ++q) {
    if (((first + q) & 3) == m.w) {                       // wave-uniform
      if (m.dma_lane[q])
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ALds)(dst + 3 * q * ATP), 16, m.voff[q] + plane_bytes, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Forward pass: sl_scalar_cell (density) + sl_mac_cell_flat (velocity) for the planes [K0, K0+KN).
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256, 3) void advect3d_fwd_tile_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                                   const float* __restrict__ U,
                                                                   const float* __restrict__ flags,
                                                                   float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                                   float* __restrict__ U_fwd, int ntx, int nty, int zchunk) {
  constexpr int NF = 4;                                   // ring fields: rho, Ux, Uy, Uz
  constexpr int FSZ = ATRR * ATP;                         // floats per field-plane
  __shared__ __attribute__((aligned(16))) float ring[4][NF][FSZ];
  __shared__ __attribute__((aligned(16))) float fstage[2][FSZ];      // flags of the plane in flight / just landed
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  const ABuf rs_r = atile_rsrc(m, g, rho + sb1, after1), rs_x = atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW),
             rs_y = atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW), rs_z = atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3),
             rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w;
  auto dma_plane = [&](int slot, int fs, int k) {          // plane k -> ring slot `slot`, flags -> fstage[fs]
    const unsigned pb = m.planeoff(g, k);
    atile_dma(m, rs_r, ring[slot][0], pb, 0);
    atile_dma(m, rs_x, ring[slot][1], pb, 1);
    atile_dma(m, rs_y, ring[slot][2], pb, 2);
    atile_dma(m, rs_z, ring[slot][3], pb, 3);
    atile_dma(m, rs_f, fstage[fs], pb, 0);
  };
  // my two rows: held rows hr0 = 2w+1, 2w+2 (tile rows 2w, 2w+1); my LDS column: lane + xs
  const int hr0 = 2 * w + 1;
  const int col = lane + m.xs;
  // fluid bits of a plane for my rows hr0-1 .. hr0+2: bit 3*rr + (dx+1)
  auto fluid_bits = [&](int fs) {
    unsigned c = 0;
    const float* f = &fstage[fs][(hr0 - 1) * ATP + col - 1];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) c |= (f[rr * ATP + dx] == FNX_FLUID ? 1u : 0u) << (3 * rr + dx);
    }
    return c;
  };

  const int i = m.x;
  const bool xin = i < g.W;
  const float ndt = -dt;
  unsigned FB[3];                                         // fluid bits of planes k-1, k, k+1 (rotated by hand below)

  // one output plane k.  SC = ring slot of plane k; planes k-1 / k+1 sit in slots (SC+3)%4 / (SC+1)%4.
  auto step = [&](auto rsl, int k, unsigned fbm, unsigned fbc, unsigned fbp) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value, SM = (SC + 3) % 4, SP = (SC + 1) % 4;
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    bool slow_s[2], slow_v[2];
    bool any_slow = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = m.j0 + 2 * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const float ctrx = (float)i + 0.5f, ctry = (float)j + 0.5f;
      // LDS word of field f, plane slot s, at my cell + (dy, dx)
      auto at = [&](int f, int s, int dy, int dx) { return ring[s][f][(hr0 + r + dy) * ATP + col + dx]; };
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      // 27 fluid bits of the neighbourhood: bit 9*(dz+1) + 3*(dy+1) + (dx+1)
      const unsigned nb = ((fbm >> (3 * r)) & 0x1ffu) | (((fbc >> (3 * r)) & 0x1ffu) << 9) | (((fbp >> (3 * r)) & 0x1ffu) << 18);
      // the 8 corners of a sample of field f: per-lane LDS offsets of the base cell on the two z levels
      auto corners = [&](int f, const ALerp& L, float (&c)[8]) {
        const int rowcol = (hr0 + r - (L.ny ? 1 : 0)) * ATP + col - (L.nx ? 1 : 0);
        const float* z0 = (L.nz ? &ring[SM][f][0] : &ring[SC][f][0]) + rowcol;
        const float* z1 = (L.nz ? &ring[SC][f][0] : &ring[SP][f][0]) + rowcol;
        c[0] = z0[0]; c[1] = z0[1]; c[2] = z0[ATP]; c[3] = z0[ATP + 1];
        c[4] = z1[0]; c[5] = z1[1]; c[6] = z1[ATP]; c[7] = z1[ATP + 1];
      };

      // ================= density: sl_scalar_cell =================
      const float x_c = at(1, SC, 0, 0), y_c = at(2, SC, 0, 0), z_c = at(3, SC, 0, 0);
      const float x_r = at(1, SC, 0, 1), y_u = at(2, SC, 1, 0), z_f = at(3, SP, 0, 0);
      const float cen0 = 0.5f * (x_c + x_r);                                    // get_centered
      const float cen1 = 0.5f * (y_c + y_u);
      const float cen2 = 0.5f * (z_c + z_f);
      const float d0 = ndt * cen0, d1 = ndt * cen1, d2 = ndt * cen2;
      // line_trace from the centre of a fluid cell (fnx_device.h): either it stays (length <= eps / <= margin) or it is ONE
      // step of length min(|d|, 1) that must end the loop and land in a fluid cell of the neighbourhood
      const float length = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
      const bool stay = (length <= FNX_EPSILON) | (0.f >= length - FNX_HIT_MARGIN);
      const float dir0 = d0 / length, dir1 = d1 / length, dir2 = d2 / length;
      const float stp = fminf(length - 0.f, 1.f);
      const float n0 = ctrx + dir0 * stp, n1 = ctry + dir1 * stp, n2 = ctrz + dir2 * stp;
      const bool ends = stp >= length - FNX_HIT_MARGIN;                        // the second iteration's exit test
      const int c0 = (int)n0 - i, c1 = (int)n1 - j, c2 = (int)n2 - kg;          // traced cell relative to this one
      const bool near = ((unsigned)(c0 + 1) <= 2u) & ((unsigned)(c1 + 1) <= 2u) & ((unsigned)(c2 + 1) <= 2u);
      const bool tfluid = (nb >> ((9 * (c2 + 1) + 3 * (c1 + 1) + (c0 + 1)) & 31)) & 1u;
      const bool moved_ok = ends & near & tfluid;
      const float p0 = stay ? ctrx : n0, p1 = stay ? ctry : n1, p2 = stay ? ctrz : n2;
      const ALerp Ls = alerp(p0, p1, p2, i, j, kg);
      float cs[8];
      corners(0, Ls, cs);
      float smp;
      if (SAMPLE_OUTSIDE) {
        smp = atrilin(cs, Ls);
      } else {
        // fluid bits of the 8 corners: base bit 9*(bz+1) + 3*(by+1) + (bx+1), corner offsets 0,1,3,4,9,10,12,13
        const unsigned sh = (Ls.nz ? 0u : 9u) + (Ls.ny ? 0u : 3u) + (Ls.nx ? 0u : 1u);
        const unsigned q = nb >> sh;
        const unsigned fb = (q & 1u) | ((q >> 1) & 1u) << 1 | ((q >> 3) & 1u) << 2 | ((q >> 4) & 1u) << 3 | ((q >> 9) & 1u) << 4 |
                            ((q >> 10) & 1u) << 5 | ((q >> 12) & 1u) << 6 | ((q >> 13) & 1u) << 7;
        smp = atrilin_fluid(cs, fb, Ls);
      }
      const float rho_c = at(0, SC, 0, 0);
      const float val = border ? 0.f : (fluid ? smp : rho_c);
      const bool keep = border | !fluid;                   // p = ctr
      const float q0 = keep ? ctrx : p0, q1 = keep ? ctry : p1, q2 = keep ? ctrz : p2;
      const int ci = clampi((int)q0, 0, g.W - 1), cj = clampi((int)q1, 0, g.H - 1);
      const int ck = clampi((int)q2, 0, g.Dglob - 1) - g.zoff;
      const int cell = (ck + 1) * g.HW + cj * g.W + ci;
      slow_s[r] = !keep & (!(stay | moved_ok) | !Ls.ok);

      // ================= velocity: sl_mac_cell_flat =================
      // get_at_mac<true, false, 0/1/2> (fnx_device.h), operand for operand
      float v0[3], v1[3], v2[3];
      v0[0] = x_c;
      v0[1] = 0.25f * (((y_c + at(2, SC, 0, -1)) + y_u) + at(2, SC, 1, -1));
      v0[2] = 0.25f * (((z_c + at(3, SC, 0, -1)) + z_f) + at(3, SP, 0, -1));
      v1[0] = 0.25f * (((x_c + at(1, SC, -1, 0)) + x_r) + at(1, SC, -1, 1));
      v1[1] = y_c;
      v1[2] = 0.25f * (((z_c + at(3, SC, -1, 0)) + z_f) + at(3, SP, -1, 0));
      v2[0] = 0.25f * (((x_c + at(1, SM, 0, 0)) + x_r) + at(1, SM, 0, 1));
      v2[1] = 0.25f * (((y_c + at(2, SM, 0, 0)) + y_u) + at(2, SM, 1, 0));
      v2[2] = z_c;
      float uo[3];
      bool okv = true;
      {
        const ALerp L = alerp(ctrx + v0[0] * ndt, ctry + v0[1] * ndt, ctrz + v0[2] * ndt, i, j, kg);
        float c[8]; corners(1, L, c);
        uo[0] = fluid ? atrilin(c, L) : y_c;              // non-fluid cell: channel 1 into channel 0 (:413-416)
        okv &= L.ok;
      }
      {
        const ALerp L = alerp(ctrx + v1[0] * ndt, ctry + v1[1] * ndt, ctrz + v1[2] * ndt, i, j, kg);
        float c[8]; corners(2, L, c);
        uo[1] = fluid ? atrilin(c, L) : 0.f;
        okv &= L.ok;
      }
      {
        const ALerp L = alerp(ctrx + v2[0] * ndt, ctry + v2[1] * ndt, ctrz + v2[2] * ndt, i, j, kg);
        float c[8]; corners(3, L, c);
        uo[2] = fluid ? atrilin(c, L) : z_c;
        okv &= L.ok;
      }
      slow_v[r] = !border & fluid & !okv;
      if (border) { uo[0] = 0.f; uo[1] = 0.f; uo[2] = 0.f; }

      if (xin && j < g.H) {
        const size_t o = (size_t)k * g.HW + (size_t)j * g.W + i;
        rho_fwd[sb1 + o] = val;
        cell_out[sb1 + o] = cell;
        U_fwd[sb3 + o] = uo[0];
        U_fwd[sb3 + g.DHW + o] = uo[1];
        U_fwd[sb3 + 2 * (size_t)g.DHW + o] = uo[2];
      } else {
        slow_s[r] = false; slow_v[r] = false;
      }
      any_slow |= slow_s[r] | slow_v[r];
    }
    // lanes the neighbourhood path does not cover (|displacement| >= 1 cell, trace into a non-fluid cell): the per-cell
    // function redoes them from memory
    if (__builtin_amdgcn_ballot_w64(any_slow) != 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        CellId c; c.b = m.b; c.k = k; c.j = m.j0 + 2 * w + r; c.i = i; c.valid = true;
        if (slow_s[r]) sl_scalar_cell<true, false, SAMPLE_OUTSIDE>(g, c, dt, rho, U, flags, rho_fwd, cell_out);
        if (slow_v[r]) sl_mac_cell_flat<true>(g, c, dt, U, U, flags, U_fwd);
      }
    }
  };

  // ---- prologue.  Ring slot of plane k: (k - k_lo + 1) mod 4, so the march always enters at the same phase.
  int k = m.k_lo;
  dma_plane(0, 0, k - 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): my share of the plane has landed
  __syncthreads();                                        // ... and everybody else's
  FB[0] = fluid_bits(0);
  dma_plane(1, 1, k);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FB[1] = fluid_bits(1);
  dma_plane(2, 0, k + 1);                                 // (everybody read fstage[0] before the barrier above)
  int fs = 0;                                             // fstage holding the flags of plane k+1
  // one step: plane k+1 lands (slot SC+1), plane k+2 is requested (slot SC+2, the slot plane k-2 left), plane k is computed
  auto one = [&](auto rsl) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value;
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // my DMA share of plane k+1
    __syncthreads();                                      // all of plane k+1 is in LDS; everybody has left step k-1
    FB[2] = fluid_bits(fs);
    if (k + 1 < m.k_hi) dma_plane((SC + 2) % 4, fs ^ 1, k + 2);
    fs ^= 1;
    step(rsl, k, FB[0], FB[1], FB[2]);
    FB[0] = FB[1]; FB[1] = FB[2];
  };
  while (true) {
    one(AIC<1>{}); if (++k >= m.k_hi) break;
    one(AIC<2>{}); if (++k >= m.k_hi) break;
    one(AIC<3>{}); if (++k >= m.k_hi) break;
    one(AIC<0>{}); if (++k >= m.k_hi) break;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // no LDS-DMA may outlive the wave
}

// ---------------------------------------------------------------------------------------------------
// Backward pass, density: sl_scalar_bwd_clamp_cell<true, false, SO> for the planes [K0, K0+KN).
// Ring fields: rho_fwd, Ux, Uy, Uz.  rho, the traced cell and its clamp bounds are per-cell global loads.
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256, 3) void advect3d_bwd_scalar_tile_kernel(GridDims g, float dt, float half_s,
                                                                          const float* __restrict__ rho,
                                                                          const float* __restrict__ rho_fwd,
                                                                          const int* __restrict__ cell_in,
                                                                          const float* __restrict__ U,
                                                                          const float* __restrict__ flags,
                                                                          const float2* __restrict__ box,
                                                                          float* __restrict__ rho_dst, int ntx, int nty, int zchunk) {
  constexpr int NF = 4;
  constexpr int FSZ = ATRR * ATP;
  __shared__ __attribute__((aligned(16))) float ring[4][NF][FSZ];
  __shared__ __attribute__((aligned(16))) float fstage[2][FSZ];
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  const ABuf rs_r = atile_rsrc(m, g, rho_fwd + sb1, after1), rs_x = atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW),
             rs_y = atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW), rs_z = atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3),
             rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w;
  auto dma_plane = [&](int slot, int fs, int k) {
    const unsigned pb = m.planeoff(g, k);
    atile_dma(m, rs_r, ring[slot][0], pb, 0);
    atile_dma(m, rs_x, ring[slot][1], pb, 1);
    atile_dma(m, rs_y, ring[slot][2], pb, 2);
    atile_dma(m, rs_z, ring[slot][3], pb, 3);
    atile_dma(m, rs_f, fstage[fs], pb, 0);
  };
  const int hr0 = 2 * w + 1;
  const int col = lane + m.xs;
  auto fluid_bits = [&](int fs) {
    unsigned c = 0;
    const float* f = &fstage[fs][(hr0 - 1) * ATP + col - 1];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) c |= (f[rr * ATP + dx] == FNX_FLUID ? 1u : 0u) << (3 * rr + dx);
    }
    return c;
  };
  const int i = m.x;
  const bool xin = i < g.W;
  unsigned FB[3];

  auto step = [&](auto rsl, int k, unsigned fbm, unsigned fbc, unsigned fbp) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value, SM = (SC + 3) % 4, SP = (SC + 1) % 4;
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    bool slow[2];
    bool any_slow = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = m.j0 + 2 * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float ctrx = (float)i + 0.5f, ctry = (float)j + 0.5f;
      auto at = [&](int f, int s, int dy, int dx) { return ring[s][f][(hr0 + r + dy) * ATP + col + dx]; };
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      const unsigned nb = ((fbm >> (3 * r)) & 0x1ffu) | (((fbc >> (3 * r)) & 0x1ffu) << 9) | (((fbp >> (3 * r)) & 0x1ffu) << 18);
      // per-cell global operands, issued first (clamped addresses for the lanes that store nothing)
      const size_t o = (size_t)k * g.HW + (size_t)(j < g.H ? j : g.H - 1) * g.W + (xin ? i : g.W - 1);
      const float src = rho[sb1 + o];
      const int cell = cell_in[sb1 + o];
      const bool inslab = (cell >= g.HW) & (cell < g.HW + g.DHW);
      const float2 bb = box[sb1 + (size_t)(inslab ? cell - g.HW : 0)];

      const float f = at(0, SC, 0, 0);
      const float x_c = at(1, SC, 0, 0), y_c = at(2, SC, 0, 0), z_c = at(3, SC, 0, 0);
      const float cen0 = 0.5f * (x_c + at(1, SC, 0, 1));
      const float cen1 = 0.5f * (y_c + at(2, SC, 1, 0));
      const float cen2 = 0.5f * (z_c + at(3, SP, 0, 0));
      const float d0 = dt * cen0, d1 = dt * cen1, d2 = dt * cen2;           // (-ndt) * cen with ndt = -dt
      const float length = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
      const bool stay = (length <= FNX_EPSILON) | (0.f >= length - FNX_HIT_MARGIN);
      const float dir0 = d0 / length, dir1 = d1 / length, dir2 = d2 / length;
      const float stp = fminf(length - 0.f, 1.f);
      const float n0 = ctrx + dir0 * stp, n1 = ctry + dir1 * stp, n2 = ctrz + dir2 * stp;
      const bool ends = stp >= length - FNX_HIT_MARGIN;
      const int c0 = (int)n0 - i, c1 = (int)n1 - j, c2 = (int)n2 - kg;
      const bool near = ((unsigned)(c0 + 1) <= 2u) & ((unsigned)(c1 + 1) <= 2u) & ((unsigned)(c2 + 1) <= 2u);
      const bool tfluid = (nb >> ((9 * (c2 + 1) + 3 * (c1 + 1) + (c0 + 1)) & 31)) & 1u;
      const bool moved_ok = ends & near & tfluid;
      const float p0 = stay ? ctrx : n0, p1 = stay ? ctry : n1, p2 = stay ? ctrz : n2;
      const ALerp Ls = alerp(p0, p1, p2, i, j, kg);
      float cs[8];
      {
        const int rowcol = (hr0 + r - (Ls.ny ? 1 : 0)) * ATP + col - (Ls.nx ? 1 : 0);
        const float* z0 = (Ls.nz ? &ring[SM][0][0] : &ring[SC][0][0]) + rowcol;
        const float* z1 = (Ls.nz ? &ring[SC][0][0] : &ring[SP][0][0]) + rowcol;
        cs[0] = z0[0]; cs[1] = z0[1]; cs[2] = z0[ATP]; cs[3] = z0[ATP + 1];
        cs[4] = z1[0]; cs[5] = z1[1]; cs[6] = z1[ATP]; cs[7] = z1[ATP + 1];
      }
      float smp;
      if (SAMPLE_OUTSIDE) {
        smp = atrilin(cs, Ls);
      } else {
        const unsigned sh = (Ls.nz ? 0u : 9u) + (Ls.ny ? 0u : 3u) + (Ls.nx ? 0u : 1u);
        const unsigned q = nb >> sh;
        const unsigned fb = (q & 1u) | ((q >> 1) & 1u) << 1 | ((q >> 3) & 1u) << 2 | ((q >> 4) & 1u) << 3 | ((q >> 9) & 1u) << 4 |
                            ((q >> 10) & 1u) << 5 | ((q >> 12) & 1u) << 6 | ((q >> 13) & 1u) << 7;
        smp = atrilin_fluid(cs, fb, Ls);
      }
      const float bwd = border ? 0.f : (fluid ? smp : f);
      float d = f;
      if (fluid) d = f + half_s * (src - bwd);             // applied on border cells too (reference :371)
      const float mn = bb.x, mx = bb.y;
      const bool any = !(mn != mn);
      const float dc = any ? fmaxf(mn, fminf(mx, d)) : f;
      if (!border) d = dc;
      slow[r] = live & !border & ((fluid & (!(stay | moved_ok) | !Ls.ok)) | !inslab);
      if (live) rho_dst[sb1 + o] = d;
      any_slow |= slow[r];
    }
    if (__builtin_amdgcn_ballot_w64(any_slow) != 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        CellId c; c.b = m.b; c.k = k; c.j = m.j0 + 2 * w + r; c.i = i; c.valid = true;
        if (slow[r]) sl_scalar_bwd_clamp_cell<true, false, SAMPLE_OUTSIDE>(g, c, dt, half_s, rho, rho_fwd, cell_in, U, flags, box, rho_dst);
      }
    }
  };

  int k = m.k_lo;
  dma_plane(0, 0, k - 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FB[0] = fluid_bits(0);
  dma_plane(1, 1, k);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FB[1] = fluid_bits(1);
  dma_plane(2, 0, k + 1);
  int fs = 0;
  auto one = [&](auto rsl) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    FB[2] = fluid_bits(fs);
    if (k + 1 < m.k_hi) dma_plane((SC + 2) % 4, fs ^ 1, k + 2);
    fs ^= 1;
    step(rsl, k, FB[0], FB[1], FB[2]);
    FB[0] = FB[1]; FB[1] = FB[2];
  };
  while (true) {
    one(AIC<1>{}); if (++k >= m.k_hi) break;
    one(AIC<2>{}); if (++k >= m.k_hi) break;
    one(AIC<3>{}); if (++k >= m.k_hi) break;
    one(AIC<0>{}); if (++k >= m.k_hi) break;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
}

// ---------------------------------------------------------------------------------------------------
// Backward pass, velocity: sl_mac_bwd_clamp_cell_flat<true> (self-advection: orig == U) for the planes [K0, K0+KN).
// Ring fields: U_fwd x,y,z (sampled), U x,y,z (face velocities, clamp boxes).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void advect3d_bwd_vel_tile_kernel(GridDims g, float dt, float half_s,
                                                                       const float* __restrict__ U,
                                                                       const float* __restrict__ U_fwd,
                                                                       const float* __restrict__ flags,
                                                                       float* __restrict__ U_dst, int ntx, int nty, int zchunk) {
  constexpr int NF = 6;
  constexpr int FSZ = ATRR * ATP;
  __shared__ __attribute__((aligned(16))) float ring[4][NF][FSZ];
  __shared__ __attribute__((aligned(16))) float fstage[2][FSZ];
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  ABuf rs[NF];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rs[a] = atile_rsrc(m, g, U_fwd + sb3 + (size_t)a * g.DHW, after3 + (size_t)(2 - a) * g.DHW);
    rs[3 + a] = atile_rsrc(m, g, U + sb3 + (size_t)a * g.DHW, after3 + (size_t)(2 - a) * g.DHW);
  }
  const ABuf rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w;
  auto dma_plane = [&](int slot, int fs, int k) {
    const unsigned pb = m.planeoff(g, k);
#pragma unroll
    for (int f = 0; f < NF; ++f) atile_dma(m, rs[f], ring[slot][f], pb, f);
    atile_dma(m, rs_f, fstage[fs], pb, 2);
  };
  const int hr0 = 2 * w + 1;
  const int col = lane + m.xs;
  auto fluid_bits = [&](int fs) {
    unsigned c = 0;
    const float* f = &fstage[fs][(hr0 - 1) * ATP + col - 1];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) c |= (f[rr * ATP + dx] == FNX_FLUID ? 1u : 0u) << (3 * rr + dx);
    }
    return c;
  };
  const int i = m.x;
  const bool xin = i < g.W;
  unsigned FB[3];

  auto step = [&](auto rsl, int k, unsigned fbm, unsigned fbc) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value, SM = (SC + 3) % 4, SP = (SC + 1) % 4;
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f, posz = (float)kg;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    bool slow[2];
    bool any_slow = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = m.j0 + 2 * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float ctrx = (float)i + 0.5f, ctry = (float)j + 0.5f, posx = (float)i, posy = (float)j;
      auto at = [&](int f, int s, int dy, int dx) { return ring[s][f][(hr0 + r + dy) * ATP + col + dx]; };
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      // flags of the -1 neighbours along x, y, z (chk[a] is true for every non-border cell)
      const bool fmx = (fbc >> (3 * (r + 1) + 0)) & 1u, fmy = (fbc >> (3 * r + 1)) & 1u, fmz = (fbm >> (3 * (r + 1) + 1)) & 1u;
      // get_at_mac<true, false, 0/1/2> on U (ring fields 3, 4, 5), operand for operand
      const float x_c = at(3, SC, 0, 0), y_c = at(4, SC, 0, 0), z_c = at(5, SC, 0, 0);
      const float x_r = at(3, SC, 0, 1), y_u = at(4, SC, 1, 0), z_f = at(5, SP, 0, 0);
      float v[3][3];
      v[0][0] = x_c;
      v[0][1] = 0.25f * (((y_c + at(4, SC, 0, -1)) + y_u) + at(4, SC, 1, -1));
      v[0][2] = 0.25f * (((z_c + at(5, SC, 0, -1)) + z_f) + at(5, SP, 0, -1));
      v[1][0] = 0.25f * (((x_c + at(3, SC, -1, 0)) + x_r) + at(3, SC, -1, 1));
      v[1][1] = y_c;
      v[1][2] = 0.25f * (((z_c + at(5, SC, -1, 0)) + z_f) + at(5, SP, -1, 0));
      v[2][0] = 0.25f * (((x_c + at(3, SM, 0, 0)) + x_r) + at(3, SM, 0, 1));
      v[2][1] = 0.25f * (((y_c + at(4, SM, 0, 0)) + y_u) + at(4, SM, 1, 0));
      v[2][2] = z_c;
      const float fwd0 = at(0, SC, 0, 0), fwd1 = at(1, SC, 0, 0), fwd2 = at(2, SC, 0, 0);
      float out[3];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float vd0 = v[a][0] * dt, vd1 = v[a][1] * dt, vd2 = v[a][2] * dt;
        const ALerp L = alerp(ctrx + vd0, ctry + vd1, ctrz + vd2, i, j, kg);
        float c[8];
        {
          const int rowcol = (hr0 + r - (L.ny ? 1 : 0)) * ATP + col - (L.nx ? 1 : 0);
          const float* z0 = (L.nz ? &ring[SM][a][0] : &ring[SC][a][0]) + rowcol;
          const float* z1 = (L.nz ? &ring[SC][a][0] : &ring[SP][a][0]) + rowcol;
          c[0] = z0[0]; c[1] = z0[1]; c[2] = z0[ATP]; c[3] = z0[ATP + 1];
          c[4] = z1[0]; c[5] = z1[1]; c[6] = z1[ATP]; c[7] = z1[ATP + 1];
        }
        const float smp = atrilin(c, L);
        float mn = INFINITY, mx = -INFINITY;
        bool okc = true;
#pragma unroll
        for (int l = 0; l < 2; ++l) {                      // doClampComponentMAC: the boxes at trunc(pos -/+ vd)
          const int qx = (int)(l == 0 ? posx - vd0 : posx + vd0);
          const int qy = (int)(l == 0 ? posy - vd1 : posy + vd1);
          const int qz = (int)(l == 0 ? posz - vd2 : posz + vd2);
          const int rx = qx - i, ry = qy - j, rz = qz - kg;
          okc &= ((unsigned)(rx + 1) <= 1u) & ((unsigned)(ry + 1) <= 1u) & ((unsigned)(rz + 1) <= 1u);
          const int rowcol = (hr0 + r + (ry == -1 ? -1 : 0)) * ATP + col + (rx == -1 ? -1 : 0);
          const float* z0 = (rz == -1 ? &ring[SM][3 + a][0] : &ring[SC][3 + a][0]) + rowcol;
          const float* z1 = (rz == -1 ? &ring[SC][3 + a][0] : &ring[SP][3 + a][0]) + rowcol;
          const float e0 = z0[0], e1 = z0[1], e2 = z0[ATP], e3 = z0[ATP + 1], e4 = z1[0], e5 = z1[1], e6 = z1[ATP], e7 = z1[ATP + 1];
          mn = fminf(mn, e0); mx = fmaxf(mx, e0); mn = fminf(mn, e1); mx = fmaxf(mx, e1);
          mn = fminf(mn, e2); mx = fmaxf(mx, e2); mn = fminf(mn, e3); mx = fmaxf(mx, e3);
          mn = fminf(mn, e4); mx = fmaxf(mx, e4); mn = fminf(mn, e5); mx = fmaxf(mx, e5);
          mn = fminf(mn, e6); mx = fmaxf(mx, e6); mn = fminf(mn, e7); mx = fmaxf(mx, e7);
        }
        const float fa = a == 0 ? fwd0 : (a == 1 ? fwd1 : fwd2);
        const float og = a == 0 ? x_c : (a == 1 ? y_c : z_c);
        const float bwd = fluid ? smp : (a == 0 ? fwd1 : (a == 1 ? 0.f : fa));     // Q1 pass-through of SL(fwd)
        const bool fm = a == 0 ? fmx : (a == 1 ? fmy : fmz);
        const bool skip = !fluid | !fm;
        const float corr = skip ? fa : fa + half_s * (og - bwd);
        out[a] = fmaxf(fminf(corr, mx), mn);
        ok &= okc & (L.ok | !fluid);
      }
      if (border) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; }
      slow[r] = live & !border & !ok;
      if (live) {
        const size_t o = (size_t)k * g.HW + (size_t)j * g.W + i;
        U_dst[sb3 + o] = out[0];
        U_dst[sb3 + g.DHW + o] = out[1];
        U_dst[sb3 + 2 * (size_t)g.DHW + o] = out[2];
      }
      any_slow |= slow[r];
    }
    if (__builtin_amdgcn_ballot_w64(any_slow) != 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        CellId c; c.b = m.b; c.k = k; c.j = m.j0 + 2 * w + r; c.i = i; c.valid = true;
        if (slow[r]) sl_mac_bwd_clamp_cell_flat<true>(g, c, dt, half_s, U, U_fwd, U, flags, U_dst);
      }
    }
  };

  int k = m.k_lo;
  dma_plane(0, 0, k - 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FB[0] = fluid_bits(0);
  dma_plane(1, 1, k);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FB[1] = fluid_bits(1);
  dma_plane(2, 0, k + 1);
  int fs = 0;
  auto one = [&](auto rsl) __attribute__((always_inline)) {
    constexpr int SC = decltype(rsl)::value;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    FB[2] = fluid_bits(fs);
    if (k + 1 < m.k_hi) dma_plane((SC + 2) % 4, fs ^ 1, k + 2);
    fs ^= 1;
    step(rsl, k, FB[0], FB[1]);
    FB[0] = FB[1]; FB[1] = FB[2];
  };
  while (true) {
    one(AIC<1>{}); if (++k >= m.k_hi) break;
    one(AIC<2>{}); if (++k >= m.k_hi) break;
    one(AIC<3>{}); if (++k >= m.k_hi) break;
    one(AIC<0>{}); if (++k >= m.k_hi) break;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
}
