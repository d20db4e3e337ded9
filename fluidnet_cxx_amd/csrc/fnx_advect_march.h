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
// A lane whose displacement is not below one cell, or whose trace ends in a non-fluid cell, is not computed here: the
// kernel records it in a bitmap (one 64-bit word per 64-cell row segment, written by every wave) and a fix-up launch of
// the per-cell function above redoes exactly those cells from memory (one thread per word: a few us when nothing is
// flagged).  The result is bit-identical to the per-cell kernels for EVERY input; the tile path only has
// to be the common case.  Every expression below is the per-cell function's own, operand for operand (fnx_device.h);
// only where the operands come from differs.  (Inlining the fallback into the tile kernels cost 40-280 bytes of scratch
// per lane and a third of the instruction cache.)
// (A first version kept the planes in registers and picked the corners with v_cndmask networks -- 38 selects + 18 DPP
// moves per sample: 1540 VALU per two rows, slower than the gather kernel it replaced; the LDS does that selection for
// the price of an address.)

constexpr int ATR = 8;               // tile rows
#ifndef FNX_ATRPW
#define FNX_ATRPW 2
#endif
constexpr int ATRPW = FNX_ATRPW;     // rows per wave (1: 8 waves per workgroup, 2: 4 waves)
constexpr int ATNW = ATR / ATRPW;    // waves per workgroup
// waves per SIMD the register budgets are sized for (workgroups per CU x ATNW / 4)
#ifndef FNX_AT_WPS_FWD
#define FNX_AT_WPS_FWD (ATRPW == 1 ? 4 : 3)
#define FNX_AT_WPS_BS (ATRPW == 1 ? 6 : 3)
#define FNX_AT_WPS_BV (ATRPW == 1 ? 4 : 2)
#endif
constexpr int ATRR = ATR + 2;        // rows held: j0-1 .. j0+8
constexpr int ATP = 68;              // row pitch in floats: columns x0-1 .. x0+66 = 17 chunks of 16 bytes
constexpr int ATNQ = (ATRR + 2) / 3; // DMA instructions per field-plane (3 rows each): 4
constexpr int AFSZ_ = ATRR * ATP;    // floats per field-plane

typedef __amdgpu_buffer_rsrc_t ABuf;
typedef __attribute__((address_space(3))) void* ALds;
__device__ __forceinline__ ABuf amake_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
template <int N> struct AIC { static constexpr int value = N; };

// lerp_setup (fnx_device.h) for a position whose base cell lands on {c-1, c} per axis, c = (i, j, kg) given as floats.
// With p' = p - 0.5 >= 0:  s1 = p' - (float)(int)p' is exactly fract(p') (the subtraction is exact, and so is v_fract for a
// non-negative argument), s0 = 1 - s1, and lerp_setup's clamp01 returns values in [0, 1] unchanged, bit for bit; the base is
// c-1 iff p' < c, and it is in {c-1, c} iff c-1 <= p' < c+1 (for a non-border cell c >= 1, so p' >= 0 follows and
// lerp_setup's index clamps are no-ops too).  ok = all three hold: the only case the tile path covers.
struct ALerp { float s0, s1, t0, t1, f0, f1; bool nx, ny, nz, ok; };
__device__ __forceinline__ ALerp alerp(float px, float py, float pz, float fi, float fj, float fk) {
  ALerp L;
  px = px - 0.5f; py = py - 0.5f; pz = pz - 0.5f;
  L.s1 = __builtin_amdgcn_fractf(px); L.t1 = __builtin_amdgcn_fractf(py); L.f1 = __builtin_amdgcn_fractf(pz);
  L.s0 = 1.f - L.s1; L.t0 = 1.f - L.t1; L.f0 = 1.f - L.f1;
  L.nx = px < fi; L.ny = py < fj; L.nz = pz < fk;
  L.ok = (px >= fi - 1.f) & (px < fi + 1.f) & (py >= fj - 1.f) & (py < fj + 1.f) & (pz >= fk - 1.f) & (pz < fk + 1.f);
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
  int q, fsel;                   // this wave's DMA instruction of a field-plane, and which field-planes it serves
  unsigned voff;                 // per-lane byte offset of "my" row + chunk inside a plane for this wave's DMA instruction
  bool dma_lane;
  unsigned hw;
  int bx, ntx;
  // fix-up bitmap word of my 64-cell row segment in plane k, row j
  __device__ __forceinline__ size_t word(const GridDims& g, int k, int j) const { return (((size_t)b * g.D + k) * g.H + j) * ntx + bx; }
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
  m.bx = bx; m.ntx = ntx;
  m.j0 = by * ATR;
  m.k_lo = g.K0 + zc * zchunk;
  m.k_hi = min(m.k_lo + zchunk, g.K0 + g.KN);
  m.hw = (unsigned)g.HW;
  m.k0 = m.k_lo - 1 < 0 ? 0 : m.k_lo - 1;
  // DMA instruction q of a field-plane moves held rows 3q .. 3q+2: lane l fetches the 16-byte chunk l%17 of row 3q + l/17
  // (51 lanes).  Wave w issues instruction q = w of EVERY field-plane (ATNQ == ATNW): one per-lane offset, one lane mask and a
  // scalar LDS row offset per wave, no per-instruction dispatch (dealing the instructions round-robin made every wave walk
  // all 20 slots of a plane behind per-lane compares: ~180 instructions of a ~1100-instruction step).
  // Rows are clamped into the grid; columns are not (a chunk is 4 columns): right of column W-1 it reads the next row's
  // cells (or 0 past the end of the tensor: the buffer range check) -- only border cells ever see those.  On the LEFT no
  // chunk may start before column 0: a negative offset fails the range check for the whole 16 bytes, columns 0..2
  // included, so the tiles of the first tile column hold columns 0 .. 67 (no left halo: column 0 is a border column and
  // never looks left) and every other tile holds x0-1 .. x0+66.
  m.xs = bx == 0 ? 0 : 1;
  static_assert(ATNW % ATNQ == 0, "whole sets of DMA instructions per workgroup");
  m.q = m.w % ATNQ; m.fsel = m.w / ATNQ;                 // (8 waves: waves 0-3 move the even field-planes, 4-7 the odd ones)
  const int rsub = m.lane / 17, cq = m.lane - rsub * 17;
  const int hr = 3 * m.q + rsub;
  int jr = m.j0 - 1 + hr;
  jr = jr < 0 ? 0 : (jr > g.H - 1 ? g.H - 1 : jr);
  m.voff = (unsigned)(jr * g.W + (bx * 64 - m.xs) + 4 * cq) * 4u;
  m.dma_lane = (m.lane < 51) & (hr < ATRR);
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

// this wave's three rows of a field-plane (ATRR rows) -> LDS at `dst` ([ATRR][ATP] floats); call under `if (m.dma_lane)`.
// The array's address is taken in the LDS address space before the (scalar) row offset is added: a cast of the sum would
// carry a null-pointer test per instruction.
typedef __attribute__((address_space(3))) float* ALdsF;
__device__ __forceinline__ void atile_dma(const ATile& m, const ABuf& rs, float (&dst)[AFSZ_], unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (ALds)((ALdsF)&dst[0] + 3 * ATP * m.q), 16, off, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// The march itself, shared by the three kernels.
//
// LDS: four ring slots (planes k-1, k, k+1 and the one in flight) of NF fields, plus two stages for the flags of the
// plane that has just landed / is in flight.  They are SEPARATE __shared__ arrays, selected by compile-time phase: the
// compiler cannot tell an LDS-DMA into one ring slot from the ds_reads of the others when they are one array, and then
// parks an s_waitcnt vmcnt(0) between a step's DMA issue and its first read -- the whole transfer latency, every step.
//
// Synchronisation per step (one plane): s_barrier (everybody has left step k-1 and everybody's share of plane k+1 has
// landed) -> fluid bits of plane k+1 -> issue the DMA of plane k+2 into the slot plane k-2 left -> compute plane k from
// LDS into registers -> s_waitcnt vmcnt(0): my share of plane k+2 (issued a whole step's compute ago) -> store.  The row
// stores of a step are thus never waited for before the NEXT step's wait, one step's compute later (a plain
// __syncthreads() would fence them right away: vmcnt counts stores too).
// ---------------------------------------------------------------------------------------------------
constexpr int AFSZ = AFSZ_;                              // floats per field-plane

#define ATILE_LDS(NF)                                                          \
  __shared__ __attribute__((aligned(16))) float ring0[NF][AFSZ];               \
  __shared__ __attribute__((aligned(16))) float ring1[NF][AFSZ];               \
  __shared__ __attribute__((aligned(16))) float ring2[NF][AFSZ];               \
  __shared__ __attribute__((aligned(16))) float ring3[NF][AFSZ];               \
  __shared__ __attribute__((aligned(16))) float fst0[AFSZ];                    \
  __shared__ __attribute__((aligned(16))) float fst1[AFSZ]

__device__ __forceinline__ void atile_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void atile_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// fluid bits of a plane for the rows hr0-1 .. hr0+ATRPW around my rows: bit 3*rr + (dx+1)
__device__ __forceinline__ unsigned atile_fluid_bits(const float* fst, int hr0, int col) {
  unsigned c = 0;
  const float* f = fst + (hr0 - 1) * ATP + col - 1;
#pragma unroll
  for (int rr = 0; rr < ATRPW + 2; ++rr) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) c |= (f[rr * ATP + dx] == FNX_FLUID ? 1u : 0u) << (3 * rr + dx);
  }
  return c;
}

// body(k, fbm, fbc, fbp, ringM, ringC, ringP, pre_store): compute plane k from the three ring slots, call pre_store(), store.
template <int NF, class Body>
__device__ __forceinline__ void atile_march(const GridDims& g, const ATile& m, const ABuf (&rs)[NF], const ABuf& rs_f,
                                            float (&ring0)[NF][AFSZ], float (&ring1)[NF][AFSZ], float (&ring2)[NF][AFSZ],
                                            float (&ring3)[NF][AFSZ], float (&fst0)[AFSZ], float (&fst1)[AFSZ], Body body) {
  const int hr0 = ATRPW * m.w + 1, col = m.lane + m.xs;
  auto dma_plane = [&](float (&ring)[NF][AFSZ], float (&fst)[AFSZ], int k) {
    const unsigned off = m.voff + m.planeoff(g, k);
    constexpr int NSET = ATNW / ATNQ;
    if (m.dma_lane) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (NSET == 1 || f % NSET == m.fsel) atile_dma(m, rs[f], ring[f], off);
      if (NSET == 1 || NF % NSET == m.fsel) atile_dma(m, rs_f, fst, off);
    }
  };
  unsigned FB0, FB1;
  int k = m.k_lo;
  // prologue: ring slot of plane p is (p - k_lo + 1) mod 4, its flags stage (p - k_lo + 1) mod 2
  dma_plane(ring0, fst0, k - 1);
  dma_plane(ring1, fst1, k);
  atile_wait_vm();
  atile_barrier();
  FB0 = atile_fluid_bits(fst0, hr0, col);
  FB1 = atile_fluid_bits(fst1, hr0, col);
  atile_barrier();                                        // everybody has read fst0 before plane k+1's flags land in it
  dma_plane(ring2, fst0, k + 1);
  atile_wait_vm();
  auto pre_store = [&]() { atile_wait_vm(); };
#define ATILE_STEP(RM, RC, RP, RN, FP, FN)                                             \
  {                                                                                     \
    atile_barrier();                                                                    \
    const unsigned FB2 = atile_fluid_bits(FP, hr0, col);                                \
    if (k + 1 < m.k_hi) dma_plane(RN, FN, k + 2);                                       \
    body(k, FB0, FB1, FB2, RM, RC, RP, pre_store);                                      \
    FB0 = FB1; FB1 = FB2;                                                               \
    if (++k >= m.k_hi) break;                                                           \
  }
  while (true) {
    ATILE_STEP(ring0, ring1, ring2, ring3, fst0, fst1)
    ATILE_STEP(ring1, ring2, ring3, ring0, fst1, fst0)
    ATILE_STEP(ring2, ring3, ring0, ring1, fst0, fst1)
    ATILE_STEP(ring3, ring0, ring1, ring2, fst1, fst0)
  }
#undef ATILE_STEP
  atile_wait_vm();                                        // no LDS-DMA may outlive the wave
}

// the 8 corners of a sample of ring field f at the per-lane base cell (b = -1 where n? is set)
template <int NF>
__device__ __forceinline__ void atile_corners(const float (&rM)[NF][AFSZ], const float (&rC)[NF][AFSZ], const float (&rP)[NF][AFSZ],
                                              int f, int rowcol0, bool nx, bool ny, bool nz, float (&c)[8]) {
  const int rowcol = rowcol0 - (ny ? ATP : 0) - (nx ? 1 : 0);
  const float* z0 = (nz ? &rM[f][0] : &rC[f][0]) + rowcol;
  const float* z1 = (nz ? &rC[f][0] : &rP[f][0]) + rowcol;
  c[0] = z0[0]; c[1] = z0[1]; c[2] = z0[ATP]; c[3] = z0[ATP + 1];
  c[4] = z1[0]; c[5] = z1[1]; c[6] = z1[ATP]; c[7] = z1[ATP + 1];
}

// fluid bits of the 8 corners out of the 27 neighbourhood bits: base bit 9*(bz+1) + 3*(by+1) + (bx+1), offsets 0,1,3,4,9,10,12,13
__device__ __forceinline__ unsigned atile_corner_bits(unsigned nb, const ALerp& L) {
  const unsigned sh = (L.nz ? 0u : 9u) + (L.ny ? 0u : 3u) + (L.nx ? 0u : 1u);
  const unsigned q = nb >> sh;
  return (q & 1u) | ((q >> 1) & 1u) << 1 | ((q >> 3) & 1u) << 2 | ((q >> 4) & 1u) << 3 | ((q >> 9) & 1u) << 4 |
         ((q >> 10) & 1u) << 5 | ((q >> 12) & 1u) << 6 | ((q >> 13) & 1u) << 7;
}

// The three quotients d_i / length of a trace with ONE reciprocal (round 6).  hipcc expands an IEEE fp32 division into v_div_scale x 2,
// v_rcp, two refinements of the reciprocal, the quotient with two residual corrections, v_div_fmas, v_div_fixup (11 instructions);
// whenever v_div_scale leaves its operands alone that sequence is plain arithmetic -- r = fma(fma(-b, rcp b, 1), rcp b, rcp b);
// q = a r; q = fma(fma(-b, q, a), r, q); q = fma(fma(-b, q, a), r, q) -- in which everything up to r depends on the denominator only.
// Here: 3 + 3 x 5 instructions instead of 33, the same bits wherever they are looked at:
//   * the quotients are only ever used as ctr + dir * stp (atile_trace), by lanes whose length is > FNX_HIT_MARGIN;
//   * v_div_scale rescales when the denominator is denormal or >= 2^126 (a wave that holds a length above 2^60 or a NaN takes the
//     plain divisions: wave-uniform branch), or when the numerator is below 2^-103 / the quotient denormal: then |dir * stp| < 2^-80
//     against ctr >= 0.5 and the sum is ctr whatever the quotient's last bits (or the sign of a zero quotient, which v_div_fixup
//     would take from the operands);
//   * lanes with length <= FNX_HIT_MARGIN (0 included: rcp gives inf, everything NaN) discard the quotients, as they discard 0 / 0 today.
// tests/test_parity_gpu.py::test_tile_trace_division_extremes runs tiles against the per-cell kernels (plain divisions) on velocity
// fields of mixed magnitudes (1, 1e-20, denormals, +-0, 1e25).
__device__ __forceinline__ void adiv3(float a0, float a1, float a2, float b, float& q0, float& q1, float& q2) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(b <= 0x1p60f)) != 0, 0)) { q0 = a0 / b; q1 = a1 / b; q2 = a2 / b; return; }
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float r = fmaf(fmaf(-b, r0, 1.f), r0, r0);
  auto quot = [&](float a) __attribute__((always_inline)) {
    float q = a * r;
    q = fmaf(fmaf(-b, q, a), r, q);
    return fmaf(fmaf(-b, q, a), r, q);
  };
  q0 = quot(a0); q1 = quot(a1); q2 = quot(a2);
}

// line_trace from the centre of a non-border FLUID cell with displacement d (fnx_device.h): either it stays (length <= eps
// / <= margin) or it is ONE step of length min(|d|, 1) that must end the loop and land in a fluid cell of the
// neighbourhood (bits nb).  Returns the end point and whether this path covers the trace.
__device__ __forceinline__ bool atile_trace(float d0, float d1, float d2, float ctrx, float ctry, float ctrz, int i, int j, int kg,
                                            unsigned nb, float& p0, float& p1, float& p2) {
  const float length = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
  const bool stay = (length <= FNX_EPSILON) | (0.f >= length - FNX_HIT_MARGIN);
  float dir0, dir1, dir2;
  adiv3(d0, d1, d2, length, dir0, dir1, dir2);
  const float stp = fminf(length - 0.f, 1.f);
  const float n0 = ctrx + dir0 * stp, n1 = ctry + dir1 * stp, n2 = ctrz + dir2 * stp;
  const bool ends = stp >= length - FNX_HIT_MARGIN;                            // the second iteration's exit test
  const int c0 = (int)n0 - i, c1 = (int)n1 - j, c2 = (int)n2 - kg;              // traced cell relative to this one
  const bool near = ((unsigned)(c0 + 1) <= 2u) & ((unsigned)(c1 + 1) <= 2u) & ((unsigned)(c2 + 1) <= 2u);
  const bool tfluid = (nb >> ((9 * (c2 + 1) + 3 * (c1 + 1) + (c0 + 1)) & 31)) & 1u;
  p0 = stay ? ctrx : n0; p1 = stay ? ctry : n1; p2 = stay ? ctrz : n2;
  return stay | (ends & near & tfluid);
}

// ---------------------------------------------------------------------------------------------------
// Forward pass: sl_scalar_cell (density) + sl_mac_cell_flat (velocity) for the planes [K0, K0+KN).
// Ring fields: rho, Ux, Uy, Uz.  WHAT: bit 0 = the density part, bit 1 = the velocity part (3: both advections of a step in one
// march; 1 / 2: the stand-alone advectScalar / advectVel entry points -- the velocity-only march holds no rho ring).
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE, int WHAT>
__global__ __launch_bounds__(64 * ATNW, FNX_AT_WPS_FWD) void advect3d_fwd_tile_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                                   const float* __restrict__ U,
                                                                   const float* __restrict__ flags,
                                                                   float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                                   float* __restrict__ U_fwd,
                                                                   float2* __restrict__ box,
                                                                   unsigned long long* __restrict__ fix_s,
                                                                   unsigned long long* __restrict__ fix_v, int ntx, int nty,
                                                                   int zchunk) {
  constexpr bool DO_S = (WHAT & 1) != 0, DO_V = (WHAT & 2) != 0;
  constexpr int UO = DO_S ? 1 : 0;                       // ring field of Ux
  constexpr int NF = 3 + UO;
  ATILE_LDS(NF);
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  ABuf rs[NF];
  if constexpr (DO_S) rs[0] = atile_rsrc(m, g, rho + sb1, after1);
  rs[UO] = atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW);
  rs[UO + 1] = atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW);
  rs[UO + 2] = atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3);
  const ABuf rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w, i = m.x, hr0 = ATRPW * w + 1, col = lane + m.xs;
  const bool xin = i < g.W;
  const float ndt = -dt;
  // Clamp bounds of the density's MacCormack step (box_minmax_kernel's field, fnx_advect.hip) from the rho planes this
  // kernel streams anyway: per plane the 3x3 reduction of my rows (cells that are fluid -- unless SAMPLE_OUTSIDE -- and inside
  // the grid; a clamped halo row / plane is a duplicate of a box member, which min and max do not see), kept for the planes
  // k-1 and k, combined with plane k+1's.  min / max are exact and order-free: the bits of the separate pass.
  const unsigned xmask = ((i >= 1 ? 1u : 0u) | (xin ? 2u : 0u) | (i + 1 < g.W ? 4u : 0u)) * 0x249u;      // 4 rows x (dx = -1, 0, +1)
  // A cell that is not a member enters as a NaN (all ones), which v_min / v_max skip; whether a box has a member at all is
  // kept as bits (a member whose value is itself a NaN must still count: the end of the body restores the separate pass's
  // +inf / -inf for a box whose members are all NaN).
  float bxmn[2][ATRPW], bxmx[2][ATRPW]; unsigned bxan[2];
  auto box_plane = [&](const float (&rX)[NF][AFSZ], unsigned fb, float (&omn)[ATRPW], float (&omx)[ATRPW], unsigned& oan) __attribute__((always_inline)) {
    const unsigned fbx = (SAMPLE_OUTSIDE ? 0xfffu : fb) & xmask;
    const int nfb = (int)~fbx;
    float rmn[ATRPW + 2], rmx[ATRPW + 2];
    const float* q = &rX[0][(hr0 - 1) * ATP + col - 1];
#pragma unroll
    for (int rr = 0; rr < ATRPW + 2; ++rr) {
      float v[3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        v[dx] = __int_as_float(__float_as_int(q[rr * ATP + dx]) | __builtin_amdgcn_sbfe(nfb, 3 * rr + dx, 1));   // member ? value : NaN
      rmn[rr] = fminf(fminf(v[0], v[1]), v[2]);
      rmx[rr] = fmaxf(fmaxf(v[0], v[1]), v[2]);
    }
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      omn[r] = fminf(fminf(rmn[r], rmn[r + 1]), rmn[r + 2]);
      omx[r] = fmaxf(fmaxf(rmx[r], rmx[r + 1]), rmx[r + 2]);
    }
    oan = fbx | (fbx >> 1) | (fbx >> 2);                                                   // bit 3 rr: a member in row rr
  };

  auto body = [&](int k, unsigned fbm, unsigned fbc, unsigned fbp, const float (&rM)[NF][AFSZ], const float (&rC)[NF][AFSZ],
                  const float (&rP)[NF][AFSZ], auto pre_store) __attribute__((always_inline)) {
    const int kg = k + g.zoff;
    if constexpr (DO_S) {
      if (k == m.k_lo) {                                   // first step of the chunk: planes k-1 and k
        box_plane(rM, fbm, bxmn[0], bxmx[0], bxan[0]);
        box_plane(rC, fbc, bxmn[1], bxmx[1], bxan[1]);
      }
      float bpmn[ATRPW], bpmx[ATRPW]; unsigned bpan;
      box_plane(rP, fbp, bpmn, bpmx, bpan);
      const unsigned anyrows = bxan[0] | bxan[1] | bpan;
#pragma unroll
      for (int r = 0; r < ATRPW; ++r) {
        float lo = fminf(fminf(bxmn[0][r], bxmn[1][r]), bpmn[r]), hi = fmaxf(fmaxf(bxmx[0][r], bxmx[1][r]), bpmx[r]);
        const bool any = ((anyrows >> (3 * r)) & 0x49u) != 0;
        lo = lo != lo ? INFINITY : lo; hi = hi != hi ? -INFINITY : hi;                    // members, but every one a NaN
        // stored right away (nothing here waits for a store before the next step's wait for its plane)
        const int j = m.j0 + ATRPW * w + r;
        if (xin && j < g.H) box[sb1 + (size_t)k * g.HW + (size_t)j * g.W + i] = make_float2(any ? lo : __builtin_nanf(""), hi);
        bxmn[0][r] = bxmn[1][r]; bxmx[0][r] = bxmx[1][r];
        bxmn[1][r] = bpmn[r]; bxmx[1][r] = bpmx[r];
      }
      bxan[0] = bxan[1]; bxan[1] = bpan;
    }
    const float ctrz = (float)kg + 0.5f;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    float o_rho[ATRPW], o_u[ATRPW][3]; int o_cell[ATRPW];
    unsigned long long ws[ATRPW], wv[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const float fi = (float)i, fj = (float)j, fk = (float)kg;
      const float ctrx = fi + 0.5f, ctry = fj + 0.5f;
      const int rc0 = (hr0 + r) * ATP + col;                          // my cell in a field-plane
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      // 27 fluid bits of the neighbourhood: bit 9*(dz+1) + 3*(dy+1) + (dx+1)
      const unsigned nb = ((fbm >> (3 * r)) & 0x1ffu) | (((fbc >> (3 * r)) & 0x1ffu) << 9) | (((fbp >> (3 * r)) & 0x1ffu) << 18);

      const float x_c = rC[UO][rc0], y_c = rC[UO + 1][rc0], z_c = rC[UO + 2][rc0];
      const float x_r = rC[UO][rc0 + 1], y_u = rC[UO + 1][rc0 + ATP], z_f = rP[UO + 2][rc0];
      const bool live = xin & (j < g.H);
      // ================= density: sl_scalar_cell =================
      if constexpr (DO_S) {
        const float cen0 = 0.5f * (x_c + x_r), cen1 = 0.5f * (y_c + y_u), cen2 = 0.5f * (z_c + z_f);   // get_centered
        float p0, p1, p2;
        const bool traced = atile_trace(ndt * cen0, ndt * cen1, ndt * cen2, ctrx, ctry, ctrz, i, j, kg, nb, p0, p1, p2);
        const ALerp Ls = alerp(p0, p1, p2, fi, fj, fk);
        float cs[8];
        atile_corners<NF>(rM, rC, rP, 0, rc0, Ls.nx, Ls.ny, Ls.nz, cs);
        // wave-uniform shortcut: when every lane that uses its sample has an all-fluid neighbourhood, interpol_with_fluid IS the
        // plain trilinear expression (every lerp1d_fluid takes its a*ta + b*tb branch): same bits, a third of the work
        const bool allfluid = SAMPLE_OUTSIDE || __builtin_amdgcn_ballot_w64(!border & fluid & (nb != 0x7ffffffu)) == 0;
        const float smp = allfluid ? atrilin(cs, Ls) : atrilin_fluid(cs, atile_corner_bits(nb, Ls), Ls);
        const float rho_c = rC[0][rc0];
        o_rho[r] = border ? 0.f : (fluid ? smp : rho_c);
        const bool keep = border | !fluid;                 // p = ctr
        const float q0 = keep ? ctrx : p0, q1 = keep ? ctry : p1, q2 = keep ? ctrz : p2;
        const int ci = clampi((int)q0, 0, g.W - 1), cj = clampi((int)q1, 0, g.H - 1);
        const int ck = clampi((int)q2, 0, g.Dglob - 1) - g.zoff;
        o_cell[r] = (ck + 1) * g.HW + cj * g.W + ci;
        ws[r] = __builtin_amdgcn_ballot_w64(live & !keep & (!traced | !Ls.ok));
      }

      // ================= velocity: sl_mac_cell_flat =================
      if constexpr (DO_V) {
        // get_at_mac<true, false, 0/1/2> (fnx_device.h), operand for operand
        float v0[3], v1[3], v2[3];
        v0[0] = x_c;
        v0[1] = 0.25f * (((y_c + rC[UO + 1][rc0 - 1]) + y_u) + rC[UO + 1][rc0 + ATP - 1]);
        v0[2] = 0.25f * (((z_c + rC[UO + 2][rc0 - 1]) + z_f) + rP[UO + 2][rc0 - 1]);
        v1[0] = 0.25f * (((x_c + rC[UO][rc0 - ATP]) + x_r) + rC[UO][rc0 - ATP + 1]);
        v1[1] = y_c;
        v1[2] = 0.25f * (((z_c + rC[UO + 2][rc0 - ATP]) + z_f) + rP[UO + 2][rc0 - ATP]);
        v2[0] = 0.25f * (((x_c + rM[UO][rc0]) + x_r) + rM[UO][rc0 + 1]);
        v2[1] = 0.25f * (((y_c + rM[UO + 1][rc0]) + y_u) + rM[UO + 1][rc0 + ATP]);
        v2[2] = z_c;
        bool okv = true;
        {
          const ALerp L = alerp(ctrx + v0[0] * ndt, ctry + v0[1] * ndt, ctrz + v0[2] * ndt, fi, fj, fk);
          float c[8]; atile_corners<NF>(rM, rC, rP, UO, rc0, L.nx, L.ny, L.nz, c);
          o_u[r][0] = fluid ? atrilin(c, L) : y_c;          // non-fluid cell: channel 1 into channel 0 (:413-416)
          okv &= L.ok;
        }
        {
          const ALerp L = alerp(ctrx + v1[0] * ndt, ctry + v1[1] * ndt, ctrz + v1[2] * ndt, fi, fj, fk);
          float c[8]; atile_corners<NF>(rM, rC, rP, UO + 1, rc0, L.nx, L.ny, L.nz, c);
          o_u[r][1] = fluid ? atrilin(c, L) : 0.f;
          okv &= L.ok;
        }
        {
          const ALerp L = alerp(ctrx + v2[0] * ndt, ctry + v2[1] * ndt, ctrz + v2[2] * ndt, fi, fj, fk);
          float c[8]; atile_corners<NF>(rM, rC, rP, UO + 2, rc0, L.nx, L.ny, L.nz, c);
          o_u[r][2] = fluid ? atrilin(c, L) : z_c;
          okv &= L.ok;
        }
        wv[r] = __builtin_amdgcn_ballot_w64(live & !border & fluid & !okv);
        if (border) { o_u[r][0] = 0.f; o_u[r][1] = 0.f; o_u[r][2] = 0.f; }
      }
    }
    pre_store();
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      if (xin && j < g.H) {
        const size_t o = (size_t)k * g.HW + (size_t)j * g.W + i;
        if constexpr (DO_S) {
          rho_fwd[sb1 + o] = o_rho[r];
          cell_out[sb1 + o] = o_cell[r];
        }
        if constexpr (DO_V) {
          U_fwd[sb3 + o] = o_u[r][0];
          U_fwd[sb3 + g.DHW + o] = o_u[r][1];
          U_fwd[sb3 + 2 * (size_t)g.DHW + o] = o_u[r][2];
        }
      }
      // lanes the neighbourhood path does not cover (|displacement| >= 1 cell, trace into a non-fluid cell) go to the fix-up
      if (lane == 0 && j < g.H) {
        const size_t wi = m.word(g, k, j);
        if constexpr (DO_S) fix_s[wi] = ws[r];
        if constexpr (DO_V) fix_v[wi] = wv[r];
      }
    }
  };
  atile_march<NF>(g, m, rs, rs_f, ring0, ring1, ring2, ring3, fst0, fst1, body);
}

// ---------------------------------------------------------------------------------------------------
// Backward pass, density: sl_scalar_bwd_clamp_cell<true, false, SO> for the planes [K0, K0+KN).
// Ring fields: rho_fwd, Ux, Uy, Uz.  rho, the traced cell and its clamp bounds are per-cell global loads.
// ---------------------------------------------------------------------------------------------------
template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(64 * ATNW, FNX_AT_WPS_BS) void advect3d_bwd_scalar_tile_kernel(GridDims g, float dt, float half_s,
                                                                          const float* __restrict__ rho,
                                                                          const float* __restrict__ rho_fwd,
                                                                          const int* __restrict__ cell_in,
                                                                          const float* __restrict__ U,
                                                                          const float* __restrict__ flags,
                                                                          const float2* __restrict__ box,
                                                                          float* __restrict__ rho_dst,
                                                                          unsigned long long* __restrict__ fix, int ntx, int nty,
                                                                          int zchunk) {
  constexpr int NF = 4;
  ATILE_LDS(NF);
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  const ABuf rs[NF] = { atile_rsrc(m, g, rho_fwd + sb1, after1), atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW),
                        atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW), atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3) };
  const ABuf rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w, i = m.x, hr0 = ATRPW * w + 1, col = lane + m.xs;
  const bool xin = i < g.W;

  auto body = [&](int k, unsigned fbm, unsigned fbc, unsigned fbp, const float (&rM)[NF][AFSZ], const float (&rC)[NF][AFSZ],
                  const float (&rP)[NF][AFSZ], auto pre_store) __attribute__((always_inline)) {
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    // per-cell global operands of both rows first (clamped addresses for the lanes that store nothing); the clamp bounds
    // hang off the traced cell
    float src[ATRPW]; int cell[ATRPW]; float2 bb[ATRPW]; bool inslab[ATRPW]; size_t oo[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      oo[r] = (size_t)k * g.HW + (size_t)(j < g.H ? j : g.H - 1) * g.W + (xin ? i : g.W - 1);
      src[r] = rho[sb1 + oo[r]];
      cell[r] = cell_in[sb1 + oo[r]];
    }
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      inslab[r] = (cell[r] >= g.HW) & (cell[r] < g.HW + g.DHW);
      bb[r] = box[sb1 + (size_t)(inslab[r] ? cell[r] - g.HW : 0)];
    }
    float o_d[ATRPW]; unsigned long long ws[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float fi = (float)i, fj = (float)j, fk = (float)kg;
      const float ctrx = fi + 0.5f, ctry = fj + 0.5f;
      const int rc0 = (hr0 + r) * ATP + col;
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      const unsigned nb = ((fbm >> (3 * r)) & 0x1ffu) | (((fbc >> (3 * r)) & 0x1ffu) << 9) | (((fbp >> (3 * r)) & 0x1ffu) << 18);
      const float f = rC[0][rc0];
      const float cen0 = 0.5f * (rC[1][rc0] + rC[1][rc0 + 1]);
      const float cen1 = 0.5f * (rC[2][rc0] + rC[2][rc0 + ATP]);
      const float cen2 = 0.5f * (rC[3][rc0] + rP[3][rc0]);
      float p0, p1, p2;                                               // displacement (-ndt) * cen with ndt = -dt
      const bool traced = atile_trace(dt * cen0, dt * cen1, dt * cen2, ctrx, ctry, ctrz, i, j, kg, nb, p0, p1, p2);
      const ALerp Ls = alerp(p0, p1, p2, fi, fj, fk);
      float cs[8];
      atile_corners<NF>(rM, rC, rP, 0, rc0, Ls.nx, Ls.ny, Ls.nz, cs);
      const bool allfluid = SAMPLE_OUTSIDE || __builtin_amdgcn_ballot_w64(!border & fluid & (nb != 0x7ffffffu)) == 0;   // (see the forward kernel)
      const float smp = allfluid ? atrilin(cs, Ls) : atrilin_fluid(cs, atile_corner_bits(nb, Ls), Ls);
      const float bwd = border ? 0.f : (fluid ? smp : f);
      float d = f;
      if (fluid) d = f + half_s * (src[r] - bwd);          // applied on border cells too (reference :371)
      const float mn = bb[r].x, mx = bb[r].y;
      const bool any = !(mn != mn);
      const float dc = any ? fmaxf(mn, fminf(mx, d)) : f;
      o_d[r] = border ? d : dc;
      ws[r] = __builtin_amdgcn_ballot_w64(live & !border & ((fluid & (!traced | !Ls.ok)) | !inslab[r]));
    }
    pre_store();
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      if (xin && j < g.H) rho_dst[sb1 + oo[r]] = o_d[r];
      if (lane == 0 && j < g.H) fix[m.word(g, k, j)] = ws[r];
    }
  };
  atile_march<NF>(g, m, rs, rs_f, ring0, ring1, ring2, ring3, fst0, fst1, body);
}

// ---------------------------------------------------------------------------------------------------
// Backward pass, velocity: sl_mac_bwd_clamp_cell_flat<true> (self-advection: orig == U) for the planes [K0, K0+KN).
// Ring fields: U_fwd x,y,z (sampled), U x,y,z (face velocities, clamp boxes): 71 KB of LDS, two workgroups per CU.
// (One component per workgroup with four ring fields and three workgroups per CU was measured: 3 x 138 us against 325 us,
// the per-plane overheads do not shrink with the work.)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * ATNW, FNX_AT_WPS_BV) void advect3d_bwd_vel_tile_kernel(GridDims g, float dt, float half_s,
                                                                       const float* __restrict__ U,
                                                                       const float* __restrict__ U_fwd,
                                                                       const float* __restrict__ flags,
                                                                       float* __restrict__ U_dst,
                                                                       unsigned long long* __restrict__ fix, int ntx, int nty,
                                                                       int zchunk) {
  constexpr int NF = 6;
  ATILE_LDS(NF);
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  const ABuf rs[NF] = { atile_rsrc(m, g, U_fwd + sb3, after3 + 2 * (size_t)g.DHW), atile_rsrc(m, g, U_fwd + sb3 + g.DHW, after3 + g.DHW),
                        atile_rsrc(m, g, U_fwd + sb3 + 2 * (size_t)g.DHW, after3), atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW),
                        atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW), atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3) };
  const ABuf rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w, i = m.x, hr0 = ATRPW * w + 1, col = lane + m.xs;
  const bool xin = i < g.W;

  auto body = [&](int k, unsigned fbm, unsigned fbc, unsigned fbp, const float (&rM)[NF][AFSZ], const float (&rC)[NF][AFSZ],
                  const float (&rP)[NF][AFSZ], auto pre_store) __attribute__((always_inline)) {
    (void)fbp;
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f, posz = (float)kg;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    float o_u[ATRPW][3]; unsigned long long ws[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float fi = (float)i, fj = (float)j, fk = (float)kg;
      const float ctrx = fi + 0.5f, ctry = fj + 0.5f, posx = fi, posy = fj;
      const int rc0 = (hr0 + r) * ATP + col;
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      // flags of the -1 neighbours along x, y, z (chk[a] is true for every non-border cell)
      const bool fmx = (fbc >> (3 * (r + 1) + 0)) & 1u, fmy = (fbc >> (3 * r + 1)) & 1u, fmz = (fbm >> (3 * (r + 1) + 1)) & 1u;
      // get_at_mac<true, false, 0/1/2> on U (ring fields 3, 4, 5), operand for operand
      const float x_c = rC[3][rc0], y_c = rC[4][rc0], z_c = rC[5][rc0];
      const float x_r = rC[3][rc0 + 1], y_u = rC[4][rc0 + ATP], z_f = rP[5][rc0];
      float v[3][3];
      v[0][0] = x_c;
      v[0][1] = 0.25f * (((y_c + rC[4][rc0 - 1]) + y_u) + rC[4][rc0 + ATP - 1]);
      v[0][2] = 0.25f * (((z_c + rC[5][rc0 - 1]) + z_f) + rP[5][rc0 - 1]);
      v[1][0] = 0.25f * (((x_c + rC[3][rc0 - ATP]) + x_r) + rC[3][rc0 - ATP + 1]);
      v[1][1] = y_c;
      v[1][2] = 0.25f * (((z_c + rC[5][rc0 - ATP]) + z_f) + rP[5][rc0 - ATP]);
      v[2][0] = 0.25f * (((x_c + rM[3][rc0]) + x_r) + rM[3][rc0 + 1]);
      v[2][1] = 0.25f * (((y_c + rM[4][rc0]) + y_u) + rM[4][rc0 + ATP]);
      v[2][2] = z_c;
      const float fwd0 = rC[0][rc0], fwd1 = rC[1][rc0], fwd2 = rC[2][rc0];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float vd0 = v[a][0] * dt, vd1 = v[a][1] * dt, vd2 = v[a][2] * dt;
        const ALerp L = alerp(ctrx + vd0, ctry + vd1, ctrz + vd2, fi, fj, fk);
        float c[8];
        atile_corners<NF>(rM, rC, rP, a, rc0, L.nx, L.ny, L.nz, c);
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
          float e[8];
          atile_corners<NF>(rM, rC, rP, 3 + a, rc0, rx == -1, ry == -1, rz == -1, e);
#pragma unroll
          for (int q = 0; q < 8; ++q) { mn = fminf(mn, e[q]); mx = fmaxf(mx, e[q]); }
        }
        const float fa = a == 0 ? fwd0 : (a == 1 ? fwd1 : fwd2);
        const float og = a == 0 ? x_c : (a == 1 ? y_c : z_c);
        const float bwd = fluid ? smp : (a == 0 ? fwd1 : (a == 1 ? 0.f : fa));     // Q1 pass-through of SL(fwd)
        const bool fm = a == 0 ? fmx : (a == 1 ? fmy : fmz);
        const bool skip = !fluid | !fm;
        const float corr = skip ? fa : fa + half_s * (og - bwd);
        o_u[r][a] = border ? 0.f : fmaxf(fminf(corr, mx), mn);
        ok &= okc & (L.ok | !fluid);
      }
      ws[r] = __builtin_amdgcn_ballot_w64(live & !border & !ok);
    }
    pre_store();
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      if (xin && j < g.H) {
        const size_t o = (size_t)k * g.HW + (size_t)j * g.W + i;
        U_dst[sb3 + o] = o_u[r][0];
        U_dst[sb3 + g.DHW + o] = o_u[r][1];
        U_dst[sb3 + 2 * (size_t)g.DHW + o] = o_u[r][2];
      }
      if (lane == 0 && j < g.H) fix[m.word(g, k, j)] = ws[r];
    }
  };
  atile_march<NF>(g, m, rs, rs_f, ring0, ring1, ring2, ring3, fst0, fst1, body);
}

// ---------------------------------------------------------------------------------------------------
// Backward pass of BOTH advections of a step in one march (round 6): sl_scalar_bwd_clamp_cell<true, false, SO> and
// sl_mac_bwd_clamp_cell_flat<true> for the planes [K0, K0+KN).  Ring fields: rho_fwd, U_fwd x,y,z, U x,y,z -- 7 of them, 81 600 bytes
// of LDS with the flags stages: two workgroups per CU, like the velocity kernel alone (163 200 of the CU's 163 840 bytes).  Against the
// two separate marches: U and the flags are streamed once instead of twice (1.43 -> ~1.0 GB per 16.8 M cells), one lead-in, one
// barrier / DMA issue / fluid-bit extraction per plane, and the density part's dependent global loads (traced cell -> its clamp bounds)
// are issued ahead of the velocity part's ~800 instructions instead of being waited for at the top of the step.
// Every expression is the two kernels' own (and through them the per-cell functions'), operand for operand.
// ---------------------------------------------------------------------------------------------------
#ifndef FNX_AT_WPS_B
#define FNX_AT_WPS_B 2
#endif
template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(64 * ATNW, FNX_AT_WPS_B) void advect3d_bwd_tile_kernel(GridDims g, float dt, float half_s,
                                                                   const float* __restrict__ rho,
                                                                   const float* __restrict__ rho_fwd,
                                                                   const int* __restrict__ cell_in,
                                                                   const float* __restrict__ U,
                                                                   const float* __restrict__ U_fwd,
                                                                   const float* __restrict__ flags,
                                                                   const float2* __restrict__ box,
                                                                   float* __restrict__ rho_dst, float* __restrict__ U_dst,
                                                                   unsigned long long* __restrict__ fix_s,
                                                                   unsigned long long* __restrict__ fix_v, int ntx, int nty,
                                                                   int zchunk) {
  constexpr int NF = 7;
  constexpr int FW = 1, FU = 4;                           // ring fields of U_fwd x and of U x
  ATILE_LDS(NF);
  ATile m;
  if (!atile_setup(m, g, ntx, nty, zchunk)) return;
  const size_t sb1 = (size_t)m.b * g.DHW, sb3 = (size_t)m.b * 3 * g.DHW;
  const size_t after1 = (size_t)(g.B - 1 - m.b) * g.DHW, after3 = 3 * after1;
  const ABuf rs[NF] = { atile_rsrc(m, g, rho_fwd + sb1, after1),
                        atile_rsrc(m, g, U_fwd + sb3, after3 + 2 * (size_t)g.DHW), atile_rsrc(m, g, U_fwd + sb3 + g.DHW, after3 + g.DHW),
                        atile_rsrc(m, g, U_fwd + sb3 + 2 * (size_t)g.DHW, after3), atile_rsrc(m, g, U + sb3, after3 + 2 * (size_t)g.DHW),
                        atile_rsrc(m, g, U + sb3 + g.DHW, after3 + g.DHW), atile_rsrc(m, g, U + sb3 + 2 * (size_t)g.DHW, after3) };
  const ABuf rs_f = atile_rsrc(m, g, flags + sb1, after1);
  const int lane = m.lane, w = m.w, i = m.x, hr0 = ATRPW * w + 1, col = lane + m.xs;
  const bool xin = i < g.W;

  auto body = [&](int k, unsigned fbm, unsigned fbc, unsigned fbp, const float (&rM)[NF][AFSZ], const float (&rC)[NF][AFSZ],
                  const float (&rP)[NF][AFSZ], auto pre_store) __attribute__((always_inline)) {
    const int kg = k + g.zoff;
    const float ctrz = (float)kg + 0.5f, posz = (float)kg;
    const bool kbord = (kg < 1) | (kg > g.Dglob - 2) | (k < 1) | (k > g.D - 2);
    // the density part's traced cells (per-cell global operands; clamped addresses for the lanes that store nothing) are asked for
    // first and arrive behind the velocity part; what hangs off them -- the clamp bounds -- and rho are fetched at the top of the
    // density part and arrive behind its traces and samples.  (Two registers live across the velocity part, which needs all 256.)
    int cell[ATRPW];
    auto own = [&](int r) __attribute__((always_inline)) {
      const int j = m.j0 + ATRPW * w + r;
      return (size_t)k * g.HW + (size_t)(j < g.H ? j : g.H - 1) * g.W + (xin ? i : g.W - 1);
    };
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) cell[r] = cell_in[sb1 + own(r)];
    float o_u[ATRPW][3]; unsigned long long wv[ATRPW];
    // ================= velocity: sl_mac_bwd_clamp_cell_flat (advect3d_bwd_vel_tile_kernel's body) =================
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float fi = (float)i, fj = (float)j, fk = (float)kg;
      const float ctrx = fi + 0.5f, ctry = fj + 0.5f, posx = fi, posy = fj;
      const int rc0 = (hr0 + r) * ATP + col;
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      const bool fmx = (fbc >> (3 * (r + 1) + 0)) & 1u, fmy = (fbc >> (3 * r + 1)) & 1u, fmz = (fbm >> (3 * (r + 1) + 1)) & 1u;
      const float x_c = rC[FU][rc0], y_c = rC[FU + 1][rc0], z_c = rC[FU + 2][rc0];
      const float x_r = rC[FU][rc0 + 1], y_u = rC[FU + 1][rc0 + ATP], z_f = rP[FU + 2][rc0];
      float v[3][3];
      v[0][0] = x_c;
      v[0][1] = 0.25f * (((y_c + rC[FU + 1][rc0 - 1]) + y_u) + rC[FU + 1][rc0 + ATP - 1]);
      v[0][2] = 0.25f * (((z_c + rC[FU + 2][rc0 - 1]) + z_f) + rP[FU + 2][rc0 - 1]);
      v[1][0] = 0.25f * (((x_c + rC[FU][rc0 - ATP]) + x_r) + rC[FU][rc0 - ATP + 1]);
      v[1][1] = y_c;
      v[1][2] = 0.25f * (((z_c + rC[FU + 2][rc0 - ATP]) + z_f) + rP[FU + 2][rc0 - ATP]);
      v[2][0] = 0.25f * (((x_c + rM[FU][rc0]) + x_r) + rM[FU][rc0 + 1]);
      v[2][1] = 0.25f * (((y_c + rM[FU + 1][rc0]) + y_u) + rM[FU + 1][rc0 + ATP]);
      v[2][2] = z_c;
      const float fwd0 = rC[FW][rc0], fwd1 = rC[FW + 1][rc0], fwd2 = rC[FW + 2][rc0];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float vd0 = v[a][0] * dt, vd1 = v[a][1] * dt, vd2 = v[a][2] * dt;
        const ALerp L = alerp(ctrx + vd0, ctry + vd1, ctrz + vd2, fi, fj, fk);
        float c[8];
        atile_corners<NF>(rM, rC, rP, FW + a, rc0, L.nx, L.ny, L.nz, c);
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
          float e[8];
          atile_corners<NF>(rM, rC, rP, FU + a, rc0, rx == -1, ry == -1, rz == -1, e);
#pragma unroll
          for (int q = 0; q < 8; ++q) { mn = fminf(mn, e[q]); mx = fmaxf(mx, e[q]); }
        }
        const float fa = a == 0 ? fwd0 : (a == 1 ? fwd1 : fwd2);
        const float og = a == 0 ? x_c : (a == 1 ? y_c : z_c);
        const float bwd = fluid ? smp : (a == 0 ? fwd1 : (a == 1 ? 0.f : fa));     // Q1 pass-through of SL(fwd)
        const bool fm = a == 0 ? fmx : (a == 1 ? fmy : fmz);
        const bool skip = !fluid | !fm;
        const float corr = skip ? fa : fa + half_s * (og - bwd);
        o_u[r][a] = border ? 0.f : fmaxf(fminf(corr, mx), mn);
        ok &= okc & (L.ok | !fluid);
      }
      wv[r] = __builtin_amdgcn_ballot_w64(live & !border & !ok);
    }
    // ================= density: sl_scalar_bwd_clamp_cell (advect3d_bwd_scalar_tile_kernel's body) =================
    float src[ATRPW]; float2 bb[ATRPW]; bool inslab[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      inslab[r] = (cell[r] >= g.HW) & (cell[r] < g.HW + g.DHW);
      bb[r] = box[sb1 + (size_t)(inslab[r] ? cell[r] - g.HW : 0)];
      src[r] = rho[sb1 + own(r)];
    }
    float o_d[ATRPW]; unsigned long long ws[ATRPW];
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      const bool border = (i < 1) | (i > g.W - 2) | (j < 1) | (j > g.H - 2) | kbord;
      const bool live = xin & (j < g.H);
      const float fi = (float)i, fj = (float)j, fk = (float)kg;
      const float ctrx = fi + 0.5f, ctry = fj + 0.5f;
      const int rc0 = (hr0 + r) * ATP + col;
      const bool fluid = (fbc >> (3 * (r + 1) + 1)) & 1u;
      const unsigned nb = ((fbm >> (3 * r)) & 0x1ffu) | (((fbc >> (3 * r)) & 0x1ffu) << 9) | (((fbp >> (3 * r)) & 0x1ffu) << 18);
      const float f = rC[0][rc0];
      const float cen0 = 0.5f * (rC[FU][rc0] + rC[FU][rc0 + 1]);
      const float cen1 = 0.5f * (rC[FU + 1][rc0] + rC[FU + 1][rc0 + ATP]);
      const float cen2 = 0.5f * (rC[FU + 2][rc0] + rP[FU + 2][rc0]);
      float p0, p1, p2;                                               // displacement (-ndt) * cen with ndt = -dt
      const bool traced = atile_trace(dt * cen0, dt * cen1, dt * cen2, ctrx, ctry, ctrz, i, j, kg, nb, p0, p1, p2);
      const ALerp Ls = alerp(p0, p1, p2, fi, fj, fk);
      float cs[8];
      atile_corners<NF>(rM, rC, rP, 0, rc0, Ls.nx, Ls.ny, Ls.nz, cs);
      const bool allfluid = SAMPLE_OUTSIDE || __builtin_amdgcn_ballot_w64(!border & fluid & (nb != 0x7ffffffu)) == 0;   // (see the forward kernel)
      const float smp = allfluid ? atrilin(cs, Ls) : atrilin_fluid(cs, atile_corner_bits(nb, Ls), Ls);
      const float bwd = border ? 0.f : (fluid ? smp : f);
      float d = f;
      if (fluid) d = f + half_s * (src[r] - bwd);          // applied on border cells too (reference :371)
      const float mn = bb[r].x, mx = bb[r].y;
      const bool any = !(mn != mn);
      const float dc = any ? fmaxf(mn, fminf(mx, d)) : f;
      o_d[r] = border ? d : dc;
      ws[r] = __builtin_amdgcn_ballot_w64(live & !border & ((fluid & (!traced | !Ls.ok)) | !inslab[r]));
    }
    pre_store();
#pragma unroll
    for (int r = 0; r < ATRPW; ++r) {
      const int j = m.j0 + ATRPW * w + r;
      if (xin && j < g.H) {
        const size_t o = (size_t)k * g.HW + (size_t)j * g.W + i;
        rho_dst[sb1 + o] = o_d[r];
        U_dst[sb3 + o] = o_u[r][0];
        U_dst[sb3 + g.DHW + o] = o_u[r][1];
        U_dst[sb3 + 2 * (size_t)g.DHW + o] = o_u[r][2];
      }
      if (lane == 0 && j < g.H) { const size_t wi = m.word(g, k, j); fix_s[wi] = ws[r]; fix_v[wi] = wv[r]; }
    }
  };
  atile_march<NF>(g, m, rs, rs_f, ring0, ring1, ring2, ring3, fst0, fst1, body);
}

// ---------------------------------------------------------------------------------------------------
// Fix-up launches: the per-cell functions on the cells the tile kernels flagged.  One thread per bitmap word (a 64-cell row
// segment of the compute window): nothing flagged = one 8-byte load per thread.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool afix_decode(const GridDims& g, int ntx, size_t t, CellId& c, size_t& wi) {
  const size_t nrow = (size_t)g.B * g.KN * g.H;
  if (t >= nrow * ntx) return false;
  const int bx = (int)(t % ntx); size_t q = t / ntx;
  c.j = (int)(q % g.H); q /= g.H;
  c.k = g.K0 + (int)(q % g.KN); c.b = (int)(q / g.KN);
  c.i = bx * 64; c.valid = true;
  wi = (((size_t)c.b * g.D + c.k) * g.H + c.j) * ntx + bx;
  return true;
}

template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256) void advect3d_fwd_fix_kernel(GridDims g, float dt, const float* __restrict__ rho,
                                                               const float* __restrict__ U, const float* __restrict__ flags,
                                                               float* __restrict__ rho_fwd, int* __restrict__ cell_out,
                                                               float* __restrict__ U_fwd,
                                                               const unsigned long long* __restrict__ fix_s,
                                                               const unsigned long long* __restrict__ fix_v, int ntx) {
  CellId c; size_t wi;
  if (!afix_decode(g, ntx, (size_t)blockIdx.x * 256 + threadIdx.x, c, wi)) return;
  unsigned long long ws = fix_s ? fix_s[wi] : 0ull, wv = fix_v ? fix_v[wi] : 0ull;   // (a stand-alone advection has one bitmap)
  const int i0 = c.i;
  for (unsigned long long a = ws | wv; a != 0; a &= a - 1) {
    const int bit = __builtin_ctzll(a);
    c.i = i0 + bit;
    if ((ws >> bit) & 1ull) sl_scalar_cell<true, false, SAMPLE_OUTSIDE>(g, c, dt, rho, U, flags, rho_fwd, cell_out);
    if ((wv >> bit) & 1ull) sl_mac_cell_flat<true>(g, c, dt, U, U, flags, U_fwd);
  }
}

template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(256) void advect3d_bwd_fix_kernel(GridDims g, float dt, float half_s, const float* __restrict__ rho,
                                                               const float* __restrict__ rho_fwd, const int* __restrict__ cell_in,
                                                               const float* __restrict__ U, const float* __restrict__ U_fwd,
                                                               const float* __restrict__ flags, const float2* __restrict__ box,
                                                               float* __restrict__ rho_dst, float* __restrict__ U_dst,
                                                               const unsigned long long* __restrict__ fix_s,
                                                               const unsigned long long* __restrict__ fix_v, int ntx) {
  CellId c; size_t wi;
  if (!afix_decode(g, ntx, (size_t)blockIdx.x * 256 + threadIdx.x, c, wi)) return;
  unsigned long long ws = fix_s ? fix_s[wi] : 0ull, wv = fix_v ? fix_v[wi] : 0ull;
  const int i0 = c.i;
  for (unsigned long long a = ws | wv; a != 0; a &= a - 1) {
    const int bit = __builtin_ctzll(a);
    c.i = i0 + bit;
    if ((ws >> bit) & 1ull) sl_scalar_bwd_clamp_cell<true, false, SAMPLE_OUTSIDE>(g, c, dt, half_s, rho, rho_fwd, cell_in, U, flags, box, rho_dst);
    if ((wv >> bit) & 1ull) sl_mac_bwd_clamp_cell_flat<true>(g, c, dt, half_s, U, U_fwd, U, flags, U_dst);
  }
}
