// FNX_PRECISION_BF16X6 -- the 3x3(x3) MFMA layers in the Winograd F(2x2,3x3) domain on the bf16 matrix cores with
// fp32-equivalent products (included by fnx_cnn.hip inside its anonymous namespace; opt-in, never the default).
//
// Arithmetic.  Every fp32 operand of the Winograd-domain GEMMs (transformed weight G g G^T, transformed input B^T d B) is cut
// into three bf16 pieces by truncation, x = h + m + l EXACTLY (8 + 8 + 8 significand bits), and a product a*b is evaluated as
//     ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh                (six v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// -- each bf16 x bf16 product is exact in fp32; what is dropped (am*bl + al*bm + al*bl) is below 2^-23 of |a*b|, the size of the
// rounding of one fp32 multiply.  Six bf16 MFMAs of K = 16 take 6 x 32 cycles where the exact-fp32 path (v_mfma_f32_32x32x2_f32,
// K = 2) takes 8 x 64: 0.375 of the matrix-core time for the same contraction.
//
// Mapping (one workgroup = 8 waves = one 64-output-channel x 64-block tile, block = 2x2 output pixels, tile = 32 x 8 pixels):
//   wave (r, cg): position ROW r of the 4x4 Winograd positions, output channels cg*32..+31, all 64 blocks (two 32-block MFMA
//   tiles) -> 4 positions x 2 tiles x 16 = 128 accumulator registers.
//   stage = 16 input channels (one MFMA K); a stage is worked off as two HALF-STAGES (position columns {0,1}, then {2,3}): one
//   barrier per half-stage, 24 MFMAs per wave between barriers.
//   A (weights): pre-split at pack time and stored in exactly the MFMA operand layout per (wave, half-stage) -- a position row
//     and 32 output channels belong to ONE wave, so the weights never pass through LDS: six 16-byte loads per lane and
//     half-stage straight into registers, one half-stage ahead.
//   B (inputs): halo tile [16 ch][10][34] fp32 global -> registers -> LDS (two copies, fetched two stages ahead); the input
//     transform of a half-stage (two position columns of B^T d B: 16 additions per channel and block), the three-way split and
//     the packing of channel pairs run as one thread per (block, channel pair) and write the LDS image
//     [position 8][piece 3][k half 2][block 64][8 channels] with conflict-free 4-byte stores; the MFMA operand is then one
//     conflict-free ds_read_b128 per piece.
//   Output transform: a wave holds row r of M; u = M[r][.] A is local, Y = A^T u needs the four row waves of a channel group:
//     they exchange through LDS (the input image's memory) and each finishes one (block tile, channel half) quarter.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WB_C = 16;                       // input channels per stage
constexpr int WB_ROWS = 10, WB_COLS = 34;      // halo tile: 8 pixel rows + 2, 32 pixels + 2
// floats per channel of the halo tile in LDS: 10 x 34 = 340, padded to six 64-lane DMA rows (384) + 4, so that the channel PAIRS
// the transform's lanes read side by side (stride 2 x 388 = 8 banks mod 32) fall on different LDS banks
constexpr int WB_CH = 388;
constexpr int WB_RAW = WB_C * WB_CH;           // floats per halo tile (24 832 B)
constexpr int WB_V16 = 8 * 3 * 2 * 64;         // 16-byte units per half-stage input image (49 152 B)
constexpr int WB_NT = 512;

// truncation split of an fp32 into three bf16 pieces (returned as fp32 values whose low 16 bits are zero; lo keeps whatever is
// left, which fits 8 bits for a normal x)
__device__ __forceinline__ void wb_split(float x, float& h, float& m, float& l) {
  h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
  const float r1 = x - h;
  m = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
  l = r1 - m;
}
// the upper halves of two fp32 as one word: a in the low 16 bits (the even channel), b in the high 16
__device__ __forceinline__ unsigned wb_pack(float a, float b) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

// G g G^T fp32 ([dz][16][Cin][Cout], pack_layer_wino_kernel) -> the A-operand image of conv3_wbf_kernel:
//   [dz][Cin/16][Cout/64][cg 2][row 4][half-stage 2][column 2][piece 3][lane 64][8 bf16]
// lane = (output channel within 32, k half); element e of a lane = input channel chunk*16 + khalf*8 + e
__global__ void pack_wbf_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int cin, int cout, int kd) {
  const int nchunk = cin / WB_C, ngrp = cout / 64;
  const size_t n = (size_t)kd * nchunk * ngrp * 2 * 4 * 2 * 2 * 3 * 64 * 4;     // 32-bit words
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    size_t t = q;
    const int e2 = (int)(t % 4); t /= 4;
    const int lane = (int)(t % 64); t /= 64;
    const int sp = (int)(t % 3); t /= 3;
    const int jj = (int)(t % 2); t /= 2;
    const int hs = (int)(t % 2); t /= 2;
    const int r = (int)(t % 4); t /= 4;
    const int cg = (int)(t % 2); t /= 2;
    const int grp = (int)(t % ngrp); t /= ngrp;
    const int chunk = (int)(t % nchunk); const int dz = (int)(t / nchunk);
    const int pos = r * 4 + 2 * hs + jj;
    const int co = grp * 64 + cg * 32 + (lane & 31);
    const int ch = chunk * WB_C + (lane >> 5) * 8 + 2 * e2;
    float p[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) wb_split(src[((size_t)(dz * 16 + pos) * cin + ch + k) * cout + co], p[k][0], p[k][1], p[k][2]);
    dst[q] = wb_pack(p[0][sp], p[1][sp]);
  }
}
inline size_t wbf_image_words(const ConvLayer& L, bool is3d) { return (size_t)(is3d ? 3 : 1) * 16 * L.cin * L.cout * 3 / 2; }

#ifndef WB_ABL
#define WB_ABL 0          // timing probes (tools/wbf_ablation.sh): compile-time bit set of pieces left out; 0 in the product
#endif
typedef __attribute__((address_space(3))) void* WbLds;
typedef __attribute__((address_space(3))) float* WbLdsF;
// s_waitcnt lgkmcnt(0) vmcnt(N) (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 left at its maximum, lgkmcnt 11:8)
#define WB_WAIT_LDS_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 0xf) | (7 << 4) | ((((N) >> 4) & 3) << 14))

// 8-byte LDS read the compiler does not see as one.  The halo tile is written by DMA into a copy chosen at run time and read by
// the transform from the OTHER copy; the compiler cannot prove that (also not with __restrict__ on the phase's parameters) and
// puts an s_waitcnt vmcnt in front of the first ds_read after every DMA issue -- a global round trip per stage.  As inline asm the
// reads carry no memory operand; the phase waits for them itself (lgkmcnt(0): LDS returns in order, and the compiler's own
// counts stay on the safe side with more operations in flight).  The result registers are NOT valid before that wait: this holds
// only while the compiler has no reason to touch them in between -- a variant of this kernel that spilled (256 VGPRs + scratch)
// copied them early and computed garbage; the kernel as built uses 234 VGPRs and no scratch, and the parity tests would show it.
// Measured and not kept (profiles/r04/e_wbf_ablation_3d_register_staged_fetch.txt): the halo through registers in 16-byte quads
// instead of the DMA (needs W % 4 == 0, 248 VGPRs): same bits, 61.5 vs 58.6 ms per 256^3 forward -- the kernel is bound by the
// LDS data path (per phase ~1 800 cycles of ds_read_b128 / ds_read_b64 / ds_write_b32 traffic against 1 536 of MFMA) and by how
// little of it two waves per SIMD overlap with their MFMAs, not by how the halo arrives.
template <int OFF>
__device__ __forceinline__ f32x2 wb_lds_read64(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// NP = 6: FNX_PRECISION_BF16X6 (all six products).  NP = 3: FNX_PRECISION_BF16X3 -- only ah*bh + ah*bm + am*bh (the three products
// that do not involve a low piece): the dropped terms are below 2^-15 of |a*b|, an "accurate bf16" mode with its own tolerance
// (1e-4 |ref|max in the tests; measured error in profiles/r05).  The phase keeps its 24 slots and its filler schedule -- the slots of
// the three small products issue nothing --, the low pieces are neither stored by the transform nor read (LDS traffic per phase 208 ->
// 160 KiB) nor loaded as A operands.
template <bool IS3D, int NP>
__global__ __launch_bounds__(WB_NT, 2) void conv3_wbf_kernel(ConvArgs a, const u32x4* __restrict__ wq, int ntx, int nty) {
  static_assert(NP == 6 || NP == 3, "six products, or the three without a low piece");
  constexpr int NPIECE = NP == 6 ? 3 : 2;
  // (separate __shared__ objects: the compiler cannot tell an LDS-DMA into one copy from the ds_reads of the other when they
  // are one array, and then parks a vmcnt(0) in front of every read)
  __shared__ __attribute__((aligned(16))) float raw0[WB_RAW];
  __shared__ __attribute__((aligned(16))) float raw1[WB_RAW];
  __shared__ __attribute__((aligned(16))) u32x4 vbuf0[WB_V16];
  __shared__ __attribute__((aligned(16))) u32x4 vbuf1[WB_V16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n31 = lane & 31, khl = lane >> 5;
  const int r = wave & 3, cg = wave >> 2;                    // MFMA role: position row, output-channel group
  const int ngrp = a.cout / 64, nchunk = a.cin / WB_C;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;
  const unsigned stage_bytes = (unsigned)((size_t)WB_C * vol * 4 - 1) + 1u;

  // Tile numbering.  3D: x fastest, then y, output-channel group, z, sample -- workgroups go to the 8 XCDs round-robin
  // (blockIdx % 8), so with 8 tiles across 256 cells an XCD keeps one x column and walks it in y, then z.  2D: XCD x takes the
  // x-th eighth of the sequence and the sequence runs channel group fastest, so the workgroups an XCD runs side by side include
  // both channel groups of a tile: the second read of its input hits that XCD's L2 (1024^2 forward: conv3_wbf 1.68 -> 1.60 ms).
  // (3D measured with the same renumbering, alone and with 4 / 8 / 16 planes of one (x, y) tile side by side: 60.3-61.7 ms per
  // 256^3 forward against 58.6 in the plain order -- the planes are 256 KB apart and crowd the same L2 sets.)
  int t = blockIdx.x;
  int tx, ty, grp, tz, tb;
  if (!IS3D && (gridDim.x & 7) == 0) {
    t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
    grp = t % ngrp; t /= ngrp;
    tx = t % ntx; t /= ntx;
    ty = t % nty; tb = t / nty; tz = 0;
  } else {
    tx = t % ntx; t /= ntx;
    ty = t % nty; t /= nty;
    grp = t % ngrp; t /= ngrp;
    tz = t % a.D; tb = t / a.D;
  }
  const int tx0 = tx * 32, ty0 = ty * 8;
  const int dz_lo = IS3D && tz == 0 ? 1 : 0, dz_hi = IS3D ? (tz == a.D - 1 ? 2 : 3) : 1;
  const int nstage = (dz_hi - dz_lo) * nchunk;              // >= 2 (the host checks Cin >= 32)

  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  // ---- halo fetch: the [16][10][34] tile goes global -> LDS by DMA (buffer_load_dword ... lds: 64 consecutive floats per wave
  // instruction).  A channel is six such rows (340 positions + padding); the stage's 96 rows are dealt round-robin to the
  // 8 waves, 12 each, ALL of them issued unconditionally by every wave (a DMA behind a branch makes the compiler assume it may
  // not have been issued, and every later wait for an A load then also waits for the DMA): row i = wave + 8 k is channel i / 6,
  // part i % 6 -- three distinct parts per wave, so three per-lane offsets; the channel travels in the scalar offset.  A position
  // outside the image (or in the padding) gets an out-of-range offset: the DMA writes 0.  Behind the last stage the fetch
  // stream re-reads the last stage (into a copy nobody reads any more): the phases stay the same straight-line code.
  const float* xs = a.x + (size_t)tb * a.cin * vol;
  unsigned voff3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int part = (wave + 8 * k) % 6;
    const int p = part * 64 + lane, row = p / WB_COLS, col = p - row * WB_COLS;
    const int gx = tx0 - 1 + col, gy = ty0 - 1 + row;
    const bool ok = (p < WB_ROWS * WB_COLS) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    voff3[k] = ok ? (unsigned)(((size_t)gy * a.W + gx) * 4) : 0xfffffff0u;
  }
  const unsigned chan_bytes = (unsigned)(vol * 4);
  int rdz = dz_lo, rc0 = 0;                                   // cursor of the fetch stream
  auto fetch = [&](float* dst) __attribute__((always_inline)) {
    const int zz = IS3D ? tz + rdz - 1 : 0;
    const BufRsrcC rs = make_rsrc_c(xs + (size_t)rc0 * vol + (size_t)zz * plane, stage_bytes - (unsigned)((size_t)zz * plane * 4));
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int i = wave + 8 * k, cc = i / 6, part = i - 6 * cc;       // (scalar)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (WbLds)((WbLdsF)&dst[0] + cc * WB_CH + part * 64), 4, voff3[k % 3],
                                               cc * chan_bytes, 0, 0);
    }
    const bool wrap = rc0 + WB_C >= a.cin;
    const bool last = wrap && rdz + 1 >= dz_hi;               // stay on the last stage
    rc0 = last ? rc0 : (wrap ? 0 : rc0 + WB_C);
    rdz = last ? rdz : (wrap ? rdz + 1 : rdz);
  };

  // ---- A operands: [dz][chunk][grp][cg][r][hs][jj][piece][lane] 16-byte units; half-stage q of this tile is block
  // (dz_lo * nchunk + q / 2) of the wave's stream, half q & 1.  Register sets A[q & 1][jj][piece].
  const u32x4* wa = wq + ((((size_t)(dz_lo * nchunk) * ngrp + grp) * 2 + cg) * 4 + r) * (2 * 6 * 64) + lane;
  const size_t wstage = (size_t)ngrp * 2 * 4 * 2 * 6 * 64;
  u32x4 A[2][2][3];
  auto load_a = [&](u32x4 (&Aj)[3], int q, int jj) __attribute__((always_inline)) {
    const u32x4* p = wa + (size_t)(q >> 1) * wstage + ((q & 1) * 6 + jj * 3) * 64;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) Aj[i] = p[i * 64];
  };
  auto load_a1 = [&](u32x4& dst, int q, int jj, int i) __attribute__((always_inline)) {      // one piece
    dst = wa[(size_t)(q >> 1) * wstage + ((q & 1) * 6 + jj * 3 + i) * 64];
  };

  // ---- input transform + split of one half-stage, in pieces the phase schedule deals out between the MFMAs.
  // Thread = (block n, channel pair cp).  Wave: k half kt, block-column half, block-row pair; lane: cp fastest, then 8 blocks in
  // x, then 2 in y -- 32 consecutive lanes then store 32 consecutive words of the image (8 blocks x 16 B) and read, per 16
  // lanes, 4 channel pairs x 4 blocks x 8 B from 32 different banks (WB_CH): no LDS bank conflicts on either side.
  const int kt = wave & 1, cp = lane & 3;
  const int bxt = 8 * ((wave >> 1) & 1) + ((lane >> 2) & 7), bt = 2 * (wave >> 2) + (lane >> 5);
  const int roff = (kt * 8 + 2 * cp) * WB_CH + (2 * bt) * WB_COLS + 2 * bxt;
  const int woff = (kt * 64 + bt * 16 + bxt) * 4 + cp;          // 32-bit word within a (position, piece) image
  f32x2 d01[4], d23[4];
  float v[2][4][2];                                             // [channel][row r][column jj]
  auto xf_load = [&](auto ch_, const float* rawsrc) __attribute__((always_inline)) {
    constexpr int CH = decltype(ch_)::value;
    const unsigned ad = (unsigned)(size_t)(WbLdsF)rawsrc + (unsigned)roff * 4u;
    d01[0] = wb_lds_read64<(CH * WB_CH + 0 * WB_COLS) * 4>(ad); d23[0] = wb_lds_read64<(CH * WB_CH + 0 * WB_COLS + 2) * 4>(ad);
    d01[1] = wb_lds_read64<(CH * WB_CH + 1 * WB_COLS) * 4>(ad); d23[1] = wb_lds_read64<(CH * WB_CH + 1 * WB_COLS + 2) * 4>(ad);
    d01[2] = wb_lds_read64<(CH * WB_CH + 2 * WB_COLS) * 4>(ad); d23[2] = wb_lds_read64<(CH * WB_CH + 2 * WB_COLS + 2) * 4>(ad);
    d01[3] = wb_lds_read64<(CH * WB_CH + 3 * WB_COLS) * 4>(ad); d23[3] = wb_lds_read64<(CH * WB_CH + 3 * WB_COLS + 2) * 4>(ad);
  };
  auto xf_combo = [&](auto hs_, int ch) __attribute__((always_inline)) {
    constexpr int HS = decltype(hs_)::value;
    // lgkmcnt(0) for the asm reads of xf_load, with their registers as operands: the wait is ordered before every consumer
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d01[0]), "+v"(d01[1]), "+v"(d01[2]), "+v"(d01[3]), "+v"(d23[0]), "+v"(d23[1]),
                 "+v"(d23[2]), "+v"(d23[3]));
    float c[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (HS == 0) { c[i][0] = d01[i].x - d23[i].x; c[i][1] = d01[i].y + d23[i].x; }
      else { c[i][0] = d23[i].x - d01[i].y; c[i][1] = d01[i].y - d23[i].y; }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      v[ch][0][jj] = c[0][jj] - c[2][jj]; v[ch][1][jj] = c[1][jj] + c[2][jj];
      v[ch][2][jj] = c[2][jj] - c[1][jj]; v[ch][3][jj] = c[1][jj] - c[3][jj];
    }
  };
  auto xf_unit = [&](int k, u32x4* vdst) __attribute__((always_inline)) {   // position k = 2 r + jj of the half-stage
    unsigned* o = (unsigned*)vdst + woff + (k * 3 * 2 * 64) * 4;
    float p0[3], p1[3];
    wb_split(v[0][k >> 1][k & 1], p0[0], p0[1], p0[2]);
    wb_split(v[1][k >> 1][k & 1], p1[0], p1[1], p1[2]);
#pragma unroll
    for (int sp = 0; sp < NPIECE; ++sp) o[(sp * 2 * 64) * 4] = wb_pack(p0[sp], p1[sp]);
  };

  // ---- MFMA stream.  acc[position column j][block tile]; B operand sets X and Y alternate between the groups of six MFMAs
  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][tt][e] = 0.f;
  u32x4 BX[3], BY[3];
  auto read_b1 = [&](u32x4& dst, const u32x4* vsrc, int jj, int tt, int sp) __attribute__((always_inline)) {   // one piece
    dst = vsrc[((r * 2 + jj) * 3 * 2 + khl) * 64 + 32 * tt + n31 + sp * 2 * 64];
  };
  // product i of a group: (al,bh) (ah,bl) (am,bm) (am,bh) (ah,bm) (ah,bh) -- the small terms first
  auto mfma1 = [&](f32x16& c, const u32x4 (&Aj)[3], const u32x4 (&B)[3], int i) __attribute__((always_inline)) {
    const int ia = i == 0 ? 2 : (i == 2 || i == 3 ? 1 : 0), ib = i == 1 ? 2 : (i == 2 || i == 4 ? 1 : 0);
    if (NP == 3 && i < 3) return;                            // (al,bh) (ah,bl) (am,bm): not in the three-product mode
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Aj[ia]), __builtin_bit_cast(bf16x8, B[ib]), c, 0, 0, 0);
  };

  // One phase = half-stage q (HS = q & 1), 24 MFMA slots in four groups:
  //   G0 (slots 0-5, ROT)  the LAST group of half-stage q-1, (jj 1, tile 1): its operands were read before the barrier, so the
  //                        matrix pipe restarts at once behind it while this phase's first operands are on their way
  //   G1, G2, G3           (jj 0, tile 0), (jj 0, tile 1), (jj 1, tile 0) of half-stage q
  // and between the MFMAs: the B reads of the next group (slots 1, 7, 13, 19), the A loads of half-stage q+1 (slots 2, 8), the
  // input transform of half-stage q+1 into the other image (slots 0-17), and in odd phases the halo DMA of stage s+2 (slot 20).
  // (__restrict__: the halo copy the DMA fills and the one the transform reads are chosen at run time; without the promise that
  // they are different objects the compiler waits for the DMA -- a global round trip -- in front of the next LDS read)
  // LATE: the two waves of a SIMD (w and w + 4) run the same code at the same time; with the same filler slots both want the
  // VALU in the same slots and the matrix pipe in the others.  The second one gets its eight split-and-store units (the bulk of
  // the VALU work) in the slots the first one leaves empty.
  auto phase = [&](auto hs_, auto rot_, auto next_, auto late_, int q, const u32x4* __restrict__ vcur, u32x4* __restrict__ vnext,
                   const float* __restrict__ rawnext, float* __restrict__ rawfree) __attribute__((always_inline)) {
    constexpr int HS = decltype(hs_)::value;
    constexpr bool ROT = decltype(rot_)::value, NEXT = decltype(next_)::value, LATE = decltype(late_)::value;
    constexpr int US[8] = {LATE ? 12 : 8, LATE ? 13 : 9, LATE ? 18 : 10, LATE ? 19 : 11, LATE ? 20 : 14, LATE ? 21 : 15, LATE ? 22 : 16,
                           LATE ? 23 : 17};
    using HT = std::integral_constant<int, 1 - HS>;                 // the half-stage the transform works on
#pragma unroll
    for (int slot = 0; slot < 24; ++slot) {
      const int g = slot / 6, i = slot % 6;
      if (!(WB_ABL & 4)) {
        if (g == 0) { if (ROT) mfma1(acc[2 * (1 - HS) + 1][1], A[1 - HS][1], BX, i); }
        else if (g == 1) mfma1(acc[2 * HS][0], A[HS][0], BY, i);
        else if (g == 2) mfma1(acc[2 * HS][1], A[HS][0], BX, i);
        else mfma1(acc[2 * HS + 1][0], A[HS][1], BY, i);
      }
      __builtin_amdgcn_sched_barrier(0);
      // B reads of the next group one slot behind the group's first MFMA (the set is free from then on); LDS loads of the
      // transform are consumed three slots after their issue, and no store of the transform comes before its last load has
      // been waited for (the waits are lgkmcnt(0), LDS operations return in order)
      // (one 1-KiB operand per slot: bursts of three from all eight waves at once fill the LDS / vector-memory queues and the
      // waves then wait at the issue of whatever comes next)
      if (!(WB_ABL & 8)) {
        // piece order of a group's MFMAs: B pieces h, l, m, h, m, h are first needed by products 0, 1, 2 -> read h, l, m
        // (NP == 3: the low piece, slots 2 / 8 / 14 / 20, is not read)
        if (slot >= 1 && slot <= 3 && (NP == 6 || slot != 2)) read_b1(BY[slot == 1 ? 0 : (slot == 2 ? 2 : 1)], vcur, 0, 0, slot == 1 ? 0 : (slot == 2 ? 2 : 1));
        if (slot >= 7 && slot <= 9 && (NP == 6 || slot != 8)) read_b1(BX[slot == 7 ? 0 : (slot == 8 ? 2 : 1)], vcur, 0, 1, slot == 7 ? 0 : (slot == 8 ? 2 : 1));
        if (slot >= 13 && slot <= 15 && (NP == 6 || slot != 14)) read_b1(BY[slot == 13 ? 0 : (slot == 14 ? 2 : 1)], vcur, 1, 0, slot == 13 ? 0 : (slot == 14 ? 2 : 1));
        if (slot >= 19 && slot <= 21 && (NP == 6 || slot != 20)) read_b1(BX[slot == 19 ? 0 : (slot == 20 ? 2 : 1)], vcur, 1, 1, slot == 19 ? 0 : (slot == 20 ? 2 : 1));
      }
      if (NEXT) {
        if (!(WB_ABL & 16)) {
          if (slot == 0 || slot == 4 || (slot == 5 && NP == 6)) load_a1(A[1 - HS][0][slot == 0 ? 0 : slot - 3], q + 1, 0, slot == 0 ? 0 : slot - 3);
          if (slot == 10 || slot == 12 || (slot == 16 && NP == 6)) load_a1(A[1 - HS][1][slot == 10 ? 0 : (slot == 12 ? 1 : 2)], q + 1, 1, slot == 10 ? 0 : (slot == 12 ? 1 : 2));
        }
        if (!(WB_ABL & 2)) {
          if (slot == 0) xf_load(H0{}, rawnext);
          if (slot == 3) { xf_combo(HT{}, 0); xf_load(H1{}, rawnext); }
          if (slot == 6) xf_combo(HT{}, 1);
        }
        if (!(WB_ABL & 1)) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (slot == US[k]) xf_unit(k, vnext);
        }
      }
      if (HS == 1 && NEXT && !(WB_ABL & 32) && slot == (LATE ? 16 : 20)) fetch(rawfree);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- prologue: halo tiles of stages 0 and 1 by DMA, A and the input image of half-stage 0
  fetch(raw0);
  fetch(raw1);
  load_a(A[0][0], 0, 0);
  load_a(A[0][1], 0, 1);
  WB_WAIT_LDS_VM(2 * NPIECE + 12);                              // stage 0's DMA has landed (stage 1's and A may still be in flight)
  __builtin_amdgcn_s_barrier();
  xf_load(H0{}, raw0); xf_combo(H0{}, 0);
  xf_load(H1{}, raw0); xf_combo(H0{}, 1);
#pragma unroll
  for (int k = 0; k < 8; ++k) xf_unit(k, vbuf0);
  WB_WAIT_LDS_VM(0);
  __builtin_amdgcn_s_barrier();

  // ---- main loop: two phases per stage.  Even phase (columns 0, 1 of stage s): the transform reads stage s again (columns 2, 3);
  // odd phase: the transform reads stage s+1, and the DMA of stage s+2 goes into the copy stage s has left.  Barriers: LDS writes
  // and reads of the phase are through (lgkmcnt 0); at the END OF AN EVEN phase the DMA issued in the odd phase before it must
  // have landed -- the only younger VMEM instructions are that even phase's A loads (six, or four without the low pieces).
  auto run = [&](auto late_) __attribute__((always_inline)) {
    // s = 0, even phase: nothing to rotate in
    phase(H0{}, F_{}, T_{}, late_, 0, vbuf0, vbuf1, raw0, raw1);
    WB_WAIT_LDS_VM(2 * NPIECE);
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s + 1 < nstage; ++s) {
      float* rs = (s & 1) ? raw1 : raw0;                         // stage s (free for stage s+2 in the odd phase)
      const float* rn = (s & 1) ? raw0 : raw1;                   // stage s+1
      phase(H1{}, T_{}, T_{}, late_, 2 * s + 1, vbuf1, vbuf0, rn, rs);
      WB_WAIT_LDS_VM(63);
      __builtin_amdgcn_s_barrier();
      phase(H0{}, T_{}, T_{}, late_, 2 * s + 2, vbuf0, vbuf1, rn, rs);
      WB_WAIT_LDS_VM(2 * NPIECE);
      __builtin_amdgcn_s_barrier();
    }
    // the last odd phase: no half-stage behind it
    phase(H1{}, T_{}, F_{}, late_, 2 * nstage - 1, vbuf1, vbuf0, raw0, raw1);
#pragma unroll
    for (int i = 0; i < 6; ++i) mfma1(acc[3][1], A[1][1], BX, i);
  };
  // (LATE for the second wave of each SIMD -- `if (wave >= 4) run(T_{}); else run(F_{});` -- was measured: two copies of the loop
  // cost registers (256 + spills) and the forward got 5 % slower, not faster)
  run(F_{});
  WB_WAIT_LDS_VM(0);
  __builtin_amdgcn_s_barrier();

  // ---- output transform.  u[tt][jj'] = M[r][.] A (local); the four row waves of a channel group exchange u through LDS (the
  // input images' memory: every read of it is behind the last barrier) and wave r finishes block tile r & 1, accumulator
  // registers 8 (r >> 1) .. +7.  Two rounds (jj' = 0, 1), each: every wave writes all four destinations' eight registers
  // (compile-time register indices), barrier, every wave reads its four sources (own included) in row order.
  // (the bias values first: a load between the output stores would make every store wait for the one before it -- vmcnt counts
  // stores too, and s_waitcnt vmcnt(0) in front of each use was 8 HBM write latencies per tile)
  float bias8[2][4];
#pragma unroll
  for (int qd = 0; qd < 2; ++qd)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias8[qd][e] = a.bias[grp * 64 + cg * 32 + e + 8 * (2 * (r >> 1) + qd) + 4 * khl];
  float* exg = cg ? (float*)vbuf1 : (float*)vbuf0;              // [dst 4][src 4][quad 2][lane 64] x 16 B = 32 KiB per channel group
  f32x4 y[2][2][2];                                             // [jj'][ii][register quad]
  {
    const int tf = r & 1, h2 = r >> 1;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      if (jp) __syncthreads();                                  // round 0's reads are through
#pragma unroll
      for (int dst = 0; dst < 4; ++dst) {
        const int dt = dst & 1, dh = dst >> 1;                   // (compile-time after unrolling)
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          f32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int reg = 8 * dh + 4 * qd + e;
            u[e] = jp == 0 ? (acc[0][dt][reg] + acc[1][dt][reg]) + acc[2][dt][reg] : (acc[1][dt][reg] - acc[2][dt][reg]) - acc[3][dt][reg];
          }
          *(f32x4*)(exg + ((((dst * 4 + r) * 2 + qd) * 64 + lane) * 4)) = u;
        }
      }
      __syncthreads();
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        f32x4 u0 = *(const f32x4*)(exg + ((((r * 4 + 0) * 2 + qd) * 64 + lane) * 4));
        f32x4 u1 = *(const f32x4*)(exg + ((((r * 4 + 1) * 2 + qd) * 64 + lane) * 4));
        f32x4 u2 = *(const f32x4*)(exg + ((((r * 4 + 2) * 2 + qd) * 64 + lane) * 4));
        f32x4 u3 = *(const f32x4*)(exg + ((((r * 4 + 3) * 2 + qd) * 64 + lane) * 4));
        y[jp][0][qd] = (u0 + u1) + u2;
        y[jp][1][qd] = (u1 - u2) - u3;
      }
    }
    // bias, ReLU, stores: lane = block (bx, by) of tile tf, k half khl; register 8 h2 + 4 qd + e -> channel row
    const int bx = n31 & 15, by = 2 * tf + (n31 >> 4);
    const int x = tx0 + 2 * bx, yy = ty0 + 2 * by;
    const bool inx = x < a.W, iny = yy < a.H, inx1 = x + 1 < a.W, iny1 = yy + 1 < a.H;
#pragma unroll
    for (int qd = 0; qd < 2; ++qd)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = grp * 64 + cg * 32 + e + 8 * (2 * h2 + qd) + 4 * khl;
        const float bs = bias8[qd][e];
        float v00 = y[0][0][qd][e] + bs, v01 = y[1][0][qd][e] + bs, v10 = y[0][1][qd][e] + bs, v11 = y[1][1][qd][e] + bs;
        if (a.relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
        float* o = a.y + ((size_t)tb * a.cout + co) * vol + (size_t)tz * plane + (size_t)yy * a.W + x;
        if (inx & iny) {
          if (inx1) {
            W3_ST((float2*)o, make_float2(v00, v01));
            if (iny1) W3_ST((float2*)(o + a.W), make_float2(v10, v11));
          } else {
            o[0] = v00;
            if (iny1) o[a.W] = v10;
          }
        }
      }
  }
}

// false: nothing launched (shape the kernel does not take, or a launch too small to fill the chip): the caller falls back to
// the exact-fp32 Winograd / direct kernels
bool launch_conv_wbf(const ConvArgs& a, bool is3d, const unsigned* wq, hipStream_t s, int nprod = 6) {
  if (a.cin % WB_C != 0 || a.cin < 2 * WB_C || a.cout % 64 != 0) return false;
  if (a.D != 1 && !is3d) return false;
  if ((size_t)WB_C * a.D * a.H * a.W * 4 >= 0xf0000000ull) return false;        // a stage's 16 channel volumes: one 32-bit buffer range
  const int ntx = (a.W + 31) / 32, nty = (a.H + 7) / 8;
  const long nt = (long)ntx * nty * a.B * a.D * (a.cout / 64);
  if (nt < 512 || nt > 0x7fffffffl) return false;
  if (nprod == 3) {
    if (is3d) conv3_wbf_kernel<true, 3><<<(unsigned)nt, WB_NT, 0, s>>>(a, (const u32x4*)wq, ntx, nty);
    else conv3_wbf_kernel<false, 3><<<(unsigned)nt, WB_NT, 0, s>>>(a, (const u32x4*)wq, ntx, nty);
  } else if (is3d) conv3_wbf_kernel<true, 6><<<(unsigned)nt, WB_NT, 0, s>>>(a, (const u32x4*)wq, ntx, nty);
  else conv3_wbf_kernel<false, 6><<<(unsigned)nt, WB_NT, 0, s>>>(a, (const u32x4*)wq, ntx, nty);
  return true;
}
