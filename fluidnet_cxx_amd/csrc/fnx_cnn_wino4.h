// 3x3(x3) convolution in the Winograd F(4x4, 3x3) domain in (y, x), exact fp32 on v_mfma_f32_16x16x4_f32 (included by fnx_cnn.hip inside
// namespace fnx, after its DMA helpers).  Round 6; FNX_PRECISION_FP32_F4 = what FNX_PRECISION_FP32 runs for the 64- / 128-output-channel
// 3x3(x3) layers whose launch fills the chip (256^3 CNN step 92.2 -> 81.0 ms, 1024^2 2.29 -> 2.14 ms; FNX_PRECISION_FP32_F2 keeps them
// on conv3_wino3_kernel).  docs/history/r06_notes.md section 3 has the twenty-four versions that were measured and the cycle accounting:
// what paid was removing instructions from every wave's path, pinning the order of a stage with scheduling fences and not draining the
// pipeline between tiles -- never moving work between waves or phases.
//
//   Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A      d = 6x6 input patch of a 4x4 output block (Lavin & Gray, points 0, +-1, +-2, inf)
//
// 36 multiplies per 16 outputs instead of 144 (F(2x2): 64): the contraction over input channels is 36 independent GEMMs, one per
// position p of the 6x6 transform domain,
//   D_p[cout 16][block 16] += Wt_p[cout][k] * Xt_p[k][block],   k = four consecutive input channels = ONE v_mfma_f32_16x16x4_f32.
// A wave owns 16 output channels x 16 blocks (= 16 x 16 pixels) and ALL 36 positions: 36 x 4 = 144 accumulator registers, and the
// output transform A^T M A runs on registers alone -- no exchange between waves (conv3_wino3_kernel splits its 16 positions over
// two waves and swaps half of its outputs through LDS).  Workgroup = 8 waves = 64 output channels (4 groups of 16) x 32 blocks
// (2 groups of 16) = a 32 x 16-pixel tile; the four channel groups share the tile's transformed input, the two block groups share
// the stage's weights.  One PERSISTENT workgroup per CU (99 KB of LDS, two waves per SIMD at 256 registers) walks the tiles; the
// stages of its tiles form one stream (the last two stages of a tile request and transform the first chunks of the next).
//
// Stage = 4 input channels:
//   raw halo tile   4 x 18 x 40 floats (16-byte DMA pieces; W % 4 != 0: 4 x 18 x 34 by dwords), global -> LDS by DMA (buffer_load ... lds;
//                   zeros outside the image), two stages ahead
//   weights         the stage's nine taps per (output channel, k) pair, [9][4 groups][4 k][16] = 9 KB, global -> LDS by DMA two stages
//                   ahead; G g G^T is formed IN REGISTERS, one pair per lane -- the lane that feeds it to the MFMA (81 VALU per stage
//                   and wave instead of a 36 KB stream per stage and workgroup through L2 and LDS)
//   B^T d B         two passes through LDS, all threads: columns (raw -> tmp), barrier, rows (tmp -> xt [36][2 groups][4 k][16])
//   36 MFMAs        per wave; A from registers, B ONE conflict-free ds_read_b32 (64 consecutive floats per wave)
// Two barriers per stage; the transforms of stage s + 1 ride between the MFMA groups of stage s.  3D: the z taps are three times the
// stages (input plane z + dz - 1; a plane outside the grid is skipped).
//
// Numerics: the transforms are not exact in binary (G has 1/6, 1/24; B^T and A^T multiply by 2, 4, 5, 8): measured 2x the error of
// F(2x2) against an fp64 evaluation of the net, 0.07 of the tests' 1e-5 |ref|max (tools/wino_f4_error_probe.py).
#ifndef W4_V16
#define W4_V16 1
#endif
#ifndef W4_NWG
#define W4_NWG 256      // workgroups of a launch: one per CU (99 KB of LDS each)
#endif
#ifndef W4_NT
#define W4_NT 2         // cache policy: 2 = output stores non-temporal (3 % faster; non-temporal halo DMA measured 5 % slower: removed)
#endif
constexpr int W4C = 4;                       // input channels per stage
constexpr int W4_RAW = 4 * 18 * 34;          // 2448 floats of a stage's halo tile (4 bytes per lane: any W)
constexpr int W4_RAW16 = 4 * 18 * 10;        // 720 pieces of 16 bytes: rows of 40 floats, columns x0 - 4 .. x0 + 35 (W % 4 == 0)
constexpr int W4_RAWP = 3072;                // floats per buffer: 8 + 4 DMA instructions of 1 KiB (V16) / 5 of 256 B per wave
constexpr int W4_TP = 37;                    // tmp pitch per patch (odd: conflict-free row reads)
constexpr int W4_GST = 9 * 4 * 4 * 16;       // 2304 floats of a stage's taps

// blob (Cout, Cin, 3, 3) -> the nine taps as [Cin/4][Cout/64][9][4 groups of 16 cout][4 k][16]: what a wave reads as one float per lane and
// tap (its MFMA A-operand position: lane = k * 16 + cout % 16) to form G g G^T in registers (conv3_wino4_kernel)
// (3D, kd = 3: blob (Cout, Cin, 3, 3, 3); one such image per z tap, [dz][Cin/4][Cout/64]...: a stage = four channels of ONE z tap)
__global__ void pack_layer_wino4g_kernel(const float* __restrict__ w, float* __restrict__ pw, int cin, int cout, int kd) {
  const int n = cin * cout, ngrp = cout / 64;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int co = q / cin, ci = q - co * cin;
    for (int dz = 0; dz < kd; ++dz) {
      const size_t base = (((size_t)dz * (cin / 4) + ci / 4) * ngrp + co / 64) * (9 * 256) + (size_t)((co % 64) / 16) * 64 + (ci % 4) * 16 + co % 16;
      for (int t = 0; t < 9; ++t) pw[base + (size_t)t * 256] = w[((size_t)q * kd + dz) * 9 + t];
    }
  }
}

// G (6 x 3) applied to x0..x2: (x0/4, -(x0+x1+x2)/6, -(x0-x1+x2)/6, x0/24 + x1/12 + x2/6, x0/24 - x1/12 + x2/6, x2)
__device__ __forceinline__ void w4_g(float x0, float x1, float x2, float (&y)[6]) {
  const float e = (x0 + x2) * (-1.f / 6.f), f = fmaf(x0, 1.f / 24.f, x2 * (1.f / 6.f));
  y[0] = 0.25f * x0;
  y[1] = fmaf(x1, -1.f / 6.f, e);
  y[2] = fmaf(x1, 1.f / 6.f, e);
  y[3] = fmaf(x1, 1.f / 12.f, f);
  y[4] = fmaf(x1, -1.f / 12.f, f);
  y[5] = x2;
}

// B^T (6 x 6) applied to d0..d5
__device__ __forceinline__ void w4_bt(const float (&d)[6], float (&t)[6]) {
  const float a = fmaf(-4.f, d[2], d[4]), b = fmaf(-4.f, d[1], d[3]);        // d4 - 4 d2,  d3 - 4 d1
  const float c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
  t[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
  t[1] = a + b; t[2] = a - b;
  t[3] = c + e; t[4] = c - e;
  t[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}
// A^T (4 x 6) applied to m0..m5
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
  const float s1 = m1 + m2, s2 = m1 - m2, s3 = m3 + m4, s4 = m3 - m4;
  y[0] = (m0 + s1) + s3;
  y[1] = fmaf(2.f, s4, s2);
  y[2] = fmaf(4.f, s3, s1);
  y[3] = fmaf(8.f, s4, s2) + m5;
}

typedef float w4f4 __attribute__((ext_vector_type(4)));
template <int N> struct AIC4 { static constexpr int value = N; };

// IS3D: F(4x4) in (y, x), the three z taps as three times the stages (input plane z + dz - 1, the taps image of dz; a tap whose plane is
// outside the grid is skipped: zero padding)
// V16: the halo tile as 16-byte pieces (W % 4 == 0: a piece is inside the image or outside it) -- 12 DMA instructions per stage instead
// of 40 (each costs its wave an M0 write, a wait state and the issue).  (The dword form, five offsets per lane instead of two, spills 15
// registers to scratch since the stream runs across tiles: widths that are not a multiple of 4 only.)
template <bool IS3D, bool V16>
__global__ __launch_bounds__(512, 2) void conv3_wino4_kernel(ConvArgs a, const float* __restrict__ wt, int ntx, int nty, int ntiles) {
  __shared__ __attribute__((aligned(16))) float raw0[W4_RAWP];
  __shared__ __attribute__((aligned(16))) float raw1[W4_RAWP];
  __shared__ __attribute__((aligned(16))) float tmp[128 * W4_TP];
  __shared__ __attribute__((aligned(16))) float xt0[36 * 2 * 64];
  __shared__ __attribute__((aligned(16))) float xt1[36 * 2 * 64];
  __shared__ __attribute__((aligned(16))) float gw0[W4_GST];
  __shared__ __attribute__((aligned(16))) float gw1[W4_GST];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave & 3, ng = wave >> 2;
  const int ngrp = a.cout / 64, nch = a.cin / W4C;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;
  // PERSISTENT, ONE STREAM OF STAGES ACROSS TILES: a workgroup walks the tiles blockIdx.x, + gridDim.x, ...; the stages of all of them form
  // one sequence -- stage g multiplies chunk g, transforms chunk g + 1 and requests chunk g + 2 whatever tile those belong to -- so only
  // the first tile of a workgroup pays the prologue (DMA latency + two transform passes, ~4 us of the ~8 us a tile cost beside its stages);
  // between two tiles there is the output transform of the one and the clearing of the accumulators.
  struct Tile { int x0, y0, z, grp, b, dz_lo, nchunk; };
  auto decode = [&](int tile) __attribute__((always_inline)) {
    Tile T;
    int t = tile;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    T.z = 0;
    if (IS3D) { T.z = t % a.D; t /= a.D; }
    T.grp = t % ngrp; T.b = t / ngrp;
    T.dz_lo = IS3D && T.z == 0 ? 1 : 0;
    const int dz_hi = IS3D ? (T.z == a.D - 1 ? 2 : 3) : 1;
    T.nchunk = (dz_hi - T.dz_lo) * nch;                     // stages: (z tap, four input channels), z tap slowest
    T.x0 = tx * 32; T.y0 = ty * 16;
    return T;
  };

  // ---- the request iterator: the next chunk of the stream (tile f_tile, chunk f_c of it).  Halo-tile DMA: slot idx = q * 512 + tid of the
  // [4][18][34] tile (V16: piece idx of the [4][18][10] tile, q = 1 is waves 0-3's); out of the image (or beyond the tile) -> an offset the
  // range check refuses (zeros land in LDS)
  constexpr int NQ = V16 ? 2 : 5, RP = V16 ? 40 : 34, CO = V16 ? 3 : 0;     // DMA instructions per wave, row pitch, column of x0 - 1
  const unsigned stage_bytes = (unsigned)(((size_t)(W4C - 1) * vol + plane) * 4);
  const BufRsrcC wrs = make_rsrc_c(wt, 0x7ffffff0u);
  const unsigned wlane = (unsigned)(wave * 256 + lane * 4) * 4u;
  const ptrdiff_t raw_step = (ptrdiff_t)W4C * (ptrdiff_t)vol, raw_wrap = (ptrdiff_t)plane - (ptrdiff_t)(nch - 1) * raw_step;
  int raw_c = 0;
  unsigned uoff[NQ], w_next = 0;
  const float* raw_src = a.x;
  auto f_setup = [&](int f_tile) __attribute__((always_inline)) {     // the requests' state at chunk 0 of tile f_tile
    const Tile T = decode(f_tile);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int idx = q * 512 + tid;
      const int rowlen = V16 ? 10 : 34;
      const int c = idx / (18 * rowlen), rem = idx - c * (18 * rowlen);
      const int row = rem / rowlen, col = rem - row * rowlen;
      const int gy = T.y0 - 1 + row, gx = V16 ? T.x0 - 4 + 4 * col : T.x0 - 1 + col;
      const bool ok = (idx < (V16 ? W4_RAW16 : W4_RAW)) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
      uoff[q] = ok ? (unsigned)(((size_t)c * vol + (size_t)gy * a.W + gx) * 4) : 0xfffffff0u;
    }
    // the input pointer walks four channels per stage and, in 3D, from the last channel group of a z tap to the first of the next
    raw_src = a.x + (size_t)T.b * a.cin * vol + (size_t)(IS3D ? T.z + T.dz_lo - 1 : 0) * plane;
    raw_c = 0;
    w_next = (unsigned)(((size_t)(T.dz_lo * nch) * ngrp + T.grp) * W4_GST * 4);
  };
  f_setup(blockIdx.x);
  auto fetch_next = [&](float (&rawdst)[W4_RAWP], float (&wdst)[W4_GST]) __attribute__((always_inline)) {
    const BufRsrcC r = make_rsrc_c(raw_src, stage_bytes);
    if (IS3D && raw_c + 1 == nch) { raw_src += raw_wrap; raw_c = 0; } else { raw_src += raw_step; ++raw_c; }
    if (V16) {
      dma16_to_lds(r, (LdsF)&rawdst[0] + wave * 256, uoff[0], 0);
      if (wave < 4) dma16_to_lds(r, (LdsF)&rawdst[0] + 2048 + wave * 256, uoff[NQ - 1], 0);
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) dma4_to_lds(r, (LdsF)&rawdst[0] + q * 512 + wave * 64, uoff[q]);
    }
    const unsigned sb = w_next;
    w_next += (unsigned)(ngrp * W4_GST * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) {                       // 9 instructions of 1 KiB: waves 0-7, then wave 0 again
      const int wi = wave + 8 * q;
      if (wi < 9) dma16_to_lds(wrs, (LdsF)&wdst[0] + wi * 256, wlane, sb + (unsigned)(8 * 256 * q) * 4u);
    }
  };

  // ---- transform units: 768 per pass (128 patches x 6 columns / rows); thread tid takes unit tid and, below 256, unit 512 + tid
  // pass 1 (columns): unit = patch * 6 + j           pass 2 (rows): unit = i * 128 + patch (blocks fastest: conflict-free xt stores)
  int p1_rd[2], p1_wr[2], p2_rd[2], p2_wr[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int u = h * 512 + tid;
    {
      const int patch = u / 6, j = u - patch * 6;
      const int c = patch >> 5, n = patch & 31, bx = n & 7, by = n >> 3;
      p1_rd[h] = (c * 18 + 4 * by) * RP + 4 * bx + j + CO;
      p1_wr[h] = patch * W4_TP + j;
    }
    {
      const int i = u >> 7, patch = u & 127;
      const int c = patch >> 5, n = patch & 31;
      p2_rd[h] = patch * W4_TP + i * 6;
      p2_wr[h] = ((i * 6) * 2 + (n >> 4)) * 64 + c * 16 + (n & 15);
    }
  }
  // The transforms of stage s + 1 ride in the MFMA stream of stage s: each pass is cut into its LDS loads (issued ahead of a group of
  // nine MFMAs, 288 cycles) and its arithmetic + stores (behind it).
  // (unit 0 of a thread is cut into loads ahead of nine MFMAs and arithmetic + stores behind them; the second unit of the lower 256
  //  threads is done in one go behind it: six registers instead of twelve next to 144 accumulators)
  float pd[6];
  auto p1_load = [&](const float (&rawsrc)[W4_RAWP]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) pd[i] = rawsrc[p1_rd[0] + i * RP];
  };
  auto p1_store = [&](const float (&rawsrc)[W4_RAWP]) __attribute__((always_inline)) {
    float tt[6];
    w4_bt(pd, tt);
#pragma unroll
    for (int i = 0; i < 6; ++i) tmp[p1_wr[0] + i * 6] = tt[i];
    if (IS3D ? tid < 256 : wave < 4) {
      float d[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = rawsrc[p1_rd[1] + i * RP];
      w4_bt(d, tt);
#pragma unroll
      for (int i = 0; i < 6; ++i) tmp[p1_wr[1] + i * 6] = tt[i];
    }
  };
  auto p2_load = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 6; ++j) pd[j] = tmp[p2_rd[0] + j];
  };
  auto p2_store = [&](float (&xtdst)[36 * 2 * 64]) __attribute__((always_inline)) {
    float v[6];
    w4_bt(pd, v);
#pragma unroll
    for (int j = 0; j < 6; ++j) xtdst[p2_wr[0] + j * 128] = v[j];
    if (IS3D ? tid < 256 : wave < 4) {
      float d[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = tmp[p2_rd[1] + j];
      w4_bt(d, v);
#pragma unroll
      for (int j = 0; j < 6; ++j) xtdst[p2_wr[1] + j * 128] = v[j];
    }
  };

  w4f4 acc[36];
  // 36 MFMAs per stage in four groups of nine positions; a group's operands are read ahead of the MFMAs of the group before (left to
  // itself the compiler reads two positions, waits for the LDS, issues two MFMAs).
  // A operands: G g G^T of this wave's (output channel, k) pairs -- one pair per lane, lane = k * 16 + cout % 16, exactly the lane that
  // feeds it to the MFMA -- formed in registers from the nine taps (90 VALU per stage and wave) instead of streaming the 4x larger
  // transformed image through L2 and LDS (36 KB per stage and workgroup: the first two versions waited 0.47 ms per forward for it)
  // (half of the 36 at a time -- rows 0-2 ahead of the first eighteen MFMAs, rows 3-5 ahead of the last: 144 accumulators leave no room
  //  for all of them)
  float gv[9], U[18], bv[2][9];
  auto g_load = [&](const float (&gsrc)[W4_GST]) __attribute__((always_inline)) {
    const float* gp = &gsrc[cg * 64 + lane];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) gv[t9] = gp[t9 * 256];
  };
  auto g_xform = [&](auto hsel) __attribute__((always_inline)) {
    constexpr int HALF = decltype(hsel)::value;
    float tt[3][3];                                       // (G g)[3 HALF + i][s]
#pragma unroll
    for (int sidx = 0; sidx < 3; ++sidx) {
      const float x0 = gv[sidx], x1 = gv[3 + sidx], x2 = gv[6 + sidx];
      if (HALF == 0) {
        const float e = (x0 + x2) * (-1.f / 6.f);
        tt[0][sidx] = 0.25f * x0; tt[1][sidx] = fmaf(x1, -1.f / 6.f, e); tt[2][sidx] = fmaf(x1, 1.f / 6.f, e);
      } else {
        const float f = fmaf(x0, 1.f / 24.f, x2 * (1.f / 6.f));
        tt[0][sidx] = fmaf(x1, 1.f / 12.f, f); tt[1][sidx] = fmaf(x1, -1.f / 12.f, f); tt[2][sidx] = x2;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float y[6];
      w4_g(tt[i][0], tt[i][1], tt[i][2], y);              // (G g) G^T: row 3 HALF + i
#pragma unroll
      for (int j = 0; j < 6; ++j) U[i * 6 + j] = y[j];
    }
  };
  auto op_load = [&](int g, const float (&xsrc)[36 * 2 * 64]) __attribute__((always_inline)) {
    const float* xbp = &xsrc[ng * 64 + lane];
#pragma unroll
    for (int q = 0; q < 9; ++q) bv[g & 1][q] = xbp[(g * 9 + q) * 128];
  };
  auto mfma9 = [&](int g) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[g * 9 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[(g & 1) * 9 + q], bv[g & 1][q], acc[g * 9 + q], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

#ifndef W4_ABL
#define W4_ABL 0        // timing ablations (results are wrong): 1 no MFMAs, 2 no transforms, 4 no DMA
#endif
  // Stage s (steady state): rawN = halo tile of stage s + 1 (landed), rawF = free (tile s was transformed during stage s - 1): receives
  // tile s + 2; xtC = this stage's transformed input, xtN = the next stage's; the taps of stage s are in registers (gv, read at the end of
  // stage s - 1), gN = the taps of stage s + 1 (landed; read into registers at the end of this stage), gF = the slot the taps of stage s
  // came from: receives the taps of stage s + 2.
  //   B1   everybody's DMA of the previous top has landed; xtC is complete; nobody reads rawF, xtN, tmp, gF any more
  //   DMA  halo tile s + 2 -> rawF, taps s + 2 -> gF
  //   operands of the first nine MFMAs and the loads of pass 1 of stage s + 1 (rawN); G g G^T rows 0-2 in registers while they fly;
  //   nine MFMAs x 2 with the arithmetic and stores of pass 1 (-> tmp) between them;  B2 (tmp complete);  the loads of pass 2;
  //   G g G^T rows 3-5; nine MFMAs x 2 with pass 2's arithmetic and stores (-> xtN) between them; the taps of stage s + 1 into registers
  // The order is pinned with scheduling fences: left alone the compiler regroups the loads, transforms and MFMAs of the (branch-free)
  // stage, and every variant of the order that was measured is within +-2 % of this one -- on whole replayed steps: this order with
  // the fences 1024^2 2.250 -> 2.17 ms, 256^3 82.6 -> 81.5 ms (docs/history/r06_notes.md, versions 17-20).
  // (KIND 0: a stage of a tile's loop: requests chunk s + 2 of the tile, no condition in it; 1: one of the tile's last two stages: requests
  //  chunk 0 / 1 of the workgroup's NEXT tile if there is one.  The last stage of the workgroup's last tile transforms a stale tile and
  //  reads stale taps: LDS only, nobody uses the results)
  auto stage = [&](auto ksel, bool has_next, float (&rawN)[W4_RAWP], float (&rawF)[W4_RAWP], float (&xtC)[36 * 2 * 64], float (&xtN)[36 * 2 * 64],
                   float (&gN)[W4_GST], float (&gF)[W4_GST]) __attribute__((always_inline)) {
    constexpr int KIND = decltype(ksel)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!(W4_ABL & 4) && (KIND == 0 || has_next)) fetch_next(rawF, gF);
    constexpr bool nxt = !(W4_ABL & 2);
    constexpr bool mm = !(W4_ABL & 1);
#define W4_FENCE __builtin_amdgcn_sched_barrier(0)
    W4_FENCE;
    if (mm) op_load(0, xtC);
    if (nxt) p1_load(rawN);
    W4_FENCE;
    if (mm) g_xform(AIC4<0>{});
    W4_FENCE;
    if (mm) { op_load(1, xtC); mfma9(0); }
    if (nxt) p1_store(rawN);
    W4_FENCE;
    if (mm) { op_load(2, xtC); mfma9(1); }
    __syncthreads();
    if (nxt) p2_load();
    W4_FENCE;
    if (mm) { g_xform(AIC4<1>{}); op_load(3, xtC); mfma9(2); }
    if (nxt) p2_store(xtN);
    W4_FENCE;
    if (mm) mfma9(3);
#undef W4_FENCE
    g_load(gN);
  };
  // prologue (the workgroup's first tile only): chunks 0 and 1 in flight; tile 0 transformed into xt0; the taps of stage 0 in registers
  fetch_next(raw0, gw0);
  fetch_next(raw1, gw1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  p1_load(raw0); p1_store(raw0);
  g_load(gw0);
  __syncthreads();
  p2_load(); p2_store(xt0);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const Tile T = decode(tile);
  const int x0 = T.x0, y0 = T.y0, z = T.z, grp = T.grp, b = T.b;
#pragma unroll
  for (int p = 0; p < 36; ++p) acc[p] = (w4f4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s + 2 < T.nchunk; s += 2) {                // (cin % 16 == 0: an even number of stages per tile, at least four)
    stage(AIC4<0>{}, true, raw1, raw0, xt0, xt1, gw1, gw0);
    stage(AIC4<0>{}, true, raw0, raw1, xt1, xt0, gw0, gw1);
  }
  const bool has_next = tile + (int)gridDim.x < ntiles;
  if (has_next) f_setup(tile + gridDim.x);                   // (this tile's last request went out two stages ago)
  stage(AIC4<1>{}, has_next, raw1, raw0, xt0, xt1, gw1, gw0);
  stage(AIC4<1>{}, has_next, raw0, raw1, xt1, xt0, gw0, gw1);

  // ---- epilogue: A^T M A per accumulator register (an output channel), bias, ReLU, 4 x 4 pixels per block
  const int n = ng * 16 + (lane & 15), bx = n & 7, by = n >> 3;
  const int px = x0 + 4 * bx, py = y0 + 4 * by;
  const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = grp * 64 + cg * 16 + 4 * (lane >> 4) + r;
    float Tm[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
      w4_at(acc[i * 6 + 0][r], acc[i * 6 + 1][r], acc[i * 6 + 2][r], acc[i * 6 + 3][r], acc[i * 6 + 4][r], acc[i * 6 + 5][r], Tm[i]);
    const float bias = a.bias[co];
    float* yo = a.y + ((size_t)b * a.cout + co) * vol + (size_t)z * plane;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {                     // output column bb of the block: A^T along i
      float y[4];
      w4_at(Tm[0][bb], Tm[1][bb], Tm[2][bb], Tm[3][bb], Tm[4][bb], Tm[5][bb], y);
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) Tm[aa][bb] = fmaxf(y[aa] + bias, lo);      // (reuse Tm[0..3][bb] for the outputs of row aa)
    }
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
      const int yy = py + aa;
      if (yy >= a.H) continue;
      float* row = yo + (size_t)yy * a.W + px;
      if (px + 3 < a.W && ((a.W & 3) == 0)) {
        const w4f4 v = {Tm[aa][0], Tm[aa][1], Tm[aa][2], Tm[aa][3]};
        if (W4_NT & 2) __builtin_nontemporal_store(v, (w4f4*)row); else *(w4f4*)row = v;
      }
      else {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
          if (px + bb < a.W) row[bb] = Tm[aa][bb];
      }
    }
  }
  }
}

// false: nothing launched (the caller takes the F(2x2) kernel)
bool launch_conv_wino4(const ConvArgs& a, const float* wt4, bool is3d, hipStream_t s) {
  if ((!is3d && a.D != 1) || a.cin % 16 != 0 || a.cout % 64 != 0) return false;
  if ((size_t)W4C * a.D * a.H * a.W * 4 >= 0xfffffff0u) return false;
  const int ntx = (a.W + 31) / 32, nty = (a.H + 15) / 16;
  const long nt = (long)ntx * nty * a.D * a.B * (a.cout / 64);
  if (nt < 256 || nt > 0x7fffffffl) return false;        // a launch that does not fill the chip stays on the F(2x2) / direct kernels
  const bool v16 = (a.W & 3) == 0 && W4_V16;
  const unsigned nwg = (unsigned)(nt < W4_NWG ? nt : W4_NWG);        // persistent: one workgroup per CU walks the tiles
  const int nti = (int)nt;
  if (is3d) { if (v16) conv3_wino4_kernel<true, true><<<nwg, 512, 0, s>>>(a, wt4, ntx, nty, nti); else conv3_wino4_kernel<true, false><<<nwg, 512, 0, s>>>(a, wt4, ntx, nty, nti); }
  else { if (v16) conv3_wino4_kernel<false, true><<<nwg, 512, 0, s>>>(a, wt4, ntx, nty, nti); else conv3_wino4_kernel<false, false><<<nwg, 512, 0, s>>>(a, wt4, ntx, nty, nti); }
  return true;
}
