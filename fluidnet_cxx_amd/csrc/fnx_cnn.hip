// MultiScale pressure-net forward (lib/multi_scale_net.py:118-127, lib/model.py:76-227) for gfx950.
//
// The 32/64/128-channel 3x3(x3) layers (95 % of the FLOPs) run as implicit GEMM on the matrix cores in exact fp32
// (conv3_mfma_kernel, 32x32x2 MFMA); the 5x5(x5) layers 3->32 and 32->8 on the 16x16x4 MFMA (conv5_mfma16_kernel);
// the remaining thin layers (2->32 and 32->1 3x3, the final 1x1) are bandwidth-shaped and use a direct kernel
// (a strip of output pixels x CO_T output channels per thread, weights through the scalar cache).
#include "fnx_cnn.h"
#include <assert.h>
#include <stdlib.h>
#include <type_traits>
#include "fnx_kernels.h"
#include "../../include/fluidnet_hip.h"

namespace fnx {
// (every error return sets the thread's message: the Python layer raises fnx_last_error(), which would otherwise be an earlier call's text)
inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? FNX_OK : set_error(FNX_EHIP, "HIP error in a CNN launch: %s (a grid the net cannot take -- 3D: D a multiple of 4?)", hipGetErrorString(e));
}


namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int co_tile(int cout) { return cout >= 16 ? 16 : cout; }

struct PackedLayer { size_t w_off, b_off; };   // float offsets into the packed buffer

// 5x5(x5) layers (Cin 3 or 32, Cout 32 or 8) run on v_mfma_f32_16x16x4_f32: Cin is padded to a multiple of 4 (one
// k-step) and Cout to a multiple of 16 (one M block) with zero weights.
inline bool mfma16_layer(const ConvLayer& L) { return L.k == 5 && (L.cin % 8 == 0 || L.cin <= 4) && L.cout <= 32; }
inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
// Cout <= 8 (and a Cin the 8-channel stage fits): two x-adjacent pixels share one 16-row M block (conv5_mfma16_kernel PAIR)
inline bool pair_layer(int cin, int cout) { return cout <= 8 && cin % 8 == 0; }
inline bool mfma_layer(const ConvLayer& L) { return L.k == 3 && L.cin % 16 == 0 && L.cout % 32 == 0; }
// the 3x3(x3) MFMA layers also run in the Winograd domain (conv3_wino3_kernel): their transformed weights G g G^T
// ([dz][16][Cin][Cout], then the same values in the kernel's stage-contiguous order) follow the [tap][Cin][Cout] image
inline bool wino_layer(const ConvLayer& L, bool is3d) { (void)is3d; return mfma_layer(L); }
// FNX_PRECISION_BF16X6 (conv3_wbf_kernel): the wino layers with 64 output channels per workgroup; their transformed weights
// cut into three bf16 pieces, in the kernel's MFMA operand layout (1.5x the fp32 image), follow the two fp32 images
inline bool wbf_layer(const ConvLayer& L, bool is3d) { return wino_layer(L, is3d) && L.cin % 16 == 0 && L.cout % 64 == 0; }
// FNX_PRECISION_FP32 (3D) / _FP32_F4 (conv3_wino4_kernel, fnx_cnn_wino4.h): the wino layers with 64 output channels per workgroup; their nine
// taps in the kernel's lane order ([Cin/4][Cout/64][9][4][4][16], 9 Cin Cout floats: G g G^T is formed in registers) follow the bf16 image
inline bool wino4_layer_(const ConvLayer& L, bool is3d) { (void)is3d; return L.k == 3 && L.cin % 16 == 0 && L.cout % 64 == 0; }
inline size_t wino4_offset(const ConvLayer& L, bool is3d) {      // floats from the layer's w_off to its F(4x4) image
  const size_t nwino = (size_t)16 * (is3d ? 3 : 1) * L.cin * L.cout;
  return layer_weight_floats(L, is3d) + 2 * nwino + (wbf_layer(L, is3d) ? nwino * 3 / 2 : 0);
}
inline size_t packed_weight_floats(const ConvLayer& L, bool is3d) {
  if (wino_layer(L, is3d))
    return layer_weight_floats(L, is3d) + 2 * (size_t)16 * (is3d ? 3 : 1) * L.cin * L.cout +
           (wbf_layer(L, is3d) ? (size_t)16 * (is3d ? 3 : 1) * L.cin * L.cout * 3 / 2 : 0) +
           (wino4_layer_(L, is3d) ? (size_t)(is3d ? 27 : 9) * L.cin * L.cout : 0);
  if (mfma16_layer(L) && pair_layer(L.cin, L.cout)) return (size_t)(is3d ? 5 : 1) * 30 * pad_to(L.cin, 4) * 16;
  if (mfma16_layer(L)) return (size_t)layer_taps(L, is3d) * pad_to(L.cin, 4) * pad_to(L.cout, 16);
  return layer_weight_floats(L, is3d);
}

PackedLayer packed_layer(int l, bool is3d) {
  size_t off = 0;
  PackedLayer r{0, 0};
  for (int i = 0; i <= l; ++i) {
    r.w_off = off;
    off += packed_weight_floats(LAYERS[i], is3d);
    r.b_off = off;
    off += LAYERS[i].cout;
    off = (off + 63) & ~(size_t)63;
  }
  return r;
}

// blob: (Cout,Cin,taps) -> packed [taps][Cin][Cout]   (implicit-GEMM layers: the A operand of the MFMA reads 32
// consecutive output channels per lane half)
__global__ void pack_layer_mfma_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                       float* __restrict__ pw, float* __restrict__ pb, int cin, int cout, int taps) {
  const int n = cin * cout * taps;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int co = q / (cin * taps);
    const int r = q - co * cin * taps;
    const int ci = r / taps, t = r - ci * taps;
    pw[((size_t)t * cin + ci) * cout + co] = w[q];
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cout; q += gridDim.x * blockDim.x) pb[q] = bias[q];
}

// blob: (Cout,Cin,taps) -> packed [taps][Cin padded to 4][Cout padded to 16], zeros in the padding
__global__ void pack_layer_mfma16_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                         float* __restrict__ pw, float* __restrict__ pb, int cin, int cout, int taps,
                                         int cin_pad, int cout_pad) {
  const int n = taps * cin_pad * cout_pad;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int co = q % cout_pad;
    const int r = q / cout_pad;
    const int ci = r % cin_pad, t = r / cin_pad;
    pw[q] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * taps + t] : 0.f;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cout; q += gridDim.x * blockDim.x) pb[q] = bias[q];
}

// Cin of 2 or 3 (the first layer of a scale): the MFMA K runs over (tap, channel) pairs of a z plane, k = tap * Cin + channel,
// padded to a multiple of 4 -- 13 / 19 K-groups of four per plane instead of 25 with one or two empty channels each
// (conv5_mfma16_kernel<4, 2, ., false, KPC>).  0: the layer does not take that form.
inline int kpack_cin(int cin, int cout) { return (cin == 2 || cin == 3) && cout > 16 && cout <= 32 ? cin : 0; }   // (Cout padded to 32: MB = 2)
// blob: (Cout,Cin 2|3,taps 5x5[x5]) -> packed [dz][k = (r*5+s)*Cin + ci, padded to 4][Cout padded to 32], zeros in the padding
__global__ void pack_layer_kpack_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ pw,
                                        float* __restrict__ pb, int cin, int cout, int kd, int cout_pad) {
  const int krows = (25 * cin + 3) / 4 * 4;
  const int n = kd * krows * cout_pad;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int co = q % cout_pad, r1 = q / cout_pad;
    const int k = r1 % krows, dz = r1 / krows;
    const int tap = k / cin, ci = k - tap * cin;
    pw[q] = (co < cout && tap < 25) ? w[((size_t)co * cin + ci) * (kd * 25) + dz * 25 + tap] : 0.f;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cout; q += gridDim.x * blockDim.x) pb[q] = bias[q];
}

// blob: (Cout<=8,Cin,taps 5x5[x5]) -> packed [dz][r][wx 0..5][Cin padded to 4][dx*8 + cout]: the weight of tap (r, wx-dx)
// for output pixel dx of a pair, zero where wx-dx is not one of the 5 taps
__global__ void pack_layer_pair_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                       float* __restrict__ pw, float* __restrict__ pb, int cin, int cout, int kd,
                                       int cin_pad) {
  const int n = kd * 30 * cin_pad * 16;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int m = q % 16, r1 = q / 16;
    const int ci = r1 % cin_pad, r2 = r1 / cin_pad;
    const int wx = r2 % 6, r3 = r2 / 6;
    const int r = r3 % 5, dz = r3 / 5;
    const int dx = m >> 3, co = m & 7, sx = wx - dx;
    const int taps = kd * 25;
    pw[q] = (co < cout && ci < cin && sx >= 0 && sx < 5) ? w[((size_t)co * cin + ci) * taps + (dz * 5 + r) * 5 + sx] : 0.f;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cout; q += gridDim.x * blockDim.x) pb[q] = bias[q];
}

// blob: (Cout,Cin,taps) -> packed [Cout/CO_T][Cin][taps][CO_T]
__global__ void pack_layer_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ pw,
                                  float* __restrict__ pb, int cin, int cout, int taps, int cot) {
  const int n = cin * cout * taps;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const int co = q / (cin * taps);
    const int r = q - co * cin * taps;
    const int ci = r / taps, t = r - ci * taps;
    const int grp = co / cot, tt = co - grp * cot;
    pw[(((size_t)grp * cin + ci) * taps + t) * cot + tt] = w[q];
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cout; q += gridDim.x * blockDim.x) pb[q] = bias[q];
}

struct ConvArgs {
  const float* x; float* y; const float* w; const float* bias;
  int B, cin, cout, D, H, W, relu, groups;
  // Optional fused tail (conv5_mfma16_kernel PAIR only): a 1x1(x1) convolution cout -> 1 applied in the epilogue -- y then has ONE
  // channel: y = tail_b[0] + sum_c tail_w[c] * act(conv)[c].  multi_scale_net.py:116: the net's last 5x5 layer (32 -> 8) and its
  // final 1x1 (8 -> 1); the 8-channel tensor between them is never written.
  const float* tail_w; const float* tail_b;
};

// Thin layers (Cin 2-3 or Cout 1-8): direct convolution.  A thread owns a vertical strip of RY output pixels x CO_T
// output channels, so each input row it loads (coalesced along x across the wave) feeds up to KS output rows:
// (RY+KS-1)*KS loads per RY*KS*KS taps instead of one load per tap.
template <int KS, int CO_T, bool IS3D, int RY>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  constexpr int KD = IS3D ? KS : 1, PAD = KS / 2, PD = IS3D ? PAD : 0, TAPS = KD * KS * KS;
  const int i = blockIdx.x * 64 + threadIdx.x, j0 = (blockIdx.y * 4 + threadIdx.y) * RY;
  int z = blockIdx.z;
  const int k = z % a.D; z /= a.D;
  const int grp = z % a.groups; const int b = z / a.groups;
  if (i >= a.W || j0 >= a.H) return;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;
  float acc[RY][CO_T];
#pragma unroll
  for (int t = 0; t < CO_T; ++t) {
    const float bv = a.bias[grp * CO_T + t];
#pragma unroll
    for (int ry = 0; ry < RY; ++ry) acc[ry][t] = bv;
  }
  const float* wg = a.w + (size_t)grp * a.cin * TAPS * CO_T;
  const float* xb = a.x + (size_t)b * a.cin * vol;
  for (int ci = 0; ci < a.cin; ++ci) {
    const float* xc = xb + (size_t)ci * vol;
    const float* wc = wg + (size_t)ci * TAPS * CO_T;
#pragma unroll
    for (int dz = 0; dz < KD; ++dz) {
      const int zz = k + dz - PD;
      const bool zin = (zz >= 0) & (zz < a.D);
#pragma unroll
      for (int row = 0; row < RY + KS - 1; ++row) {        // input row j0 - PAD + row
        const int yy = j0 + row - PAD;
        const bool yin = zin & (yy >= 0) & (yy < a.H);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int xx = i + s - PAD;
          const bool in = yin & (xx >= 0) & (xx < a.W);
          const float v = in ? xc[(size_t)(zin ? zz : 0) * plane + (size_t)(yin ? yy : 0) * a.W + (in ? xx : 0)] : 0.f;
#pragma unroll
          for (int ry = 0; ry < RY; ++ry) {
            const int r = row - ry;                        // tap row for output row ry
            if (r >= 0 && r < KS) {
              const float* wt = wc + ((dz * KS + r) * KS + s) * CO_T;
#pragma unroll
              for (int t = 0; t < CO_T; ++t) acc[ry][t] = fmaf(v, wt[t], acc[ry][t]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int ry = 0; ry < RY; ++ry) {
    if (j0 + ry >= a.H) break;
    float* yb = a.y + ((size_t)b * a.cout + grp * CO_T) * vol + (size_t)k * plane + (size_t)(j0 + ry) * a.W + i;
#pragma unroll
    for (int t = 0; t < CO_T; ++t) {
      float v = acc[ry][t];
      if (a.relu) v = fmaxf(v, 0.f);
      yb[(size_t)t * vol] = v;
    }
  }
}

template <int KS, bool IS3D>
void launch_conv_k(const ConvArgs& a, hipStream_t s) {
  constexpr int RY = (KS == 1) ? 1 : 4;
  const dim3 grid((a.W + 63) / 64, (a.H + 4 * RY - 1) / (4 * RY), a.B * a.groups * a.D), block(64, 4);
  const int cot = co_tile(a.cout);
  if (cot == 16) conv_direct_kernel<KS, 16, IS3D, RY><<<grid, block, 0, s>>>(a);
  else if (cot == 8) conv_direct_kernel<KS, 8, IS3D, RY><<<grid, block, 0, s>>>(a);
  else conv_direct_kernel<KS, 1, IS3D, RY><<<grid, block, 0, s>>>(a);
}

// 3x3(x3) convolution to ONE output channel (the towers' last layers, 32 -> 1, at half and quarter resolution).  In the strip
// kernel above a thread walks all Cin channels of its 4 output rows alone: 4 chains of 9 Cin dependent FMAs on launches of a few
// dozen workgroups (256^2: 25.6 us for 37 MFLOP).  Here the block's four waves split the INPUT CHANNELS: wave q accumulates
// channels q, q + 4, ... of the block's 4 output rows (each input row it loads feeds up to 3 output rows), the partial sums meet in
// LDS and wave r finishes output row r: sums of the four partials in wave order + bias.  4x the workgroups, chains a quarter as long.
template <bool IS3D>
__global__ __launch_bounds__(256) void conv3_to1_kernel(ConvArgs a) {
  constexpr int KS = 3, KD = IS3D ? KS : 1, PAD = 1, PD = IS3D ? 1 : 0, TAPS = KD * KS * KS, RY = 4;
  __shared__ float part[4][RY][64];
  const int lane = threadIdx.x, q = threadIdx.y;
  const int i = blockIdx.x * 64 + lane, j0 = blockIdx.y * RY;
  int z = blockIdx.z;
  const int k = z % a.D; const int b = z / a.D;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;
  float acc[RY] = {0.f, 0.f, 0.f, 0.f};
  const float* xb = a.x + (size_t)b * a.cin * vol;
  for (int ci = q; ci < a.cin; ci += 4) {
    const float* xc = xb + (size_t)ci * vol;
    const float* wc = a.w + (size_t)ci * TAPS;              // packed [1][Cin][taps][1]
#pragma unroll
    for (int dz = 0; dz < KD; ++dz) {
      const int zz = k + dz - PD;
      const bool zin = (zz >= 0) & (zz < a.D);
#pragma unroll
      for (int row = 0; row < RY + KS - 1; ++row) {
        const int yy = j0 + row - PAD;
        const bool yin = zin & (yy >= 0) & (yy < a.H);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          const int xx = i + t - PAD;
          const bool in = yin & (xx >= 0) & (xx < a.W);
          const float v = in ? xc[(size_t)(zin ? zz : 0) * plane + (size_t)(yin ? yy : 0) * a.W + (in ? xx : 0)] : 0.f;
#pragma unroll
          for (int ry = 0; ry < RY; ++ry) {
            const int r = row - ry;
            if (r >= 0 && r < KS) acc[ry] = fmaf(v, wc[(dz * KS + r) * KS + t], acc[ry]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int ry = 0; ry < RY; ++ry) part[q][ry][lane] = acc[ry];
  __syncthreads();
  const int j = j0 + q;                                     // wave q finishes output row q
  if (i < a.W && j < a.H) {
    float v = ((part[0][q][lane] + part[1][q][lane]) + part[2][q][lane]) + part[3][q][lane];
    v += a.bias[0];
    if (a.relu) v = fmaxf(v, 0.f);
    a.y[(size_t)b * vol + (size_t)k * plane + (size_t)j * a.W + i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// Implicit-GEMM 3x3(x3) convolution on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//   D[cout 32][pixel 32] += A[cout][k] * B[k][pixel],  k = two consecutive input channels of one tap
//   A: lane -> W[tap][c + (lane>>5)][cout0 + (lane&31)]      (one coalesced 256-B global/L1 read per wave)
//   B: lane -> X[c + (lane>>5)][y + r - 1][x0 + (lane&31) + s - 1]  from the LDS halo tile (conflict-free rows)
//   D: lane holds pixel (lane&31) and 16 output channels -> every store is a 128-B row segment
// Workgroup = 4 waves stacked in y; wave tile = 32 px x PR rows x (CB*32) output channels (PR*CB accumulators of
// 16 VGPRs).  Input channels are staged through LDS 8 at a time ((4PR+2) x 34 halo rows).  3D: the z taps are an
// outer loop over the three input planes.
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int MF_CHUNK = 8;       // input channels per LDS stage
constexpr int MF_COLS = 34;       // 32 + halo

// CH = input channels per LDS stage; WPS = waves per SIMD the register budget is sized for (2: 256 VGPRs, 3: 168)
typedef __amdgpu_buffer_rsrc_t BufRsrcC;
__device__ __forceinline__ BufRsrcC make_rsrc_c(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// 16 bytes per lane global -> LDS (buffer_load_dwordx4 ... lds: the wave's 1 KiB lands at `dst`, wave-uniform).  In a __device__
// function of its own: called from a __global__ template directly, the 16-byte form fails the builtin's size check in the HOST
// pass (no gfx950 there), silently, and the kernel's host stub is then never emitted (undefined symbol at load time).
typedef __attribute__((address_space(3))) float* LdsF;
__device__ __forceinline__ void dma4_to_lds(const BufRsrcC r, LdsF dst, unsigned voff) {      // 4 bytes per lane: 64 consecutive floats at `dst`
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 4, voff, 0, 0, 0);
}
__device__ __forceinline__ void dma16_to_lds(const BufRsrcC r, LdsF dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

template <int CB, int PR, bool IS3D, int CH = MF_CHUNK, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void conv3_mfma_kernel(ConvArgs a) {
  constexpr int ROWS = 4 * PR + 2;
  constexpr int KD = IS3D ? 3 : 1;
  constexpr int RW = CB * 32;                    // output channels (floats) per weight row
  constexpr int WROWS = 9 * CH;            // (tap, cin) rows per stage
  constexpr int LPR = RW / 4;                    // lanes per row at 16 B per lane
  constexpr int RPI = 64 / LPR;                  // rows per wave-wide LDS DMA instruction
  constexpr int NWI = (WROWS + RPI - 1) / RPI;   // wave-instructions per stage
  __shared__ __attribute__((aligned(16))) float tile2[2][CH * ROWS * MF_COLS];     // halo tile, double-buffered
  // weights of the stage, [tap][cin][cout], double-buffered in two SEPARATE arrays: the compiler then knows that the DMA
  // into one cannot alias the reads of the other and does not drain vmcnt (the next stage's loads, just issued) in
  // front of the MFMAs
  __shared__ __attribute__((aligned(16))) float wbuf0[NWI * 256];
  __shared__ __attribute__((aligned(16))) float wbuf1[NWI * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * (4 * PR);
  int zb = blockIdx.z;
  const int ngrp = a.cout / (CB * 32);
  const int grp = zb % ngrp; zb /= ngrp;
  const int z = zb % a.D; const int b = zb / a.D;
  const int cout0 = grp * CB * 32;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;

  f32x16 acc[PR][CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = a.bias[cout0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
      for (int pr = 0; pr < PR; ++pr) acc[pr][cb][r] = bv;
    }
  }

  const float* xb = a.x + (size_t)b * a.cin * vol;
  // Per-thread staging slots: element idx = threadIdx.x + 256*t of the [CH][ROWS][34] halo tile.  The global
  // offset (relative to the chunk's first channel and the z plane) and the in-image predicate never change, so they
  // are computed once; each stage is then NLD independent loads issued back to back (clamped address + select).
  constexpr int NEL = CH * ROWS * MF_COLS;
  constexpr int NLD = (NEL + 255) / 256;
  // Loads go through a buffer resource spanning the stage's CH channel volumes: an out-of-image element gets an
  // out-of-range offset and the hardware returns 0 -- no select between the load and the LDS store (the compiler hoists
  // such a select to the load and drains vmcnt there, in front of the MFMAs the load is supposed to hide under).
  unsigned uoff[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int idx = threadIdx.x + 256 * t;
    const int cc = idx / (ROWS * MF_COLS);
    const int rem = idx - cc * ROWS * MF_COLS;
    const int row = rem / MF_COLS, col = rem - row * MF_COLS;
    const int gx = x0 - 1 + col, gy = y0 - 1 + row;
    const bool ok = (idx < NEL) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    uoff[t] = ok ? (unsigned)(((size_t)cc * vol + (size_t)gy * a.W + gx) * 4) : 0xfffffff0u;   // vol*CH*4 < 2^32 is checked by the host
  }
  const unsigned stage_bytes = (unsigned)((size_t)CH * vol * 4 - 1) + 1u;
  float stage[NLD];
  auto prefetch = [&](int dz, int c0) {
    const int zz = IS3D ? z + dz - 1 : 0;
    // (the plane offset is folded into the base; the range then ends zz planes late, inside the next channel or the
    // allocation's tail, never beyond what an in-image offset of this stage can reach)
    const BufRsrcC r = make_rsrc_c(xb + (size_t)c0 * vol + (size_t)zz * plane, stage_bytes - (unsigned)((size_t)zz * plane * 4));
#pragma unroll
    for (int t = 0; t < NLD; ++t) stage[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, uoff[t], 0, 0));
  };
  // Weights of one stage go global -> LDS directly (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no VGPRs),
  // one stage ahead of their use; the A operand is then a conflict-free ds_read_b32.  (A buffer load, not global_load_lds:
  // behind the FLAT-encoded DMA the compiler turns every later wait into a wait for everything -- the operand reads of the
  // next MFMA group, just issued, included; docs/history/r05_wino3_pipeline.md.)  The lane's part of the offset is fixed,
  // the stage's part is scalar: no address arithmetic on the vector ALU per stage.
  const BufRsrcC wrs = make_rsrc_c(a.w, 0x7ffffff0u);
  constexpr int NWQ = (NWI + 3) / 4;
  unsigned wvoff[NWQ];
#pragma unroll
  for (int q = 0; q < NWQ; ++q) {
    int row = (wave + 4 * q) * RPI + lane / LPR;
    if (row > WROWS - 1) row = WROWS - 1;               // tail lanes re-read the last row (their LDS slots are never used)
    const int tap = row / CH, ci = row - tap * CH;
    wvoff[q] = (unsigned)(((size_t)tap * a.cin + ci) * a.cout + cout0 + (lane % LPR) * 4) * 4u;
  }
  auto stage_weights = [&](int dz, int c0, float* wdst) {
    const unsigned soff = (unsigned)(((size_t)dz * 9 * a.cin + c0) * a.cout) * 4u;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int wi = wave + 4 * q;                      // wave-uniform
      if (wi < NWI) dma16_to_lds(wrs, (LdsF)&wdst[0] + wi * 256, wvoff[q], soff);
    }
  };
  // iteration space: (dz, c0) pairs with an in-range z plane
  const int nchunk = a.cin / CH;
  int dz_lo = 0, dz_hi = KD;
  if (IS3D) { if (z == 0) dz_lo = 1; if (z == a.D - 1) dz_hi = KD - 1; }
  const int niter = (dz_hi - dz_lo) * nchunk;
  if (niter > 0) { prefetch(dz_lo, 0); stage_weights(dz_lo, 0, wbuf0); }
  auto stage_body = [&](int it, float* wcur, float* wnext) __attribute__((always_inline)) {
    // Both LDS images are double-buffered: buffer (it&1) was last read in iteration it-2 and every wave has passed
    // the barrier of iteration it-1 since, so it can be overwritten without a barrier in front -> one barrier per stage.
    float* tile = tile2[it & 1];
#pragma unroll
    for (int t = 0; t < NLD; ++t)
      if (threadIdx.x + 256 * t < NEL) tile[threadIdx.x + 256 * t] = stage[t];
    __syncthreads();                                   // tile stores + the stage's weight DMA (vmcnt(0)) visible to all
    if (it + 1 < niter) {                              // both in flight during the MFMAs
      prefetch(dz_lo + (it + 1) / nchunk, ((it + 1) % nchunk) * CH);
      stage_weights(dz_lo + (it + 1) / nchunk, ((it + 1) % nchunk) * CH, wnext);
    }
    {
      // Operand pipeline: the A/B values of group g+1 (one tap x 2 channels: CB + PR LDS values) are read while the
      // PR*CB MFMAs of group g issue, so no MFMA waits on an LDS round trip (left to itself the scheduler parks each
      // ds_read right in front of its first use).
      const float* wl = &wcur[l31 + half * RW];
      const float* tl = &tile[half * ROWS * MF_COLS + (wave * PR) * MF_COLS + l31];
      constexpr int NCP = CH / 2, NG = 9 * NCP;
      float av[2][CB], bv[2][PR];
      auto load_group = [&](int g, float (&A)[CB], float (&B)[PR]) __attribute__((always_inline)) {
        const int tap = g / NCP, cp = (g % NCP) * 2;
        const int r = tap / 3, sx = tap - 3 * r;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) A[cb] = wl[(tap * CH + cp) * RW + cb * 32];
#pragma unroll
        for (int pr = 0; pr < PR; ++pr) B[pr] = tl[cp * ROWS * MF_COLS + (pr + r) * MF_COLS + sx];
      };
      load_group(0, av[0], bv[0]);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load_group(g + 1, av[(g + 1) & 1], bv[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < PR; ++pr)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[pr][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][cb], bv[g & 1][pr], acc[pr][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int it = 0; it < niter; it += 2) {                // niter is even: Cin is a multiple of 2*CH
    stage_body(it, wbuf0, wbuf1);
    stage_body(it + 1, wbuf1, wbuf0);
  }
  const int x = x0 + l31;
  if (x < a.W) {
#pragma unroll
    for (int pr = 0; pr < PR; ++pr) {
      const int y = y0 + wave * PR + pr;
      if (y >= a.H) continue;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cout0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[pr][cb][r];
          if (a.relu) v = fmaxf(v, 0.f);
          a.y[((size_t)b * a.cout + co) * vol + (size_t)z * plane + (size_t)y * a.W + x] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 3x3(x3) convolution in the Winograd domain, F(2x2, 3x3), fp32 on v_mfma_f32_32x32x2_f32:
//   Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A      (Lavin & Gray; d = 4x4 input patch of a 2x2 output block)
// 16 multiplies per 4 outputs instead of 36: the contraction over input channels becomes 16 independent GEMMs (one per
// position p of the 4x4 transform domain)
//   D_p[cout 32][block 32] += Wt_p[cout][k] * Xt_p[k][block],   k = two consecutive input channels
//   raw halo tile  global -> registers (fetched three stages ahead) -> LDS at the start of the next phase
//                  (8-wave variant: global -> LDS by DMA into a ring of four copies, HDMA below)
//   B^T d B        per (channel, block) half patch by all threads, LDS -> LDS [position pair][c][block][2]
//   G g G^T        precomputed at pack time; the stage's slice goes global -> LDS by buffer_load_dwordx4 ... lds (DMA, one stage ahead)
// and the epilogue applies A^T . A per output channel register, adds the bias, clamps (ReLU) and stores pixel pairs.
// The 16 positions of a (32 output channels x 32 blocks) tile are split over TWO waves (8 positions = 128 accumulator
// registers each), so that a wave fits a 256-register budget and two waves share a SIMD: while one is between its barriers
// the other one's MFMAs keep the matrix cores busy (all 16 positions in one 512-register wave, one wave per SIMD, exposes
// every such gap: measured 2.78 vs 2.42 ms per 1024^2 forward).  Workgroup = 2 position halves x NCG output-channel groups
// x NPG pixel groups stacked in y (4 or 8 waves); stage = 4 input channels.  Each wave ends with partial outputs (A^T . A
// is linear in the positions); the two halves swap half of their 16 output-channel registers through LDS and each finishes
// (bias, ReLU, store) its own 8.
// Numerics: not the summation order of a direct convolution; the transforms are exact in binary up to the roundings of
// their additions (measured against the CPU oracle: max |d| 7.1e-8 at |ref| <= 7.7e-2, the direct kernel 7.5e-8;
// tools/cnn_error_probe.py, tests/test_parity_gpu.py::test_cnn_winograd_layers_vs_oracle).  FNX_PRECISION_FP32_DIRECT
// (fnx_multiscale_forward's precision_mode) keeps every layer on the direct kernels instead.
// The kernel is a PERSISTENT software pipeline (the round-1 kernel serialised every stage -- halo tile -> LDS, barrier,
// transform, barrier, operand reads, MFMAs -- and paid a full prologue and epilogue per 256-pixel tile: time = rounds x
// (5.5 us + stages x 1.31 us) against 0.91 us of MFMA work per stage, 47 % of the MFMA peak at 1024^2):
//  * every LDS image has two copies and a phase has ONE barrier.  Between two barriers a wave issues
//      registers -> LDS      halo tile of stage s+2, fetched during the phase before (into the copy the transform of stage s read
//                            one phase ago), then
//      global -> registers   halo tile of stage s+3          (it has until this point of the NEXT phase to land; the barrier
//                            in between waits for vmcnt(NLD): these loads stay in flight, the weight DMA issued before them lands)
//      global -> LDS (DMA)   transformed weights of stage s+1
//      LDS -> LDS            input transform of stage s+1 (halo tile s+1 -> xt[s+1]) in the gaps of the MFMA stream
//      16 MFMAs              second k-step of stage s-1, then first k-step of stage s: the stream is rotated by half a
//                            stage against the barriers and each k-step's operands are read half a phase before its
//                            MFMAs, so a wave leaves a barrier with its next 8 MFMAs' operands already in registers
//  * a workgroup walks over tiles (tile = blockIdx.x, += gridDim.x) and the three streams -- halo fetch (three stages ahead),
//    weights + transform (one ahead), MFMAs -- each carry their own tile: the fetches run on into the next tile while the
//    MFMAs finish the current one, so only the first tile of a workgroup pays a prologue.  Between tiles: the last
//    k-step, the output transform + half-exchange (its own LDS buffer: the xt copies already hold the next tile), stores.
//  * the transform work is dealt out in half patches (two of the four rows of B^T d B: 4 ds_read2_b64, 4 xor, 16 adds,
//    4 ds_write2) so that all threads carry the same share, with no per-lane selects.
//  * the transformed weights a workgroup DMAs per stage are one contiguous 16*4*RW-float block (repack_wino3_kernel).
// ---------------------------------------------------------------------------------------------------
// Output stores carry the non-temporal hint (the next layer reads them after this launch is through; measured 2.345 ->
// 2.312 ms per 1024^2 forward).  The same hint on the halo fetch (aux = 2) measured 4 % SLOWER.  Neither changes the
// L2 picture (PMC: 10.8 M hits / 7.9 M misses per launch): the streamed activations flush an XCD's 4 MB L2 about once per
// tile period, so about half of the transformed-weight reads miss it and are served by the Infinity Cache -- that, not
// HBM, is the "traffic beyond the activations" FETCH_SIZE shows for these launches (it counts Infinity-Cache hits).
#define W3_ST(p, v) do { const float2 v_ = (v); __builtin_nontemporal_store((f32x2){v_.x, v_.y}, (f32x2*)(p)); } while (0)
#ifndef W3_ABL
#define W3_ABL 0
#endif
#ifndef W3_HDMA
#define W3_HDMA 1
#endif
constexpr int WCOLS = 34;                              // halo tile row: 32 pixels + halo
constexpr int W3C = 4;                                 // input channels per stage of conv3_wino3_kernel
inline int wino3_rw(int cout) { return cout % 64 == 0 ? 64 : 32; }   // its output channels per workgroup
template <int NCG, int NPG, bool IS3D = false>
__global__ __launch_bounds__(128 * NCG * NPG, 2) void conv3_wino3_kernel(ConvArgs a, const float* __restrict__ wt, int ntx,
                                                                         int nty, int ntiles) {
  constexpr int C = W3C;
  constexpr int NWV = 2 * NCG * NPG, NT = 64 * NWV;
  constexpr int ROWS = 4 * NPG + 2;
  constexpr int NB = 32 * NPG;
  constexpr int RW = 32 * NCG;
  constexpr int WROWS = 16 * C;
  constexpr int WD = NCG * NPG == 4 ? 2 : 1;             // stages the weight DMA runs ahead (two 4-wave workgroups per CU: no room for four copies)
  constexpr int NWI = WROWS * RW / 256, NDMA = NWI / NWV;       // 1-KiB DMA instructions per stage / per wave
  constexpr int NEL = C * ROWS * WCOLS, NLD = (NEL + NT - 1) / NT;
  constexpr int NUNIT = 2 * C * NB, UPT = NUNIT / NT;          // half patches per stage / per thread
  static_assert(NUNIT % NT == 0 && (C * NB) % 64 == 0, "half patches must deal out evenly, wave-uniform in the half");
  static_assert(NWI % NWV == 0 && UPT <= 4, "stage shape");
  // HDMA (the 8-wave variant: one workgroup per CU has the LDS): the halo tile goes global -> LDS by DMA into a ring of four copies;
  // the 4-wave variants (two workgroups per CU, 80 KB each) take it through registers into two
  constexpr bool HDMA = W3_HDMA && NCG * NPG == 4;
  constexpr int RAWN = HDMA ? NLD * NT : NEL;            // (DMA: the lanes of the last instruction beyond the tile land in the padding)
  __shared__ __attribute__((aligned(16))) float raw0[RAWN];
  __shared__ __attribute__((aligned(16))) float raw1[RAWN];
  __shared__ __attribute__((aligned(16))) float raw2[HDMA ? RAWN : 4];
  __shared__ __attribute__((aligned(16))) float raw3[HDMA ? RAWN : 4];
  __shared__ __attribute__((aligned(16))) float xt0[16 * C * NB];
  __shared__ __attribute__((aligned(16))) float xt1[16 * C * NB];
  __shared__ __attribute__((aligned(16))) float wbuf0[WROWS * RW];
  __shared__ __attribute__((aligned(16))) float wbuf1[WROWS * RW];
  __shared__ __attribute__((aligned(16))) float wbuf2[WD == 2 ? WROWS * RW : 4];
  __shared__ __attribute__((aligned(16))) float wbuf3[WD == 2 ? WROWS * RW : 4];
  __shared__ __attribute__((aligned(16))) float exch[NWV * 16 * 64];   // epilogue: 4 registers x 4 outputs x 64 lanes per wave
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int h = wave & 1, g = wave >> 1;
  const int cg = g % NCG, pg = g / NCG;
  const int ngrp = a.cout / RW, nchunk = a.cin / C;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;
  const unsigned stage_bytes = (unsigned)((size_t)C * vol * 4 - 1) + 1u;
  const int tstep = gridDim.x;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();   // max(v, -inf) == v: the ReLU without a select per value

  // Tiles are numbered x fastest, then y, output-channel group, z, sample.  A workgroup's tiles are blockIdx.x + k*gridDim.x:
  // the tile's digits are advanced by the digits of the step with carries (one integer division chain per workgroup, not
  // three per tile: scalar division is a ~30-instruction dependent chain).
  struct Tile { int tx, ty, grp, z, b; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Tile T;
    T.tx = t % ntx; t /= ntx;
    T.ty = t % nty; t /= nty;
    T.grp = t % ngrp; t /= ngrp;
    T.z = t % a.D; T.b = t / a.D;
    return T;
  };
  const Tile T0 = decode(blockIdx.x), TS = decode(tstep);
  auto next_tile = [&](Tile T) __attribute__((always_inline)) {
    int c;
    T.tx += TS.tx;        c = T.tx >= ntx;   T.tx -= c ? ntx : 0;
    T.ty += TS.ty + c;    c = T.ty >= nty;   T.ty -= c ? nty : 0;
    T.grp += TS.grp + c;  c = T.grp >= ngrp; T.grp -= c ? ngrp : 0;
    T.z += TS.z + c;      c = T.z >= a.D;    T.z -= c ? a.D : 0;
    T.b += TS.b + c;
    return T;
  };
  auto dz_lo_of = [&](const Tile& T) __attribute__((always_inline)) { return IS3D && T.z == 0 ? 1 : 0; };
  auto dz_hi_of = [&](const Tile& T) __attribute__((always_inline)) { return IS3D ? (T.z == a.D - 1 ? 2 : 3) : 1; };

  // ---- halo-fetch stream (three stages ahead of the MFMAs): its tile is (xR, rz), cursor (rdz, rc0) ----
  const float* xR;                                      // sample base of the stream's tile
  int rz, rdz, rc0 = 0;
  // per-thread slots of the [C][ROWS][34] halo tile: the tile-independent part of the byte offset and (row, col); a slot
  // outside the image gets an out-of-range offset (the buffer load then returns 0)
  unsigned uoff[NLD], ubase[NLD], urc[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int idx = threadIdx.x + NT * t;
    const int cc = idx / (ROWS * WCOLS);
    const int rem = idx - cc * ROWS * WCOLS;
    const int row = rem / WCOLS, col = rem - row * WCOLS;
    ubase[t] = (unsigned)(((size_t)cc * vol + (size_t)row * a.W + col) * 4);
    urc[t] = idx < NEL ? (unsigned)(row << 16 | col) : 0xffffffffu;
  }
  auto enter_R = [&](const Tile& T) __attribute__((always_inline)) {
    xR = a.x + (size_t)T.b * a.cin * vol; rz = T.z; rdz = dz_lo_of(T); rc0 = 0;
    const int tx0 = T.tx * 32, ty0 = T.ty * (4 * NPG);
    const unsigned toff = (unsigned)(((size_t)(ty0 - 1) * a.W + (tx0 - 1)) * 4);      // (wraps for the first row/column: those slots are out of the image)
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
      const int gx = tx0 - 1 + (int)(urc[t] & 0xffff), gy = ty0 - 1 + (int)(urc[t] >> 16);
      const bool ok = (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
      uoff[t] = ok ? ubase[t] + toff : 0xfffffff0u;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  enter_R(T0);
  float stage[NLD];
  // HDMA: straight into `rawdst` (buffer_load_dword ... lds: 64 consecutive floats per wave instruction, zeros for the out-of-range
  // offsets) -- no registers, no LDS store, and nothing waits for it before the barrier of the NEXT phase
  auto prefetch = [&](float* rawdst = nullptr) __attribute__((always_inline)) {
    const int zz = IS3D ? rz + rdz - 1 : 0;
    const BufRsrcC r = (W3_ABL & 128) ? make_rsrc_c(a.x, 1u << 20)
                                      : make_rsrc_c(xR + (size_t)rc0 * vol + (size_t)zz * plane, stage_bytes - (unsigned)((size_t)zz * plane * 4));
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
      const unsigned vo = (W3_ABL & 128) ? (unsigned)(threadIdx.x * 4 + t * 2048) : uoff[t];
      if (HDMA) dma4_to_lds(r, (LdsF)&rawdst[0] + wave * 64 + NT * t, vo);
      else stage[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0));
    }
    rc0 += C;
    if (rc0 >= a.cin) { rc0 = 0; ++rdz; }
  };
  auto store_raw = [&](float* rawdst) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NLD; ++t)
      if (threadIdx.x + NT * t < NEL) rawdst[threadIdx.x + NT * t] = stage[t];
  };
  // ---- weight stream (one stage ahead): the stage's [16][C][RW] image is one contiguous block of the packed weights
  // ([dz][Cin/C][Cout/RW] blocks); DMA instruction q of this wave moves its floats [wi*256, (wi+1)*256), wi = wave + NWV*q
  // (buffer_load_dwordx4 ... lds, not global_load_lds: the compiler counts a FLAT-encoded DMA as a flat access that may return
  // out of order and turns every later wait -- the halo registers, the transform's LDS reads -- into a wait for ALL of them)
  const BufRsrcC wrs = make_rsrc_c(wt, 0x7ffffff0u);
  const unsigned wlane = (unsigned)(wave * 256 + lane * 4) * 4u;   // per-lane byte offset inside a DMA instruction's 1 KiB x NWV
  unsigned wgrp;                                        // byte offset of block (dz 0, chunk 0) of the stream's output-channel group
  int wblk;                                             // (dz, chunk) block index of the cursor
  const unsigned wstep = (unsigned)ngrp * (16 * C * RW) * 4u;
  auto enter_W = [&](const Tile& T) __attribute__((always_inline)) {
    wgrp = (unsigned)T.grp * (16 * C * RW) * 4u; wblk = dz_lo_of(T) * nchunk;
    __builtin_amdgcn_sched_barrier(0);
  };
  enter_W(T0);
  auto stage_weights = [&](float* wdst) __attribute__((always_inline)) {
    const unsigned sb = wgrp + (unsigned)wblk * wstep;
#pragma unroll
    for (int q = 0; q < NDMA; ++q)
      dma16_to_lds(wrs, (LdsF)&wdst[0] + (wave + NWV * q) * 256, wlane, sb + (unsigned)(NWV * 256 * q) * 4u);
    ++wblk;
  };
  // ---- input transform.  Half patch u = threadIdx.x + NT*i: block n = u % NB, channel c = (u / NB) % C, row half
  // hp = u / (NB*C) (wave-uniform).
  //   hp 0 -> rows 0,1 of B^T d B:  t0 = d0 - d2, t1 = d1 + d2        hp 1 -> rows 2,3:  t2 = d2 - d1, t3 = d1 - d3
  // as ONE instruction stream: t_a = P - Q, t_b = R + (S ^ sign) with (P,Q,R,S) = (d0,d2,d1,d2) or (d2,d1,d1,d3) and
  // sign = 0 or the sign bit (x + (-y) == x - y bit for bit).
  int rd_off[UPT][4], wr_off[UPT];
  unsigned sgn[UPT];
#pragma unroll
  for (int i = 0; i < UPT; ++i) {
    const int u = threadIdx.x + NT * i;
    const int n = u % NB, c = (u / NB) % C, hp = u / (NB * C);
    const int bx = n & 15, by = n >> 4;
    const int d = c * ROWS * WCOLS + (2 * by) * WCOLS + 2 * bx;
    rd_off[i][0] = d + (hp ? 2 : 0) * WCOLS; rd_off[i][1] = d + (hp ? 1 : 2) * WCOLS;
    rd_off[i][2] = d + 1 * WCOLS;            rd_off[i][3] = d + (hp ? 3 : 2) * WCOLS;
    wr_off[i] = 2 * ((4 * hp) * C * NB + c * NB + n);   // xt is [position pair][channel][block][2]
    sgn[i] = hp ? 0x80000000u : 0u;
  }
  float2 e[4][2];
  float tc[2][4];
  auto xf_load = [&](int i, const float* rawsrc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[j][0] = *(const float2*)(rawsrc + rd_off[i][j]); e[j][1] = *(const float2*)(rawsrc + rd_off[i][j] + 2);
    }
  };
  // hpk: the row half where the caller knows it at compile time (0 / 1: a plain add or subtract), 2: by the sign mask
  auto xf_cols = [&](int i, int hpk = 2) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float P = s & 1 ? e[0][s >> 1].y : e[0][s >> 1].x, Q = s & 1 ? e[1][s >> 1].y : e[1][s >> 1].x;
      const float R = s & 1 ? e[2][s >> 1].y : e[2][s >> 1].x, S = s & 1 ? e[3][s >> 1].y : e[3][s >> 1].x;
      tc[0][s] = P - Q;
      tc[1][s] = hpk == 0 ? R + S : hpk == 1 ? R - S : R + __builtin_bit_cast(float, __builtin_bit_cast(unsigned, S) ^ sgn[i]);
    }
  };
  auto xf_rows_store = [&](int i, float* xtdst) __attribute__((always_inline)) {
    float* o = xtdst + wr_off[i];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                      // row 2 hp + rr of B^T d B: positions 4 row + 0..3 = pairs 2 row, 2 row + 1
      *(float2*)(o + (2 * rr + 0) * 2 * C * NB) = make_float2(tc[rr][0] - tc[rr][2], tc[rr][1] + tc[rr][2]);
      *(float2*)(o + (2 * rr + 1) * 2 * C * NB) = make_float2(tc[rr][2] - tc[rr][1], tc[rr][1] - tc[rr][3]);
    }
  };

  // ---- prologue (first tile of the workgroup only): halo tiles of stages 0 and 1, weights of stage 0, transform of stage 0
  if (HDMA) {
    prefetch(raw0);
    stage_weights(wbuf0);
    if (WD == 2) stage_weights(wbuf1);
    prefetch(raw1);
    prefetch(raw2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < UPT; ++i) { xf_load(i, raw0); xf_cols(i); xf_rows_store(i, xt0); }
    __syncthreads();
  } else {
    prefetch();
    stage_weights(wbuf0);
    if (WD == 2) stage_weights(wbuf1);
    store_raw(raw0);
    prefetch();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < UPT; ++i) { xf_load(i, raw0); xf_cols(i); xf_rows_store(i, xt0); }
    store_raw(raw1);
    prefetch();
    __syncthreads();
  }

  // ---- MFMA stream ----
  f32x16 acc[8];
  // both LDS images keep the two positions of a pair next to each other ([pair][channel][RW or NB][2]): one ds_read_b64
  // per operand and position PAIR (half the LDS time of two 4-byte reads)
  const int wlo = 2 * ((4 * h * C + half) * RW + cg * 32 + l31), tlo = 2 * ((4 * h * C + half) * NB + pg * 32 + l31);
  float av[2][8], bv[2][8];
  auto load_k = [&](int ks, const float* wcur, const float* xcur) __attribute__((always_inline)) {
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
      const float2 wa = *(const float2*)(wcur + wlo + 2 * (p2 * C + 2 * ks) * RW), xb = *(const float2*)(xcur + tlo + 2 * (p2 * C + 2 * ks) * NB);
      av[ks][2 * p2] = wa.x; av[ks][2 * p2 + 1] = wa.y; bv[ks][2 * p2] = xb.x; bv[ks][2 * p2 + 1] = xb.y;
    }
  };
  // One phase (between two barriers), stage s of the MFMA stream's tile.  M0: a stage s-1 exists in this tile (its second
  // k-step is issued first; without it the accumulators start from zero).  The weight/transform stream (stage s+1) and the
  // halo stream (stage s+3) are always live: behind a workgroup's last tile they run on a stand-in tile whose results
  // nobody reads, which keeps every phase the same straight-line code.
  auto phase_sched = [&](auto m0, auto late_, auto hpw_, const float* wcur, float* wnext, const float* xcur, float* xnext,
                         const float* rawnext, float* rawfree) __attribute__((always_inline)) {
    constexpr bool M0 = decltype(m0)::value;
    constexpr bool LATE = decltype(late_)::value;
    constexpr int HPW = decltype(hpw_)::value;            // the row half of this wave's half patches (UPT == 1); with UPT == 2 it is i         // the second wave of each SIMD runs its fillers half a phase later
    // (the fetches of the phase are issued BEHIND its first MFMAs, whose operands are already in registers: the matrix
    // pipe restarts right behind the barrier instead of idling through ~40 address/VMEM/LDS instructions per wave)
    constexpr int S_XF = LATE ? 4 : 0, S_DMA = LATE ? 5 : 1, S_K0 = LATE ? 6 : 2, S_K1 = LATE ? 11 : 7;
#pragma unroll
    for (int slot = 0; slot < 16; ++slot) {
      const int ks = slot < 8 ? 1 : 0, p = slot & 7;
      if (!(W3_ABL & 2) && (slot >= 8 || M0)) {
        if (slot >= 8 && !M0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][p], bv[ks][p], zero, 0, 0, 0);
        } else {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][p], bv[ks][p], acc[p], 0, 0, 0);
        }
      }
      if ((W3_ABL & 2) && (slot >= 8 || M0)) asm volatile("; keep" :: "v"(av[ks][p]), "v"(bv[ks][p]));
      __builtin_amdgcn_sched_barrier(0);
      if (!(W3_ABL & 4) && slot == S_XF) xf_load(0, rawnext);
      if (!(W3_ABL & 16) && slot == S_DMA) stage_weights(wnext);
      if (!(W3_ABL & 8) && slot == S_DMA + 1) {
        if (HDMA) prefetch(rawfree);                      // (behind the weights: the barrier waits for those, not for these)
        else { store_raw(rawfree); prefetch(); }          // the halo tile fetched one phase ago; then the next one
      }
      if (!(W3_ABL & 32) && slot == S_K0) load_k(0, wcur, xcur);
      // half patch i: loaded at slot S_XF + 4i, columns 3 slots later, rows + stores 4 slots later
      const int rel = slot - S_XF, ui = rel / 4, us = rel % 4;
      if (!(W3_ABL & 4) && rel >= 0) {
        if (us == 0 && ui > 0 && ui < UPT) xf_load(ui, rawnext);
        if (us == 3 && ui < UPT) xf_cols(ui, UPT == 2 ? ui : HPW);
        if (us == 0 && ui > 0 && ui - 1 < UPT) xf_rows_store(ui - 1, xnext);
      }
      if (!(W3_ABL & 32) && slot == S_K1) load_k(1, wcur, xcur);             // the registers of k-step 1 are free: its MFMAs have all been issued
      __builtin_amdgcn_sched_barrier(0);
    }
    // the phase's own weight DMA (stage s+2) may still be in flight; everything older -- the DMA of stage s+1, the halo loads --
    // has landed: vmcnt(NDMA), lgkmcnt(0)
    asm volatile("" ::: "memory");
    constexpr int NFLY = NLD + (WD == 2 ? NDMA : 0);
    __builtin_amdgcn_s_waitcnt((NFLY & 15) | ((NFLY >> 4) << 14) | 0x70);
    if (!(W3_ABL & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  // The two waves of a SIMD (waves w and w + 4 of an 8-wave workgroup) would otherwise run the same instruction at the same
  // time and wait at the same points, leaving the matrix pipe idle together: the second one runs the whole tile loop with
  // the late filler schedule (ONE branch around two copies of the loop: a branch per phase made the register allocator
  // spill 600 registers at the joins).
  // (2D only: with the 3D layers' 48 stages per tile the late schedule measured 5 % slower, 82 -> 86 ms at 256^3)
  constexpr bool STAGGER = NWV == 8 && UPT == 1 && !IS3D;
  auto run = [&](auto late_, auto hpw_) __attribute__((always_inline)) {
    auto phase = [&](auto m0, const float* wcur, float* wnext, const float* xcur, float* xnext, const float* rawnext,
                     float* rawfree) __attribute__((always_inline)) {
      phase_sched(m0, late_, hpw_, wcur, wnext, xcur, xnext, rawnext, rawfree);
    };
    int tM = blockIdx.x;
    Tile TM = T0;
    for (;;) {
      const int niter = (dz_hi_of(TM) - dz_lo_of(TM)) * nchunk;   // even, >= 4 (the host checks Cin % (4 C))
      const bool more = tM + tstep < ntiles;
      const Tile TN = more ? next_tile(TM) : TM;          // the streams' next tile (stand-in behind the last one: this tile again)
      // four stages per round: stage s computes on weight copy s % 4 and DMAs stage s+2 into copy (s+2) % 4
      auto group = [&](auto m0, bool last) __attribute__((always_inline)) {
        // (halo copies: stage s transforms copy (s+1) % 4 and DMAs the tile of stage s+3 into copy (s+3) % 4; through registers:
        // transforms copy (s+1) % 2 and stores the tile of stage s+2 into copy s % 2)
        phase(m0, wbuf0, WD == 2 ? wbuf2 : wbuf1, xt0, xt1, raw1, HDMA ? raw3 : raw0);
        if (last) enter_R(TN);                            // stage niter-3 fetches the halo tile of the next tile's stage 0
        phase(T_{}, wbuf1, WD == 2 ? wbuf3 : wbuf0, xt1, xt0, HDMA ? raw2 : raw0, HDMA ? raw0 : raw1);
        if (last && WD == 2) enter_W(TN);                 // ... and stage niter-WD its weights
        phase(T_{}, WD == 2 ? wbuf2 : wbuf0, WD == 2 ? wbuf0 : wbuf1, xt0, xt1, HDMA ? raw3 : raw1, HDMA ? raw1 : raw0);
        if (last && WD == 1) enter_W(TN);
        phase(T_{}, WD == 2 ? wbuf3 : wbuf1, WD == 2 ? wbuf1 : wbuf0, xt1, xt0, raw0, HDMA ? raw2 : raw1);
      };
      group(F_{}, niter == 4);
      for (int it = 4; it < niter; it += 4) group(T_{}, it + 4 == niter);
      // the second k-step of the last stage
#pragma unroll
      for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][p], bv[1][p], acc[p], 0, 0, 0);
      if (W3_ABL & 64) {
#pragma unroll
        for (int p = 0; p < 8; ++p) asm volatile("; keep" :: "v"(acc[p]));
        if (!more) break;
        tM += tstep; TM = TN; continue;
      }

      // Output transform A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]); this wave holds rows 2h, 2h+1 of M (acc[4*(row-2h) + s]):
      //   h = 0:  t0 = M0 + M1, t1 = M1          h = 1:  t0 = M2, t1 = -M2 - M3
      // Register r belongs to half (r >> 3): the partial outputs of the other half's registers go through LDS to the
      // partner wave, the own ones are completed with the partner's, then bias, ReLU, store.  All of it on register PAIRS
      // (r, r+1) as packed fp32 (v_pk_add_f32: the same IEEE additions, two per instruction); the barriers of the
      // exchange wait for LDS only (a __syncthreads() would also wait for the stores of the round before to be acknowledged).
      {
        const int bx = l31 & 15, byl = l31 >> 4;
        const int tx0 = TM.tx * 32, ty0 = TM.ty * (4 * NPG);
        const int x = tx0 + 2 * bx, y = ty0 + 2 * (2 * pg + byl);
        // channel of register r = 8 hh + 4 round + 2 j (+1):  cg*32 + 2 j + 8 (2 hh + round) + 4 half
        const int co_own = TM.grp * RW + cg * 32 + 16 * h + 4 * half;
        f32x2 bias2[2][2];
#pragma unroll
        for (int round = 0; round < 2; ++round)
#pragma unroll
          for (int j = 0; j < 2; ++j) bias2[round][j] = *(const f32x2*)&a.bias[co_own + 8 * round + 2 * j];
        float* obase = a.y + ((size_t)TM.b * a.cout + co_own) * vol + (size_t)TM.z * plane + (size_t)y * a.W + x;
        // full tiles store through a buffer resource on the wave's 16 channels: one 32-bit lane offset per pixel row (the
        // channel quad of the lane's half included) and the channel in the scalar offset -- no 64-bit address arithmetic on
        // the vector ALU, which runs in the matrix pipe's time (tools/ubench/mfma_coexec.hip)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const BufRsrcC yrs = (W3_ABL & 256) ? make_rsrc_c(a.y + (size_t)blockIdx.x * 65536, 1u << 20)
                                            : make_rsrc_c(a.y + ((size_t)TM.b * a.cout + TM.grp * RW + cg * 32 + 16 * h) * vol, (unsigned)(16 * vol * 4 - 1) + 1u);
        const unsigned yoff0 = (W3_ABL & 256) ? threadIdx.x * 8u : (unsigned)(((size_t)(4 * half) * vol + (size_t)TM.z * plane + (size_t)y * a.W + x) * 4),
                       yoff1 = (W3_ABL & 256) ? yoff0 + 4096u : yoff0 + (unsigned)a.W * 4u;
        const unsigned chan_bytes = (W3_ABL & 256) ? 8192u : (unsigned)(vol * 4);
        const bool inx = x < a.W, iny = y < a.H, inx1 = x + 1 < a.W, iny1 = y + 1 < a.H;
        const bool full = (tx0 + 32 <= a.W) & (ty0 + 4 * NPG <= a.H);       // (wave-uniform) no clipped pixel in the tile
        f32x2* ex_out = (f32x2*)&exch[wave * 16 * 64] + lane;
        const f32x2* ex_in = (const f32x2*)&exch[(wave ^ 1) * 16 * 64] + lane;
        auto out_part = [&](auto hh_, int r, f32x2 (&pt)[4]) __attribute__((always_inline)) {
          constexpr int HH = decltype(hh_)::value;           // this wave's position half
          f32x2 t0[4], t1[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const f32x2 a0 = {acc[s][r], acc[s][r + 1]}, a1 = {acc[4 + s][r], acc[4 + s][r + 1]};
            if (HH == 0) { t0[s] = a0 + a1; t1[s] = a1; }
            else { t0[s] = a0; t1[s] = (-a0) - a1; }
          }
          pt[0] = (t0[0] + t0[1]) + t0[2]; pt[1] = (t0[1] - t0[2]) - t0[3];
          pt[2] = (t1[0] + t1[1]) + t1[2]; pt[3] = (t1[1] - t1[2]) - t1[3];
        };
        auto send = [&](auto hh_, int round) __attribute__((always_inline)) {
          constexpr int HH = decltype(hh_)::value;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x2 pt[4];
            out_part(hh_, 8 * (1 - HH) + 4 * round + 2 * j, pt);
#pragma unroll
            for (int q = 0; q < 4; ++q) ex_out[(j * 4 + q) * 64] = pt[q];
          }
        };
        auto finish = [&](auto hh_, int round) __attribute__((always_inline)) {
          constexpr int HH = decltype(hh_)::value;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x2 pt[4], v[4];
            out_part(hh_, 8 * HH + 4 * round + 2 * j, pt);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q] = (pt[q] + ex_in[(j * 4 + q) * 64]) + bias2[round][j];
              v[q] = __builtin_elementwise_max(v[q], (f32x2){relu_lo, relu_lo});
            }
            float* o0 = obase + (size_t)(8 * round + 2 * j) * vol;
            float* o1 = o0 + vol;
            if (full) {
              const unsigned c0 = (unsigned)(8 * round + 2 * j) * chan_bytes, c1 = c0 + chan_bytes;
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v[0].x, v[1].x}), yrs, yoff0, c0, 2);   // (aux 2: non-temporal)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v[2].x, v[3].x}), yrs, yoff1, c0, 2);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v[0].y, v[1].y}), yrs, yoff0, c1, 2);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v[2].y, v[3].y}), yrs, yoff1, c1, 2);
            } else if (inx & iny) {
              if (inx1) {
                *(float2*)o0 = make_float2(v[0].x, v[1].x); *(float2*)o1 = make_float2(v[0].y, v[1].y);
                if (iny1) { *(float2*)(o0 + a.W) = make_float2(v[2].x, v[3].x); *(float2*)(o1 + a.W) = make_float2(v[2].y, v[3].y); }
              } else {
                o0[0] = v[0].x; o1[0] = v[0].y;
                if (iny1) { o0[a.W] = v[2].x; o1[a.W] = v[2].y; }
              }
            }
          }
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
#pragma unroll
        for (int round = 0; round < 2; ++round) {            // 4 of a half's 8 registers per round (16 values each way)
          // (round 0: the previous tile's reads of the exchange buffer are many barriers back)
          if (round) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }       // lgkmcnt(0) only
          if (h == 0) send(H0{}, round); else send(H1{}, round);
          __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
          if (h == 0) finish(H0{}, round); else finish(H1{}, round);
        }
      }
      if (!more) break;
      tM += tstep; TM = TN;
    }
  };
  // (the same split serves the input transform: with one half patch per thread the upper half of the waves has the rows 2, 3
  // of B^T d B, and each copy of the loop knows its half at compile time)
  static_assert(UPT == 2 || NB * C == NT / 2, "row half = upper half of the waves");
  if (UPT == 1 && wave >= NWV / 2) run(std::integral_constant<bool, STAGGER>{}, std::integral_constant<int, 1>{});
  else run(std::integral_constant<bool, false>{}, std::integral_constant<int, 0>{});
}

// blob: (Cout,Cin,3x3) -> G g G^T as [16][Cin][Cout], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// (3D: per z tap dz, [dz][16][Cin][Cout])
__global__ void pack_layer_wino_kernel(const float* __restrict__ w, float* __restrict__ pw, int cin, int cout, int kd) {
  const int n = cin * cout * kd;
  for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < n; q0 += gridDim.x * blockDim.x) {
    const int dz = q0 / (cin * cout), q = q0 - dz * cin * cout;
    const int co = q / cin, ci = q - co * cin;
    const float* g = w + ((size_t)q * kd + dz) * 9;
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float g0 = g[s], g1 = g[3 + s], g2 = g[6 + s];
      t[0][s] = g0; t[1][s] = 0.5f * ((g0 + g1) + g2); t[2][s] = 0.5f * ((g0 - g1) + g2); t[3][s] = g2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float u[4] = { t[r][0], 0.5f * ((t[r][0] + t[r][1]) + t[r][2]), 0.5f * ((t[r][0] - t[r][1]) + t[r][2]), t[r][2] };
#pragma unroll
      for (int s = 0; s < 4; ++s) pw[((size_t)(dz * 16 + 4 * r + s) * cin + ci) * cout + co] = u[s];
    }
  }
}

// The same G g G^T values re-ordered for conv3_wino3_kernel: [dz][Cin/4][Cout/RW][8 position pairs][4][RW][2] -- the 16*4*RW floats one
// workgroup DMAs per stage are ONE contiguous block (every workgroup of the launch reads the same blocks at about the
// same time: 64 rows of RW floats strided over the [16][Cin][Cout] image land on half of an XCD's L2 channels, the
// contiguous block on all of them).  RW = 64 where Cout % 64 == 0, else 32 (the kernel's output channels per workgroup).
__global__ void repack_wino3_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int kd, int rw) {
  const size_t n = (size_t)kd * 16 * cin * cout;
  const int nchunk = cin / W3C, ngrp = cout / rw;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(q % cout);
    size_t r = q / cout;
    const int ci = (int)(r % cin); r /= cin;
    const int pp = (int)(r % 16), dz = (int)(r / 16);
    const size_t o = (((((((size_t)dz * nchunk + ci / W3C) * ngrp + co / rw) * 8 + pp / 2) * W3C + ci % W3C) * rw) + co % rw) * 2 + pp % 2;
    dst[o] = src[q];
  }
}

#include "fnx_cnn_bf16x6.h"
#include "fnx_cnn_wino4.h"

// ---------------------------------------------------------------------------------------------------
// Implicit-GEMM 5x5(x5) convolution for the thin layers (3->32 and 32->8) on v_mfma_f32_16x16x4_f32 (exact fp32):
//   D[cout 16][pixel 16] += A[cout][k] * B[k][pixel],  k = four consecutive input channels of one tap
//   A: lane -> W[tap][c + (lane>>4)][m*16 + (lane&15)]       (LDS image [tap][CHS][CO], conflict-free)
//   B: lane -> X[c + (lane>>4)][y + r - 2][x + (lane&15) + s - 2]  from the LDS halo tile (the four 16-lane groups read
//      four channel planes whose stride, 12*68 floats, spreads them over disjoint banks)
//   D: lane holds pixel (lane&15) and output channels 4*(lane>>4)..+3
// Workgroup = 4 waves stacked in y; wave tile = 64 px (4 segments of 16) x 2 rows x MB*16 output channels.  A stage
// is one z plane x CHS input channels: the halo tile goes global -> registers -> LDS (prefetched during the previous
// stage's MFMAs), the stage's 25 x CHS x CO weights go global -> LDS by buffer_load_dwordx4 ... lds; both double-buffered,
// one barrier per stage (25*CHS/4*8*MB MFMAs per wave between barriers).  Padded output channels (32->8 uses half
// of the 16-row M block) cost MFMA cycles, not memory traffic.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAIR (Cout <= 8): the 16 rows of the M block are (dx, cout) for TWO x-adjacent output pixels, the 16 columns are pixel
// pairs, and k runs over a 6-wide window per tap row (weights zero where the tap falls outside an output's 5): 30
// k-positions per row-plane instead of 2 x 25 half-empty ones -- 0.6x the MFMAs of the plain mapping for 32->8.
// KPC (2 or 3 = Cin; 0: off): K packed over (tap, channel), see kpack_cin.
template <int CHS, int MB, bool IS3D, bool PAIR, int KPC = 0>
__global__ __launch_bounds__(256, 2) void conv5_mfma16_kernel(ConvArgs a, int cin_pad) {
  constexpr int KS = 5, PAD = 2, PR = 2;
  static_assert(KPC == 0 || (CHS == 4 && !PAIR), "packed K: one stage per z plane, all (<= 3) channels in it");
  constexpr int NKP = (25 * KPC + 3) / 4, KROWS = NKP * 4;           // K-groups of four / weight rows per plane (KPC)
  constexpr int NX = PAIR ? 2 : 4;                                    // MFMA column blocks per wave row (64 px either way)
  constexpr int KW = PAIR ? KS + 1 : KS;                              // k positions along x per tap row
  constexpr int XSEG = PAIR ? 32 : 16, XSTEP = PAIR ? 2 : 1;          // pixels per column block, pixel stride of a lane
  static_assert(!PAIR || MB == 1, "pairs fill one M block");
  constexpr int ROWS = 4 * PR + KS - 1, COLS = 64 + KS - 1;           // 12 x 68 halo tile per channel
  constexpr int CO = 16 * MB;
  constexpr int NEL = CHS * ROWS * COLS, NLD = (NEL + 255) / 256;
  constexpr int TAPF = CHS * CO;                                      // floats per tap in a stage (128 or 64)
  static_assert(TAPF == 128 || TAPF == 64, "a wave-wide 16-byte load covers 2 or 4 taps");
  constexpr int TPI = 256 / TAPF, LPT = 64 / TPI;                     // taps per wave-instruction, lanes per tap
  constexpr int NTAP = KS * KW;                                       // k positions per plane (25, or 30 for PAIR)
  constexpr int NWI = KPC ? (KROWS * CO + 255) / 256 : (NTAP + TPI - 1) / TPI;   // wave-instructions per stage
  __shared__ __attribute__((aligned(16))) float tile2[2][NEL];
  __shared__ __attribute__((aligned(16))) float wbuf0[NWI * 256];       // (two separate arrays and out-of-range-zero buffer
  __shared__ __attribute__((aligned(16))) float wbuf1[NWI * 256];       //  loads: see conv3_mfma_kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane & 15, kq = lane >> 4;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * (4 * PR);
  int zb = blockIdx.z;
  const int z = zb % a.D; const int b = zb / a.D;
  const size_t plane = (size_t)a.H * a.W, vol = plane * a.D;

  f32x4 acc[PR][NX][MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = PAIR ? (4 * kq + r) % 8 : mb * 16 + 4 * kq + r;
      const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
      for (int pr = 0; pr < PR; ++pr)
#pragma unroll
        for (int nx = 0; nx < NX; ++nx) acc[pr][nx][mb][r] = bv;
    }
  }

  const float* xb = a.x + (size_t)b * a.cin * vol;
  // per-thread staging slots of the [CHS][ROWS][COLS] halo tile (offsets and predicates are stage-invariant: Cin is a
  // multiple of CHS, or smaller than CHS with a single chunk whose missing channels read as zero)
  unsigned uoff[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    const int idx = threadIdx.x + 256 * t;
    const int cc = idx / (ROWS * COLS);
    const int rem = idx - cc * ROWS * COLS;
    const int row = rem / COLS, col = rem - row * COLS;
    const int gx = x0 - PAD + col, gy = y0 - PAD + row;
    const bool ok = (idx < NEL) & (cc < a.cin) & (gx >= 0) & (gx < a.W) & (gy >= 0) & (gy < a.H);
    uoff[t] = ok ? (unsigned)(((size_t)cc * vol + (size_t)gy * a.W + gx) * 4) : 0xfffffff0u;
  }
  const unsigned stage_bytes = (unsigned)((size_t)CHS * vol * 4 - 1) + 1u;
  float stage[NLD];
  auto prefetch = [&](int dz, int c0) {
    const int zz = IS3D ? z + dz - PAD : 0;
    const BufRsrcC r = make_rsrc_c(xb + (size_t)c0 * vol + (size_t)zz * plane, stage_bytes - (unsigned)((size_t)zz * plane * 4));
#pragma unroll
    for (int t = 0; t < NLD; ++t) stage[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, uoff[t], 0, 0));
  };
  // (weights by buffer_load_dwordx4 ... lds with a fixed lane offset and the stage in the scalar offset: see conv3_mfma_kernel)
  const BufRsrcC wrs = make_rsrc_c(a.w, 0x7ffffff0u);
  constexpr int NWQ = (NWI + 3) / 4;
  unsigned wvoff[NWQ];
#pragma unroll
  for (int q = 0; q < NWQ; ++q) {
    if (KPC) {                                            // the plane's KROWS x CO floats are one linear block
      int f = (wave + 4 * q) * 256 + lane * 4;
      if (f > KROWS * CO - 4) f = KROWS * CO - 4;         // (the tail re-reads the last floats into slots nobody reads)
      wvoff[q] = (unsigned)f * 4u;
    } else {
      int tap = TPI * (wave + 4 * q) + lane / LPT;
      if (tap > NTAP - 1) tap = NTAP - 1;                 // the tail re-reads the last tap into slots nobody reads
      wvoff[q] = (unsigned)((size_t)tap * cin_pad * CO + (lane % LPT) * 4) * 4u;
    }
  }
  // packed K: the lane's element of K-group m is k = 4 m + kq = (tap, channel); its offset in the halo tile
  int boff[KPC ? NKP : 1];
  if (KPC) {
#pragma unroll
    for (int m = 0; m < NKP; ++m) {
      const int k = 4 * m + kq, tap = k / (KPC ? KPC : 1), ci = k - tap * KPC;
      const int r = tap / KS, sx = tap - r * KS;
      boff[m] = tap < 25 ? ci * ROWS * COLS + r * COLS + sx : 0;      // (padding: zero weights on a valid address)
    }
  }
  auto stage_weights = [&](int dz, int c0, float* wdst) {
    const unsigned soff = KPC ? (unsigned)((size_t)dz * KROWS * CO) * 4u : (unsigned)(((size_t)dz * NTAP * cin_pad + c0) * CO) * 4u;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int wi = wave + 4 * q;                        // wave-uniform
      if (wi < NWI) dma16_to_lds(wrs, (LdsF)&wdst[0] + wi * 256, wvoff[q], soff);
    }
  };
  const int nchunk = cin_pad / CHS;
  int dz_lo = 0, dz_hi = IS3D ? KS : 1;
  if (IS3D) { dz_lo = PAD - z > 0 ? PAD - z : 0; dz_hi = a.D + PAD - z < KS ? a.D + PAD - z : KS; }
  const int niter = (dz_hi - dz_lo) * nchunk;
  if (niter > 0) { prefetch(dz_lo, 0); stage_weights(dz_lo, 0, wbuf0); }
  auto stage_body = [&](int it, float* wcur, float* wnext) __attribute__((always_inline)) {
    float* tile = tile2[it & 1];
#pragma unroll
    for (int t = 0; t < NLD; ++t)
      if (threadIdx.x + 256 * t < NEL) tile[threadIdx.x + 256 * t] = stage[t];
    __syncthreads();                                     // tile stores + the stage's weight DMA visible to all
    if (it + 1 < niter) {
      prefetch(dz_lo + (it + 1) / nchunk, ((it + 1) % nchunk) * CHS);
      stage_weights(dz_lo + (it + 1) / nchunk, ((it + 1) % nchunk) * CHS, wnext);
    }
    const float* wl = &wcur[kq * CO + seg];
    const float* tl = &tile[kq * ROWS * COLS + (wave * PR) * COLS + XSTEP * seg];
    if (KPC) {
      const float* tb = &tile[(wave * PR) * COLS + XSTEP * seg];
#pragma unroll
      for (int m = 0; m < NKP; ++m) {
        float av[MB], bv[PR][NX];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = wl[(4 * m) * CO + mb * 16];
#pragma unroll
        for (int pr = 0; pr < PR; ++pr)
#pragma unroll
          for (int nx = 0; nx < NX; ++nx) bv[pr][nx] = tb[boff[m] + pr * COLS + nx * XSEG];
#pragma unroll
        for (int pr = 0; pr < PR; ++pr)
#pragma unroll
          for (int nx = 0; nx < NX; ++nx)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
              acc[pr][nx][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[pr][nx], acc[pr][nx][mb], 0, 0, 0);
      }
      return;
    }
    // (an explicit operand pipeline as in conv3_mfma_kernel measured slower here: 2.27 -> 2.74 ms at 128^3)
#pragma unroll
    for (int r = 0; r < KS; ++r) {
#pragma unroll
      for (int s = 0; s < KW; ++s) {
#pragma unroll
        for (int ks = 0; ks < CHS / 4; ++ks) {
          float av[MB], bv[PR][NX];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) av[mb] = wl[((r * KW + s) * CHS + ks * 4) * CO + mb * 16];
#pragma unroll
          for (int pr = 0; pr < PR; ++pr)
#pragma unroll
            for (int nx = 0; nx < NX; ++nx) bv[pr][nx] = tl[(ks * 4) * ROWS * COLS + (pr + r) * COLS + nx * XSEG + s];
#pragma unroll
          for (int pr = 0; pr < PR; ++pr)
#pragma unroll
            for (int nx = 0; nx < NX; ++nx)
#pragma unroll
              for (int mb = 0; mb < MB; ++mb)
                acc[pr][nx][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[pr][nx], acc[pr][nx][mb], 0, 0, 0);
        }
      }
    }
  };
  for (int it = 0; it < niter; it += 2) {
    stage_body(it, wbuf0, wbuf1);
    if (it + 1 < niter) stage_body(it + 1, wbuf1, wbuf0);
  }
  if (PAIR && a.tail_w != nullptr) {
    // fused 1x1 tail: a lane holds channels (4 kq + r) % 8 of pixel x (kq 0, 1) or x + 1 (kq 2, 3); its four weighted values are
    // summed in the lane, the two halves of a pixel (lanes 16 apart) through a DPP row... the lanes sit in different rows of 16:
    // ds_swizzle / bpermute; the lane with even kq adds the bias and stores the ONE output channel
    float tw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tw[r] = a.tail_w[(4 * kq + r) % 8];
    const float tb = a.tail_b[0];
#pragma unroll
    for (int pr = 0; pr < PR; ++pr) {
      const int y = y0 + wave * PR + pr;
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[pr][nx][0][r];
          if (a.relu) v = fmaxf(v, 0.f);
          sum = fmaf(tw[r], v, sum);
        }
        const float other = __shfl_xor(sum, 16);             // (every lane takes part: no early exits above)
        const int x = x0 + nx * XSEG + XSTEP * seg + (kq >> 1);
        if ((kq & 1) == 0 && y < a.H && x < a.W)
          a.y[(size_t)b * vol + (size_t)z * plane + (size_t)y * a.W + x] = (sum + other) + tb;
      }
    }
    return;
  }
#pragma unroll
  for (int pr = 0; pr < PR; ++pr) {
    const int y = y0 + wave * PR + pr;
    if (y >= a.H) continue;
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int x = x0 + nx * XSEG + XSTEP * seg + (PAIR ? (kq >> 1) : 0);   // PAIR: rows 8..15 of the block are the +1 pixel
      if (x >= a.W) continue;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = PAIR ? (4 * kq + r) % 8 : mb * 16 + 4 * kq + r;
          if (co < a.cout) {
            float v = acc[pr][nx][mb][r];
            if (a.relu) v = fmaxf(v, 0.f);
            a.y[((size_t)b * a.cout + co) * vol + (size_t)z * plane + (size_t)y * a.W + x] = v;
          }
        }
      }
    }
  }
}

void launch_conv_mfma16(const ConvArgs& a, bool is3d, hipStream_t s) {
  const dim3 grid((a.W + 63) / 64, (a.H + 7) / 8, a.B * a.D);
  const int cin_pad = pad_to(a.cin, 4);
  if (pair_layer(a.cin, a.cout)) {
    if (is3d) conv5_mfma16_kernel<4, 1, true, true><<<grid, 256, 0, s>>>(a, cin_pad);
    else conv5_mfma16_kernel<4, 1, false, true><<<grid, 256, 0, s>>>(a, cin_pad);
  } else if (a.cin % 8 == 0 && a.cout <= 16) {
    if (is3d) conv5_mfma16_kernel<8, 1, true, false><<<grid, 256, 0, s>>>(a, cin_pad);
    else conv5_mfma16_kernel<8, 1, false, false><<<grid, 256, 0, s>>>(a, cin_pad);
  } else if (kpack_cin(a.cin, a.cout) == 3) {
    if (is3d) conv5_mfma16_kernel<4, 2, true, false, 3><<<grid, 256, 0, s>>>(a, cin_pad);
    else conv5_mfma16_kernel<4, 2, false, false, 3><<<grid, 256, 0, s>>>(a, cin_pad);
  } else if (kpack_cin(a.cin, a.cout) == 2) {
    if (is3d) conv5_mfma16_kernel<4, 2, true, false, 2><<<grid, 256, 0, s>>>(a, cin_pad);
    else conv5_mfma16_kernel<4, 2, false, false, 2><<<grid, 256, 0, s>>>(a, cin_pad);
  } else {
    if (is3d) conv5_mfma16_kernel<4, 2, true, false><<<grid, 256, 0, s>>>(a, cin_pad);
    else conv5_mfma16_kernel<4, 2, false, false><<<grid, 256, 0, s>>>(a, cin_pad);
  }
}

template <int CB, int PR, int CH = MF_CHUNK, int WPS = 2>
void launch_conv_mfma_t(const ConvArgs& a, bool is3d, hipStream_t s) {
  const dim3 grid((a.W + 31) / 32, (a.H + 4 * PR - 1) / (4 * PR), a.B * a.D * (a.cout / (CB * 32)));
  if (is3d) conv3_mfma_kernel<CB, PR, true, CH, WPS><<<grid, 256, 0, s>>>(a);
  else conv3_mfma_kernel<CB, PR, false, CH, WPS><<<grid, 256, 0, s>>>(a);
}

void launch_conv_mfma(const ConvArgs& a, bool is3d, hipStream_t s) {
  // small images (the quarter-resolution tower): shorter tiles (16 -> 8 -> 4 rows per block) until the launch has a
  // block for every CU slot.  (Measured and not kept: 128 output channels per wave; 8-row tiles with 4-channel stages at
  // 3 waves/SIMD, 2.5 % slower.)
  const int ngrp = a.cout / (a.cout % 64 == 0 ? 64 : 32);
  auto blocks = [&](int rows) { return (long)((a.W + 31) / 32) * ((a.H + rows - 1) / rows) * a.B * a.D * ngrp; };
  const int pr = blocks(16) >= 512 ? 4 : (blocks(8) >= 512 ? 2 : 1);
  if (a.cout % 64 == 0) {
    if (pr == 4) launch_conv_mfma_t<2, 4>(a, is3d, s); else if (pr == 2) launch_conv_mfma_t<2, 2>(a, is3d, s); else launch_conv_mfma_t<2, 1>(a, is3d, s);
  } else {
    if (pr == 4) launch_conv_mfma_t<1, 4>(a, is3d, s); else if (pr == 2) launch_conv_mfma_t<1, 2>(a, is3d, s); else launch_conv_mfma_t<1, 1>(a, is3d, s);
  }
}

// Winograd F(2x2,3x3) for the 3x3(x3) MFMA layers, when the launch fills the chip (one persistent workgroup per CU slot);
// false: nothing launched (small grid), the caller uses the direct kernel
bool launch_conv_wino(const ConvArgs& a, bool is3d, const float* wt, hipStream_t s) {
  if (a.cin % (4 * W3C) != 0 || a.cout % 32 != 0) return false;
  if (a.D != 1 && !is3d) return false;
  if ((size_t)16 * a.D * a.H * a.W * 4 >= 0xfffffff0u) return false;   // the epilogue's 32-bit offsets span a wave's 16 output channels
  // 64 output channels per workgroup where Cout allows it; 8-wave workgroups (two pixel groups share a stage's weights)
  // when the launch is large enough (measured at 1024^2: 64->32 297 -> 274 us against 4-wave workgroups)
  const int ncg = a.cout % 64 == 0 ? 2 : 1;
  int npg = 2 / ncg;
  if (ncg == 2 && (long)((a.W + 31) / 32) * ((a.H + 7) / 8) * a.B * a.D * (a.cout / 64) >= 512) npg = 2;   // 8 waves
  if (ncg == 1) npg = 2;
  assert(wino3_rw(a.cout) == 32 * ncg);
  const float* w3 = wt + (size_t)16 * (is3d ? 3 : 1) * a.cin * a.cout;   // the stage-contiguous image follows [dz][16][Cin][Cout]
  const int ntx = (a.W + 31) / 32, nty = (a.H + 4 * npg - 1) / (4 * npg);
  const long nt = (long)ntx * nty * a.B * a.D * (a.cout / (32 * ncg));
  static const int ncu = [] { int d = 0; hipDeviceProp_t pr; hipGetDevice(&d); hipGetDeviceProperties(&pr, d); return pr.multiProcessorCount; }();
  const int slots = ncu * (ncg * npg == 4 ? 1 : 2);      // 8-wave workgroups fill a CU's registers; two 4-wave ones fit
  if (nt < (npg * ncg == 4 ? 512 : 1024) || nt > 0x7fffffffl) return false;
  const int nwg = (int)(nt < slots ? nt : slots);
  if (is3d) {
    if (ncg == 2 && npg == 2) conv3_wino3_kernel<2, 2, true><<<nwg, 512, 0, s>>>(a, w3, ntx, nty, (int)nt);
    else if (ncg == 2) conv3_wino3_kernel<2, 1, true><<<nwg, 256, 0, s>>>(a, w3, ntx, nty, (int)nt);
    else conv3_wino3_kernel<1, 2, true><<<nwg, 256, 0, s>>>(a, w3, ntx, nty, (int)nt);
  } else if (ncg == 2 && npg == 2) conv3_wino3_kernel<2, 2><<<nwg, 512, 0, s>>>(a, w3, ntx, nty, (int)nt);
  else if (ncg == 2) conv3_wino3_kernel<2, 1><<<nwg, 256, 0, s>>>(a, w3, ntx, nty, (int)nt);
  else conv3_wino3_kernel<1, 2><<<nwg, 256, 0, s>>>(a, w3, ntx, nty, (int)nt);
  return true;
}

// mode: FNX_PRECISION_* (FP32_DIRECT: no Winograd, every layer a direct sum over its taps; BF16X6: conv3_wbf_kernel where it applies)
// tail: the packed layer of a following 1x1 convolution to one channel that the launch applies in its epilogue (y then has one
// channel); only for the PAIR 5x5 layer (fuses_tail)
inline bool fuses_tail(const ConvLayer& L, const ConvLayer& next) {
  return mfma16_layer(L) && pair_layer(L.cin, L.cout) && L.cout == 8 && next.k == 1 && next.cin == L.cout && next.cout == 1 && !next.relu;
}
void launch_conv(const ConvLayer& L, bool is3d, int mode, const float* packed, const PackedLayer& pl, const float* x, float* y,
                 int B, int D, int H, int W, hipStream_t s, const PackedLayer* tail = nullptr) {
  ConvArgs a{x, y, packed + pl.w_off, packed + pl.b_off, B, L.cin, L.cout, D, H, W, L.relu, L.cout / co_tile(L.cout),
             tail ? packed + tail->w_off : nullptr, tail ? packed + tail->b_off : nullptr};
  // (the MFMA kernels address a stage of 8 channel volumes through one 32-bit buffer range: 2^27 cells per sample at
  // most; beyond that -- 137 GB per 128-channel activation -- the direct kernel below still works)
  const bool direct = mode == FNX_PRECISION_FP32_DIRECT;
  if ((mode == FNX_PRECISION_BF16X6 || mode == FNX_PRECISION_BF16X3) && wbf_layer(L, is3d)) {
    ProfScope ps(FNX_PROF_CONV_BF16, s);
    const size_t nwino = (size_t)16 * (is3d ? 3 : 1) * L.cin * L.cout;
    const int nprod = mode == FNX_PRECISION_BF16X3 ? 3 : 6;
    if (launch_conv_wbf(a, is3d, (const unsigned*)(packed + pl.w_off + layer_weight_floats(L, is3d) + 2 * nwino), s, nprod)) {
      // six (three) bf16 MFMA products per Winograd-domain multiply (16 per 2x2 outputs and z tap)
      prof_add_work(FNX_PROF_CONV_BF16, (double)B * D * H * W * 2.0 * L.cin * L.cout * 4.0 * (is3d ? 3 : 1) * (double)nprod);
      return;
    }
  }
  if (mfma_layer(L) && (size_t)MF_CHUNK * D * H * W * 4 < 0xf0000000ull) {
    ProfScope ps(FNX_PROF_CONV_MFMA, s);
    const double px = (double)B * D * H * W, mac = 2.0 * L.cin * L.cout;
    // F(4x4): the default (256^3 CNN step 92.2 -> 81.0 ms; 1024^2 2.29 -> 2.14 ms: replayed graphs, alternating runs on one box)
    if ((mode == FNX_PRECISION_FP32_F4 || mode == FNX_PRECISION_FP32) && wino4_layer_(L, is3d) &&
        launch_conv_wino4(a, packed + pl.w_off + wino4_offset(L, is3d), is3d, s)) {
      prof_add_work(FNX_PROF_CONV_MFMA, px * mac * 2.25 * (is3d ? 3 : 1));  // 36 multiplies per 4x4 outputs (per z tap)
      return;
    }
    if (!direct && wino_layer(L, is3d) && launch_conv_wino(a, is3d, packed + pl.w_off + layer_weight_floats(L, is3d), s)) {
      prof_add_work(FNX_PROF_CONV_MFMA, px * mac * 4.0 * (is3d ? 3 : 1));   // 16 multiplies per 2x2 outputs (per z tap)
      return;
    }
    launch_conv_mfma(a, is3d, s);
    prof_add_work(FNX_PROF_CONV_MFMA, px * mac * layer_taps(L, is3d));
    return;
  }
  if (mfma16_layer(L)) {
    ProfScope ps(FNX_PROF_CONV_MFMA16, s); launch_conv_mfma16(a, is3d, s);
    prof_add_work(FNX_PROF_CONV_MFMA16, (double)B * D * H * W * 2.0 * L.cin * L.cout * layer_taps(L, is3d));
    return;
  }
  ProfScope ps(FNX_PROF_CONV_DIRECT, s);
  if (L.k == 3 && L.cout == 1 && L.cin % 4 == 0) {          // the towers' last layers: input channels split over the block's waves
    const dim3 grid((W + 63) / 64, (H + 3) / 4, B * D), block(64, 4);
    if (is3d) conv3_to1_kernel<true><<<grid, block, 0, s>>>(a); else conv3_to1_kernel<false><<<grid, block, 0, s>>>(a);
    return;
  }
  if (is3d) {
    if (L.k == 3) launch_conv_k<3, true>(a, s);
    else if (L.k == 5) launch_conv_k<5, true>(a, s);
    else launch_conv_k<1, true>(a, s);
  } else {
    if (L.k == 3) launch_conv_k<3, false>(a, s);
    else if (L.k == 5) launch_conv_k<5, false>(a, s);
    else launch_conv_k<1, false>(a, s);
  }
}

// torch upsample_{bi,tri}linear(align_corners=False): src = scale*(dst+0.5)-0.5, clamped at 0
__device__ __forceinline__ void src_index(int dst, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float scale = (float)in / (float)out;
  float sidx = scale * ((float)dst + 0.5f) - 0.5f;
  if (sidx < 0.f) sidx = 0.f;
  i0 = (int)sidx;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = sidx - (float)i0;
  l0 = 1.f - l1;
}

// x (B,C,Di,Hi,Wi) -> channels [c_off, c_off+C) of y (B,Ctot,Do,Ho,Wo)
// ZWin: either tensor may hold a window of planes of a deeper notional tensor (the nested crops of multiscale_forward_crop):
// the interpolation runs in the notional depths (full_in -> full_out), x holds notional planes [in_off, in_off + Di) and y the
// notional planes [out_off, out_off + Do); a source plane outside x's window is clamped into it (only planes nobody uses read
// such values).  {Di, 0, Do, 0} = whole tensors.
struct ZWin { int full_in, in_off, full_out, out_off; };
// One launch resamples up to TWO sources into consecutive channel ranges of y (the concatenations of multi_scale_net.py:121-125:
// [x resampled (2 channels), the coarser tower's output resampled (1 channel)]): source 0 fills channels [c_off, c_off + s0.C), source 1
// the next s1.C (s1.C = 0: one source).
struct RSrc { const float* x; int C, Di, Hi, Wi; ZWin zw; };
// One thread per output pixel (lanes along x, blockIdx.y = row, blockIdx.z = (sample, plane)): the interpolation indices and weights
// of a source are computed once and applied to all of its channels (the flat-index version paid an integer division chain and the
// weights per channel value: 22 us for the 3 M values of the full-resolution concat at 1024^2).
__device__ __forceinline__ void resize_source(const RSrc& S, float* __restrict__ y, int b, int k, int j, int i, int Do, int Ho, int Wo,
                                              int Ctot, int c0) {
  const int Di = S.Di, Hi = S.Hi, Wi = S.Wi;
  int x0, x1, y0, y1, z0, z1; float sx0, sx1, t0, t1, f0, f1;
  src_index(i, Wi, Wo, x0, x1, sx0, sx1);
  src_index(j, Hi, Ho, y0, y1, t0, t1);
  src_index(k + S.zw.out_off, S.zw.full_in, S.zw.full_out, z0, z1, f0, f1);
  z0 -= S.zw.in_off; z1 -= S.zw.in_off;
  z0 = z0 < 0 ? 0 : (z0 > Di - 1 ? Di - 1 : z0);
  z1 = z1 < 0 ? 0 : (z1 > Di - 1 ? Di - 1 : z1);
  const bool in_z = S.zw.full_in > 1 || S.zw.full_out > 1;
  const size_t vin = (size_t)Di * Hi * Wi, vout = (size_t)Do * Ho * Wo;
  const float* xi = S.x + (size_t)b * S.C * vin;
  float* yo = y + ((size_t)b * Ctot + c0) * vout + ((size_t)k * Ho + j) * Wo + i;
  const size_t o00 = ((size_t)z0 * Hi + y0) * Wi, o01 = ((size_t)z0 * Hi + y1) * Wi, o10 = ((size_t)z1 * Hi + y0) * Wi, o11 = ((size_t)z1 * Hi + y1) * Wi;
  for (int c = 0; c < S.C; ++c, xi += vin, yo += vout) {
    const float lo = t0 * (sx0 * xi[o00 + x0] + sx1 * xi[o00 + x1]) + t1 * (sx0 * xi[o01 + x0] + sx1 * xi[o01 + x1]);
    float v = lo;
    if (in_z) {
      const float hi = t0 * (sx0 * xi[o10 + x0] + sx1 * xi[o10 + x1]) + t1 * (sx0 * xi[o11 + x0] + sx1 * xi[o11 + x1]);
      v = f0 * lo + f1 * hi;
    }
    *yo = v;
  }
}
__global__ __launch_bounds__(256) void resize_kernel(RSrc s0, RSrc s1, float* __restrict__ y, int B, int Do, int Ho, int Wo, int Ctot,
                                                     int c_off) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int k = blockIdx.z % Do, b = blockIdx.z / Do;
  if (i >= Wo || j >= Ho) return;
  resize_source(s0, y, b, k, j, i, Do, Ho, Wo, Ctot, c_off);
  if (s1.C > 0) resize_source(s1, y, b, k, j, i, Do, Ho, Wo, Ctot, c_off + s0.C);
}

void launch_resize(const float* x, float* y, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int Ctot,
                   int c_off, hipStream_t s, const ZWin* zw = nullptr) {
  const ZWin whole{Di, 0, Do, 0};
  const dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, B * Do), block(64, 4);
  resize_kernel<<<grid, block, 0, s>>>(RSrc{x, C, Di, Hi, Wi, zw ? *zw : whole}, RSrc{nullptr, 0, 1, 1, 1, whole}, y, B, Do, Ho, Wo, Ctot, c_off);
}
// two sources -> channels [0, C0 + C1) of y in one launch
void launch_resize2(const RSrc& s0, const RSrc& s1, float* y, int B, int Do, int Ho, int Wo, hipStream_t s) {
  const dim3 grid((Wo + 63) / 64, (Ho + 3) / 4, B * Do), block(64, 4);
  resize_kernel<<<grid, block, 0, s>>>(s0, s1, y, B, Do, Ho, Wo, s0.C + s1.C, 0);
}

struct Sizes { int Dq, Hq, Wq, Dh, Hh, Wh; };
inline Sizes sizes(const GridDims& g, bool is3d) {
  Sizes z;
  z.Dq = is3d ? (int)(g.D * 0.25) : 1; z.Hq = (int)(g.H * 0.25); z.Wq = (int)(g.W * 0.25);
  z.Dh = is3d ? (int)(g.D * 0.5) : 1; z.Hh = (int)(g.H * 0.5); z.Wh = (int)(g.W * 0.5);
  return z;
}

}  // namespace

size_t scalenet_weight_floats(bool is3d) {
  size_t n = 0;
  for (int l = 0; l < N_LAYERS; ++l) n += layer_weight_floats(LAYERS[l], is3d) + LAYERS[l].cout;
  return n;
}

size_t scalenet_packed_bytes(bool is3d) {
  const PackedLayer last = packed_layer(N_LAYERS - 1, is3d);
  return al256((last.b_off + LAYERS[N_LAYERS - 1].cout + 64) * sizeof(float));
}

void scalenet_pack(bool is3d, const float* blob, void* packed, hipStream_t s) {
  size_t off = 0;
  float* pk = (float*)packed;
  for (int l = 0; l < N_LAYERS; ++l) {
    const ConvLayer& L = LAYERS[l];
    const PackedLayer pl = packed_layer(l, is3d);
    const size_t nw = layer_weight_floats(L, is3d);
    if (mfma_layer(L)) {
      pack_layer_mfma_kernel<<<64, 256, 0, s>>>(blob + off, blob + off + nw, pk + pl.w_off, pk + pl.b_off, L.cin, L.cout,
                                                layer_taps(L, is3d));
      if (wino_layer(L, is3d)) {
        const int kd = is3d ? 3 : 1;
        float* w2 = pk + pl.w_off + nw;                         // [dz][16][Cin][Cout]
        float* w3 = w2 + (size_t)16 * kd * L.cin * L.cout;      // stage-contiguous     (conv3_wino3_kernel)
        pack_layer_wino_kernel<<<64, 256, 0, s>>>(blob + off, w2, L.cin, L.cout, kd);
        repack_wino3_kernel<<<64, 256, 0, s>>>(w2, w3, L.cin, L.cout, kd, wino3_rw(L.cout));
        if (wbf_layer(L, is3d))                                 // FNX_PRECISION_BF16X6: three bf16 pieces, MFMA operand layout
          pack_wbf_kernel<<<256, 256, 0, s>>>(w2, (unsigned*)(w3 + (size_t)16 * kd * L.cin * L.cout), L.cin, L.cout, kd);
        if (wino4_layer_(L, is3d))                              // FNX_PRECISION_FP32_F4: the taps in lane order, stage-contiguous
          pack_layer_wino4g_kernel<<<64, 256, 0, s>>>(blob + off, pk + pl.w_off + wino4_offset(L, is3d), L.cin, L.cout, is3d ? 3 : 1);
      }
    }
    else if (mfma16_layer(L) && pair_layer(L.cin, L.cout))
      pack_layer_pair_kernel<<<64, 256, 0, s>>>(blob + off, blob + off + nw, pk + pl.w_off, pk + pl.b_off, L.cin, L.cout,
                                                is3d ? 5 : 1, pad_to(L.cin, 4));
    else if (mfma16_layer(L) && kpack_cin(L.cin, L.cout))
      pack_layer_kpack_kernel<<<64, 256, 0, s>>>(blob + off, blob + off + nw, pk + pl.w_off, pk + pl.b_off, L.cin, L.cout,
                                                 is3d ? 5 : 1, pad_to(L.cout, 16));
    else if (mfma16_layer(L))
      pack_layer_mfma16_kernel<<<64, 256, 0, s>>>(blob + off, blob + off + nw, pk + pl.w_off, pk + pl.b_off, L.cin, L.cout,
                                                  layer_taps(L, is3d), pad_to(L.cin, 4), pad_to(L.cout, 16));
    else
      pack_layer_kernel<<<64, 256, 0, s>>>(blob + off, blob + off + nw, pk + pl.w_off, pk + pl.b_off, L.cin, L.cout,
                                           layer_taps(L, is3d), co_tile(L.cout));
    off += nw + L.cout;
  }
}

// workspace: two ping-pong activation buffers of 128 channels at full resolution + the small tower I/O
size_t multiscale_ws_bytes(const GridDims& g, bool is3d) {
  const Sizes z = sizes(g, is3d);
  const size_t full = (size_t)g.B * g.DHW, half = (size_t)g.B * z.Dh * z.Hh * z.Wh, quart = (size_t)g.B * z.Dq * z.Hq * z.Wq;
  return 2 * al256(full * 128 * 4) + al256(full * 3 * 4) + al256(half * 3 * 4) + al256(quart * 2 * 4) +
         al256(half * 4) + al256(quart * 4);
}

void multiscale_forward(const GridDims& g, bool is3d, const void* packed, const float* x, float* p, int precision_mode, void* ws,
                        hipStream_t s) {
  const int none[4] = {0, 0, 0, 0};
  multiscale_forward_crop(g, is3d, packed, x, p, precision_mode, ws, s, none);
}

// The forward pass on NESTED z-crops (the z-slab driver's CNN projection, fnx_slab.hip): x covers g.D planes; the quarter-resolution
// tower runs on all of them, the half-resolution tower on planes [trim[2], g.D - trim[3]) and the full-resolution tower on
// [trim[0], g.D - trim[1]) (full-resolution plane counts, multiples of 4 so that every window starts on a plane of each coarser
// grid; trim[2] <= trim[0], trim[3] <= trim[1]); p receives g.D - trim[0] - trim[1] planes.  A tower whose window ends at an
// artificial face computes garbage within its receptive radius of that face (8 / 14 / 16 full-resolution planes for the full / half /
// quarter tower) -- the caller sizes the windows so that this never reaches what it uses (SlabSimulator.NET_TRIMS).  Everything
// else is the arithmetic of the untrimmed pass: the resampling runs in the untrimmed grids' coordinates.
void multiscale_forward_crop(const GridDims& g, bool is3d, const void* packed, const float* x, float* p, int precision_mode,
                             void* ws, hipStream_t s, const int trim[4]) {
  const int mode = precision_mode;
  const Sizes z = sizes(g, is3d);
  const size_t full = (size_t)g.B * g.DHW, half = (size_t)g.B * z.Dh * z.Hh * z.Wh, quart = (size_t)g.B * z.Dq * z.Hq * z.Wq;
  char* w = (char*)ws;
  float* bufA = (float*)w; w += al256(full * 128 * 4);
  float* bufB = (float*)w; w += al256(full * 128 * 4);
  float* in1 = (float*)w; w += al256(full * 3 * 4);
  float* in2 = (float*)w; w += al256(half * 3 * 4);
  float* xq = (float*)w; w += al256(quart * 2 * 4);
  float* c2 = (float*)w; w += al256(half * 4);
  float* c4 = (float*)w;
  const float* pk = (const float*)packed;
  auto tower = [&](int l0, int n, const float* in, float* out, int D, int H, int W) {
    const float* cur = in;
    for (int l = 0; l < n; ++l) {
      // the last 5x5 layer takes the final 1x1 into its epilogue (multi_scale_net.py:116): one launch, no 8-channel tensor
      const bool fuse = l + 2 == n && fuses_tail(LAYERS[l0 + l], LAYERS[l0 + l + 1]);
      float* dst = (l == n - 1 || fuse) ? out : ((l & 1) ? bufB : bufA);
      const PackedLayer tl = fuse ? packed_layer(l0 + l + 1, is3d) : PackedLayer{0, 0};
      launch_conv(LAYERS[l0 + l], is3d, mode, pk, packed_layer(l0 + l, is3d), cur, dst, g.B, D, H, W, s, fuse ? &tl : nullptr);
      if (fuse) break;
      cur = dst;
    }
  };
  // windows (planes of each tower's own grid)
  const int f_lo = trim[0], f_n = g.D - trim[0] - trim[1];              // full-resolution tower
  const int h_lo = trim[2] / 2, h_n = z.Dh - trim[2] / 2 - trim[3] / 2;  // half-resolution tower
  // multi_scale_net.py:119-126
  launch_resize(x, xq, g.B, 2, g.D, g.H, g.W, z.Dq, z.Hq, z.Wq, 2, 0, s);
  tower(0, 4, xq, c4, z.Dq, z.Hq, z.Wq);
  const ZWin x_to_h{g.D, 0, z.Dh, h_lo}, q_to_h{z.Dq, 0, z.Dh, h_lo};
  launch_resize2(RSrc{x, 2, g.D, g.H, g.W, x_to_h}, RSrc{c4, 1, z.Dq, z.Hq, z.Wq, q_to_h}, in2, g.B, h_n, z.Hh, z.Wh, s);
  tower(4, 6, in2, c2, h_n, z.Hh, z.Wh);
  const ZWin x_to_f{g.D, 0, g.D, f_lo}, h_to_f{z.Dh, h_lo, g.D, f_lo};
  launch_resize2(RSrc{x, 2, g.D, g.H, g.W, x_to_f}, RSrc{c2, 1, h_n, z.Hh, z.Wh, h_to_f}, in1, g.B, f_n, g.H, g.W, s);
  // convN_1 (6 layers) then final 1x1: 7 convs, the last one writes p
  tower(10, 7, in1, p, f_n, g.H, g.W);
}

// ---------------------------------------------------------------------------------------------------
// FluidNet.forward glue
// ---------------------------------------------------------------------------------------------------
namespace {

// _ScaleNet (model.py:8-23): unbiased std over C*D*H*W per sample, clamp(thr, inf).  Reproducible: every workgroup writes
// its own pair of fp64 partial sums (fixed element -> thread -> wave -> workgroup order, no atomics) and one workgroup per
// sample adds them up in index order.
constexpr int STD_MAXB = 1024;                              // partial pairs per sample
__device__ __forceinline__ void std_block_sum(double& s, double& ss, double (&red)[8]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); ss += __shfl_down(ss, off, 64); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * wave] = s; red[2 * wave + 1] = ss; }
  __syncthreads();
  s = (red[0] + red[2]) + (red[4] + red[6]);
  ss = (red[1] + red[3]) + (red[5] + red[7]);
}

__global__ __launch_bounds__(256) void std_partial_kernel(size_t n, const float* __restrict__ U, double* __restrict__ partial) {
  const int b = blockIdx.y;
  const float* u = U + (size_t)b * n;
  double s = 0.0, ss = 0.0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const double v = u[q];
    s += v; ss += v * v;
  }
  __shared__ double red[8];
  std_block_sum(s, ss, red);
  if (threadIdx.x == 0) {
    double* o = partial + 2 * ((size_t)b * STD_MAXB + blockIdx.x);
    o[0] = s; o[1] = ss;
  }
}

__global__ __launch_bounds__(256) void std_finish_kernel(int nb, size_t n, const double* __restrict__ partial, float thr,
                                                         float* __restrict__ scale) {
  const int b = blockIdx.x;
  const double* pb = partial + 2 * (size_t)b * STD_MAXB;
  double s = 0.0, ss = 0.0;
  for (int q = threadIdx.x; q < nb; q += 256) { s += pb[2 * q]; ss += pb[2 * q + 1]; }
  __shared__ double red[8];
  std_block_sum(s, ss, red);
  if (threadIdx.x == 0) {
    double var = (ss - s * s / (double)n) / (double)(n - 1);
    if (var < 0.0) var = 0.0;
    const float sd = (float)sqrt(var);
    scale[b] = sd < thr ? thr : sd;
  }
}

// x[b,0] = div/s ; x[b,1] = occupancy(flags) ; U /= s      (model.py:129-168)
__global__ __launch_bounds__(256) void pack_input_kernel(size_t n1, int nc, const float* __restrict__ div,
                                                         const float* __restrict__ flags,
                                                         const float* __restrict__ scale, float* __restrict__ U,
                                                         float* __restrict__ x) {
  const int b = blockIdx.y;
  const float s = scale[b];
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n1; q += (size_t)gridDim.x * 256) {
    x[((size_t)b * 2) * n1 + q] = div[(size_t)b * n1 + q] / s;
    const float f = flags[(size_t)b * n1 + q];
    x[((size_t)b * 2 + 1) * n1 + q] = f == FNX_FLUID ? 0.f : (f == FNX_OBST ? 1.f : f);
    for (int c = 0; c < nc; ++c) {
      const size_t o = ((size_t)b * nc + c) * n1 + q;
      U[o] = U[o] / s;
    }
  }
}

// the fused step's variant: x[b,0] = velocityDivergence(U, flags) / s (velocity_divergence.py:46-74, the divergence_kernel's
// expression) ; x[b,1] = occupancy(flags); U is left as it is -- the pass behind the net divides and multiplies
// (launch_post_projection with `scale`).  x fastest: i = q % W.
template <bool IS3D>
__global__ __launch_bounds__(256) void pack_div_kernel(GridDims g, const float* __restrict__ U, const float* __restrict__ flags,
                                                       const float* __restrict__ scale, float* __restrict__ x) {
  const int b = blockIdx.y;
  const size_t n1 = (size_t)g.DHW;
  constexpr int NC = IS3D ? 3 : 2;
  const float s = scale[b];
  const float* u = U + (size_t)b * NC * n1;
  // (planes of the compute window only: a z-slab's last ghost plane has no +1 neighbour in the array)
  const size_t q0 = (size_t)g.K0 * g.HW, q1 = (size_t)(g.K0 + g.KN) * g.HW;
  for (size_t q = q0 + (size_t)blockIdx.x * 256 + threadIdx.x; q < q1; q += (size_t)gridDim.x * 256) {
    const int i = (int)(q % g.W), j = (int)((q / g.W) % g.H), k = (int)(q / g.HW);
    const float f = flags[(size_t)b * n1 + q];
    float d = 0.f;
    if (!is_border<IS3D>(g, i, j, k)) {
      d = ((u[q] - u[q + 1]) + u[n1 + q]) - u[n1 + q + g.W];
      if (IS3D) d = d + (u[2 * n1 + q] - u[2 * n1 + q + g.HW]);
    }
    if (f == FNX_OBST) d = 0.f;
    x[((size_t)b * 2) * n1 + q] = d / s;
    x[((size_t)b * 2 + 1) * n1 + q] = f == FNX_FLUID ? 0.f : (f == FNX_OBST ? 1.f : f);
  }
}

__global__ __launch_bounds__(256) void unscale_kernel(size_t n1, int nc, const float* __restrict__ scale,
                                                      float* __restrict__ p, float* __restrict__ U) {
  const int b = blockIdx.y;
  const float s = scale[b];
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n1; q += (size_t)gridDim.x * 256) {
    p[(size_t)b * n1 + q] = p[(size_t)b * n1 + q] * s;
    for (int c = 0; c < nc; ++c) {
      const size_t o = ((size_t)b * nc + c) * n1 + q;
      U[o] = U[o] * s;
    }
  }
}

// input (B, nc+3, ...) = [p, U, flags, rho] -> U (B,nc,...), flags (B,1,...)
__global__ __launch_bounds__(256) void gather_input_kernel(size_t n1, int nc, const float* __restrict__ input,
                                                           float* __restrict__ U, float* __restrict__ flags) {
  const int b = blockIdx.y;
  const float* in = input + (size_t)b * (nc + 3) * n1;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n1; q += (size_t)gridDim.x * 256) {
    for (int c = 0; c < nc; ++c) U[((size_t)b * nc + c) * n1 + q] = in[(size_t)(1 + c) * n1 + q];
    flags[(size_t)b * n1 + q] = in[(size_t)(1 + nc) * n1 + q];
  }
}

inline dim3 bgrid(size_t n1, int B) {
  size_t blocks = (n1 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  return dim3((unsigned)blocks, B);
}

// ---- _ScaleNet on a z-slab (fnx_slab_step, method 1): the std over the WHOLE domain from per-rank sums ----
constexpr int WIN_BLOCKS = 256;
// block q of sample b: sum and sum of squares (fp64) of its share of the planes [k0, k1) of every channel, in a fixed order
__global__ __launch_bounds__(256) void window_sums_partial_kernel(GridDims g, int nc, int k0, int k1, const float* __restrict__ U,
                                                                  double* __restrict__ partial) {
  const int b = blockIdx.y;
  const size_t per = (size_t)(k1 - k0) * g.HW, n = per * nc;
  const float* u = U + (size_t)b * nc * g.DHW;
  double s = 0.0, ss = 0.0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const size_t c = q / per, r = q - c * per;
    const double v = (double)u[c * g.DHW + (size_t)k0 * g.HW + r];
    s += v; ss += v * v;
  }
  __shared__ double red[8];
  std_block_sum(s, ss, red);
  if (threadIdx.x == 0) { partial[2 * ((size_t)b * WIN_BLOCKS + blockIdx.x)] = s; partial[2 * ((size_t)b * WIN_BLOCKS + blockIdx.x) + 1] = ss; }
}
// One block per sample: the partials in index order -> this rank's (sum, sumsq), written as three floats per double (exact:
// 24 + 24 + 5 bits) into ITS slots of `red` ([rank][sample][2][3]); every other slot is zeroed.  A float all-reduce(sum) over the
// ranks then IS an all-gather (x + 0 + ... + 0 is exact in any order), with the communicator's existing entry point.
__global__ __launch_bounds__(256) void window_sums_encode_kernel(int rank, int nranks, int B, const double* __restrict__ partial,
                                                                 float* __restrict__ red) {
  const int b = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int q = threadIdx.x; q < WIN_BLOCKS; q += 256) { s += partial[2 * ((size_t)b * WIN_BLOCKS + q)]; ss += partial[2 * ((size_t)b * WIN_BLOCKS + q) + 1]; }
  __shared__ double sh[8];
  std_block_sum(s, ss, sh);
  for (int q = threadIdx.x; q < nranks * 6; q += 256) red[((size_t)(q / 6) * B + b) * 6 + q % 6] = 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = red + ((size_t)rank * B + b) * 6;
    const double v[2] = {s, ss};
    for (int k = 0; k < 2; ++k) {
      const float f1 = (float)v[k]; const double r1 = v[k] - (double)f1;
      const float f2 = (float)r1; const float f3 = (float)(r1 - (double)f2);
      o[3 * k] = f1; o[3 * k + 1] = f2; o[3 * k + 2] = f3;
    }
  }
}
// the ranks' sums added in RANK ORDER (the same bits on every rank), unbiased std over n elements, clamp(thr, inf)   (model.py:14-21)
__global__ void scale_from_sums_kernel(int nranks, int B, double n, float thr, const float* __restrict__ red, float* __restrict__ scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0, ss = 0.0;
  for (int r = 0; r < nranks; ++r) {
    const float* o = red + ((size_t)r * B + b) * 6;
    s += ((double)o[0] + (double)o[1]) + (double)o[2];
    ss += ((double)o[3] + (double)o[4]) + (double)o[5];
  }
  double var = (ss - s * s / n) / (n - 1.0);
  if (var < 0.0) var = 0.0;
  const float sd = (float)sqrt(var);
  scale[b] = sd < thr ? thr : sd;
}

}  // namespace

size_t window_sums_scratch_bytes(int B) { return sizeof(double) * 2 * WIN_BLOCKS * (size_t)B; }
void launch_window_sums_encode(const GridDims& g, int nc, int k0, int k1, const float* U, int rank, int nranks, double* partial,
                               float* red, hipStream_t s) {
  window_sums_partial_kernel<<<dim3(WIN_BLOCKS, g.B), 256, 0, s>>>(g, nc, k0, k1, U, partial);
  window_sums_encode_kernel<<<g.B, 256, 0, s>>>(rank, nranks, g.B, partial, red);
}
void launch_scale_from_sums(int nranks, int B, double n, float thr, const float* red, float* scale, hipStream_t s) {
  scale_from_sums_kernel<<<(B + 63) / 64, 64, 0, s>>>(nranks, B, n, thr, red, scale);
}
void launch_pack_div(const GridDims& g, bool is3d, const float* U, const float* flags, const float* scale, float* x, hipStream_t s) {
  size_t blocks = ((size_t)g.KN * g.HW + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks, g.B);
  if (is3d) pack_div_kernel<true><<<grid, 256, 0, s>>>(g, U, flags, scale, x);
  else pack_div_kernel<false><<<grid, 256, 0, s>>>(g, U, flags, scale, x);
}

size_t scale_std_scratch_bytes(int B) { return sizeof(double) * 2 * STD_MAXB * (size_t)B; }

void launch_scale_std(const GridDims& g, int nc, const float* U, float thr, double* partial, float* scale, hipStream_t s) {
  const size_t n = (size_t)nc * g.DHW;
  size_t nb = (n + 256 * 8 - 1) / (256 * 8);
  if (nb > STD_MAXB) nb = STD_MAXB;
  if (nb < 1) nb = 1;
  std_partial_kernel<<<dim3((unsigned)nb, g.B), 256, 0, s>>>(n, U, partial);
  std_finish_kernel<<<g.B, 256, 0, s>>>((int)nb, n, partial, thr, scale);
}

void launch_pack_input(const GridDims& g, int nc, const float* div, const float* flags, const float* scale, float* U,
                       float* x, hipStream_t s) {
  pack_input_kernel<<<bgrid(g.DHW, g.B), 256, 0, s>>>((size_t)g.DHW, nc, div, flags, scale, U, x);
}

void launch_unscale(const GridDims& g, int nc, const float* scale, float* p, float* U, hipStream_t s) {
  unscale_kernel<<<bgrid(g.DHW, g.B), 256, 0, s>>>((size_t)g.DHW, nc, scale, p, U);
}

void launch_gather_input(const GridDims& g, int nc, const float* input, float* U, float* flags, hipStream_t s) {
  gather_input_kernel<<<bgrid(g.DHW, g.B), 256, 0, s>>>((size_t)g.DHW, nc, input, U, flags);
}

size_t fluidnet_ws_bytes(const GridDims& g, bool is3d) {
  const size_t full = (size_t)g.B * g.DHW;
  return multiscale_ws_bytes(g, is3d) + al256(full * 4) /*flags*/ + al256(full * 4) /*div, or the net's p before the fused tail*/ + al256(full * 2 * 4) /*x*/ +
         al256(scale_std_scratch_bytes(g.B)) + al256(sizeof(float) * g.B);
}

}  // namespace fnx

#include "../../include/fluidnet_hip.h"

// FluidNet.forward after the channel split (model.py:120-227), U in place: U holds UDiv on entry and the
// projected velocity on exit.  ws needs fluidnet_ws_bytes() minus the flags copy.
namespace fnx {
int fluidnet_core(const FnxGrid* g, const void* packed, const float* flags, float thr, int precision_mode, float* p_out, float* U,
                  void* ws, void* stream, const FnxState* bcs) {
  const GridDims d = make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global);
  hipStream_t s = (hipStream_t)stream;
  const int nc = g->is3D ? 3 : 2;
  const size_t full = (size_t)g->B * d.DHW;
  char* w = (char*)ws;
  auto take = [&](size_t bytes) { void* r = w; w += (bytes + 255) & ~(size_t)255; return r; };
  float* div = (float*)take(full * 4);
  float* x = (float*)take(full * 2 * 4);
  double* partial = (double*)take(scale_std_scratch_bytes(g->B));
  float* scale = (float*)take(sizeof(float) * g->B);
  void* msws = w;
  if (bcs) {
    // The fused step (fnx_simulate_step, convnet, no flags_stick): three launches around the net instead of eight, the same
    // arithmetic per value.  The divergence goes straight into the net's input, U is not rewritten before the net, and
    // velocityUpdate, the un-normalisation, setWallBcs and the step's last setConstVals (simulate.py:168) are one pass.
    launch_scale_std(d, nc, U, thr, partial, scale, s);                          // model.py:129-144
    {
      size_t blocks = ((size_t)d.DHW + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      const dim3 grid((unsigned)blocks, g->B);
      if (g->is3D) pack_div_kernel<true><<<grid, 256, 0, s>>>(d, U, flags, scale, x);       // model.py:125-126, :146-168
      else pack_div_kernel<false><<<grid, 256, 0, s>>>(d, U, flags, scale, x);
    }
    float* pnet = div;                                                           // (the divergence buffer is free here)
    multiscale_forward(d, g->is3D, packed, x, pnet, precision_mode, msws, s);    // model.py:174-175
    const bool ubc = bcs->UBC && bcs->UBCInvMask, rbc = bcs->densityBC && bcs->densityBCInvMask;
    ProfScope ps(FNX_PROF_STAGE, s);
    launch_post_projection(d, g->is3D, pnet, U, bcs->density, flags, ubc ? bcs->UBC : nullptr, ubc ? bcs->UBCInvMask : nullptr,
                           rbc ? bcs->densityBC : nullptr, rbc ? bcs->densityBCInvMask : nullptr, s, bcs->bc_class,
                           bcs->density_bc_applied != 0, scale, p_out);          // model.py:213-226, simulate.py:168
    return fnx::launch_status();
  }
  if (int rc = fnx_velocity_divergence(g, U, flags, div, stream)) return rc;     // model.py:125-126
  launch_scale_std(d, nc, U, thr, partial, scale, s);                            // model.py:129-144
  launch_pack_input(d, nc, div, flags, scale, U, x, s);                          // model.py:146-168
  multiscale_forward(d, g->is3D, packed, x, p_out, precision_mode, msws, s);     // model.py:174-175
  if (int rc = fnx_velocity_update(g, p_out, U, flags, stream)) return rc;       // model.py:213-218
  launch_unscale(d, nc, scale, p_out, U, s);                                     // model.py:221-223
  if (int rc = fnx_set_wall_bcs(g, U, flags, stream)) return rc;                 // model.py:226
  return fnx::launch_status();
}
}  // namespace fnx

// ---------------------------------------------------------------------------------------------------
// C ABI (include/fluidnet_hip.h)
// ---------------------------------------------------------------------------------------------------

extern "C" {

size_t fnx_scalenet_weight_floats(int is3D) { return fnx::scalenet_weight_floats(is3D != 0); }
size_t fnx_scalenet_packed_bytes(int is3D) { return fnx::scalenet_packed_bytes(is3D != 0); }

int fnx_scalenet_pack(int is3D, const float* weights_blob, void* packed, void* stream) {
  if (!weights_blob || !packed) return fnx::set_error(FNX_EINVAL, "%s: null argument", __func__);
  fnx::scalenet_pack(is3D != 0, weights_blob, packed, (hipStream_t)stream);
  return fnx::launch_status();
}

static bool bad_precision(int m) { return m < FNX_PRECISION_FP32 || m > FNX_PRECISION_FP32_F2; }

int fnx_multiscale_forward(const FnxGrid* g, const void* packed, const float* x, float* p, int precision_mode, void* ws,
                           size_t ws_bytes, void* stream) {
  if (!g || !packed || !x || !p || !ws) return fnx::set_error(FNX_EINVAL, "%s: null argument", __func__);
  if (bad_precision(precision_mode)) return fnx::set_error(FNX_EINVAL, "unknown precision_mode %d (FNX_PRECISION_FP32, _FP32_DIRECT, _BF16X6 or _BF16X3)", precision_mode);
  if (g->H < 4 || g->W < 4 || (g->is3D && g->D < 4))
    return fnx::set_error(FNX_EINVAL, "%s: the three-scale net needs at least 4 cells per axis (D %d, H %d, W %d)", __func__, g->D, g->H, g->W);
  const GridDims d = make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global);
  if (ws_bytes < fnx::multiscale_ws_bytes(d, g->is3D)) return fnx::set_error(FNX_EWORKSPACE, "%s: workspace of %zu bytes is too small", __func__, ws_bytes);
  fnx::multiscale_forward(d, g->is3D, packed, x, p, precision_mode, ws, (hipStream_t)stream);
  return fnx::launch_status();
}

int fnx_multiscale_forward_crop(const FnxGrid* g, const void* packed, const float* x, float* p, int precision_mode,
                                const int trim[4], void* ws, size_t ws_bytes, void* stream) {
  if (!g || !packed || !x || !p || !ws || !trim) return fnx::set_error(FNX_EINVAL, "%s: null argument", __func__);
  if (bad_precision(precision_mode)) return fnx::set_error(FNX_EINVAL, "unknown precision_mode %d (FNX_PRECISION_FP32, _FP32_DIRECT, _BF16X6 or _BF16X3)", precision_mode);
  if (!g->is3D && (trim[0] | trim[1] | trim[2] | trim[3])) return fnx::set_error(FNX_EINVAL, "multiscale_forward_crop: z windows need a 3D grid");
  for (int a = 0; a < 4; ++a)
    if (trim[a] < 0 || trim[a] % 4) return fnx::set_error(FNX_EINVAL, "multiscale_forward_crop: trim[%d] = %d must be a non-negative multiple of 4", a, trim[a]);
  if (trim[2] > trim[0] || trim[3] > trim[1]) return fnx::set_error(FNX_EINVAL, "multiscale_forward_crop: the half-resolution window must contain the full-resolution one");
  if ((trim[0] | trim[1] | trim[2] | trim[3]) && (g->D % 4 || g->D - trim[0] - trim[1] < 4))
    return fnx::set_error(FNX_EINVAL, "multiscale_forward_crop: D = %d must be a multiple of 4 and leave at least 4 planes (trims %d + %d)", g->D, trim[0], trim[1]);
  if (g->H < 4 || g->W < 4 || (g->is3D && g->D < 4))
    return fnx::set_error(FNX_EINVAL, "%s: the three-scale net needs at least 4 cells per axis (D %d, H %d, W %d)", __func__, g->D, g->H, g->W);
  const GridDims d = make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global);
  if (ws_bytes < fnx::multiscale_ws_bytes(d, g->is3D)) return fnx::set_error(FNX_EWORKSPACE, "%s: workspace of %zu bytes is too small", __func__, ws_bytes);
  fnx::multiscale_forward_crop(d, g->is3D, packed, x, p, precision_mode, ws, (hipStream_t)stream, trim);
  return fnx::launch_status();
}

int fnx_fluidnet_forward(const FnxGrid* g, const void* packed, const float* input, float thr, float* p_out,
                         float* U_out, int precision_mode, void* ws, size_t ws_bytes, void* stream) {
  if (!g || !packed || !input || !p_out || !U_out || !ws) return fnx::set_error(FNX_EINVAL, "%s: null argument", __func__);
  if (bad_precision(precision_mode)) return fnx::set_error(FNX_EINVAL, "unknown precision_mode %d (FNX_PRECISION_FP32, _FP32_DIRECT, _BF16X6 or _BF16X3)", precision_mode);
  if (g->H < 4 || g->W < 4 || (g->is3D && g->D < 4))
    return fnx::set_error(FNX_EINVAL, "%s: the three-scale net needs at least 4 cells per axis (D %d, H %d, W %d)", __func__, g->D, g->H, g->W);
  const GridDims d = make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global);
  if (ws_bytes < fnx::fluidnet_ws_bytes(d, g->is3D)) return fnx::set_error(FNX_EWORKSPACE, "%s: workspace of %zu bytes is too small", __func__, ws_bytes);
  hipStream_t s = (hipStream_t)stream;
  const int nc = g->is3D ? 3 : 2;
  const size_t full = (size_t)g->B * d.DHW;
  float* flags = (float*)ws;                       // contiguous (B,1,..) copy of the flags channel
  void* rest = (char*)ws + ((full * 4 + 255) & ~(size_t)255);
  // model.py:104-119: split the channels
  fnx::launch_gather_input(d, nc, input, U_out, flags, s);
  return fnx::fluidnet_core(g, packed, flags, thr, precision_mode, p_out, U_out, rest, stream);
}

#ifdef W4_TIMELINE
int fnx_debug_w4_timeline(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fnx::w4_tl), sizeof(fnx::w4_tl)); }
#endif

}  // extern "C"
