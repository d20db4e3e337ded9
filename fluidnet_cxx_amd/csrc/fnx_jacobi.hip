// Jacobi pressure solve for gfx950 -- replaces solveLinearSystemJacobi (cpp/fluids_init.cpp:809-1004),
// which issues ~120 ATen ops plus a host sync per sweep.
//
// 2D: LDS temporal blocking.  A 1024^2 field is 4 MiB, so one sweep per launch would be launch-latency
// bound (2 us of HBM time per sweep vs ~2 us per kernel boundary).  Each workgroup instead loads a
// (TILE+2K)^2 halo tile of p, div and a 5-bit neighbour mask into LDS once, runs K sweeps entirely in LDS
// (the valid region shrinks by one ring per sweep) and writes back its TILE^2 centre: HBM traffic per K
// sweeps is ~1 read + 1 write of the field instead of K, and a 28-sweep solve is 4 launches.
// 3D: one sweep per launch, z-marching with coalesced 256-B rows (HBM-bound at 16 B/cell/sweep).
//
// Arithmetic per cell is exactly the reference's: ((((((n1+n2)+n3)+n4)+n5)+n6)+div)/denom, with
// obstacle neighbours replaced by the centre value (Neumann) and border cells held at 0 (Dirichlet).
#include "fnx_device.h"
#include "fnx_kernels.h"

namespace {

constexpr int TILE = 64;          // output tile edge (2D blocked kernel)
constexpr int TB_THREADS = 512;

// mask bits
constexpr unsigned M_CONT = 1, M_OL = 2, M_OR = 4, M_OD = 8, M_OU = 16;

template <int K>
__global__ __launch_bounds__(TB_THREADS) void jacobi2d_tb_kernel(GridDims g, const float* __restrict__ flags,
                                                                 const float* __restrict__ div,
                                                                 const float* __restrict__ p_in,
                                                                 float* __restrict__ p_out, bool from_zero,
                                                                 float* __restrict__ sumsq) {
  constexpr int R = TILE + 2 * K;               // region edge
  constexpr int NCELL = R * R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* pa = reinterpret_cast<float*>(smem);
  float* pb = pa + NCELL;
  float* dv = pb + NCELL;
  unsigned char* mk = reinterpret_cast<unsigned char*>(dv + NCELL);
  float* fl = pb;                               // flags staged in pb before the first sweep overwrites it

  const int b = blockIdx.z;
  const int x0 = blockIdx.x * TILE - K, y0 = blockIdx.y * TILE - K;
  const float* fb = flags + (size_t)b * g.DHW;
  const float* db = div + (size_t)b * g.DHW;
  const float* pi = p_in + (size_t)b * g.DHW;

  for (int idx = threadIdx.x; idx < NCELL; idx += TB_THREADS) {
    const int ry = idx / R, rx = idx - ry * R;
    const int x = x0 + rx, y = y0 + ry;
    const bool in = (x >= 0) & (x < g.W) & (y >= 0) & (y < g.H);
    const size_t o = (size_t)y * g.W + x;
    fl[idx] = in ? fb[o] : FNX_OBST;            // outside the grid: never read by a 'cont' cell anyway
    dv[idx] = in ? db[o] : 0.f;
    pa[idx] = (in && !from_zero) ? pi[o] : 0.f;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NCELL; idx += TB_THREADS) {
    const int ry = idx / R, rx = idx - ry * R;
    const int x = x0 + rx, y = y0 + ry;
    unsigned m = 0;
    const bool interior = (x >= 1) & (x <= g.W - 2) & (y >= 1) & (y <= g.H - 2);
    if (interior && rx >= 1 && rx <= R - 2 && ry >= 1 && ry <= R - 2 && fl[idx] != FNX_OBST) {
      m = M_CONT;
      if (fl[idx - 1] == FNX_OBST) m |= M_OL;
      if (fl[idx + 1] == FNX_OBST) m |= M_OR;
      if (fl[idx - R] == FNX_OBST) m |= M_OD;
      if (fl[idx + R] == FNX_OBST) m |= M_OU;
    }
    mk[idx] = (unsigned char)m;
  }
  __syncthreads();

  float* cur = pa;   // holds sweep s-1
  float* nxt = pb;
  float local = 0.f;
#pragma unroll 1
  for (int s = 1; s <= K; ++s) {
    const int lo = s, hi = R - 1 - s;           // cells [lo, hi]^2 are exact after sweep s
    const int n = hi - lo + 1;
    for (int idx = threadIdx.x; idx < n * n; idx += TB_THREADS) {
      const int ty = idx / n, tx = idx - ty * n;
      const int q = (lo + ty) * R + lo + tx;
      const unsigned m = mk[q];
      float v = 0.f;
      if (m & M_CONT) {
        const float pc = cur[q];
        const float n1 = (m & M_OL) ? pc : cur[q - 1];
        const float n2 = (m & M_OR) ? pc : cur[q + 1];
        const float n3 = (m & M_OD) ? pc : cur[q - R];
        const float n4 = (m & M_OU) ? pc : cur[q + R];
        float sum = n1 + n2;
        sum = sum + n3;
        sum = sum + n4;
        sum = sum + 0.f;
        sum = sum + 0.f;
        v = (sum + dv[q]) / 4.f;
      }
      nxt[q] = v;
      if (s == K && sumsq) {
        // residual contribution of this block's own centre tile only
        const int rx = lo + tx, ry = lo + ty;
        const int x = x0 + rx, y = y0 + ry;
        if (rx >= K && rx < K + TILE && ry >= K && ry < K + TILE && x < g.W && y < g.H) {
          const float d = v - cur[q];
          local += d * d;
        }
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  // write back the centre tile
  float* po = p_out + (size_t)b * g.DHW;
  for (int idx = threadIdx.x; idx < TILE * TILE; idx += TB_THREADS) {
    const int ty = idx / TILE, tx = idx - ty * TILE;
    const int x = x0 + K + tx, y = y0 + K + ty;
    if (x < g.W && y < g.H) po[(size_t)y * g.W + x] = cur[(K + ty) * R + K + tx];
  }
  if (sumsq) {
    // wave reduce then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sumsq[b], local);
  }
}

// Generic single sweep (3D, and any 2D shape): one thread per cell.
constexpr int BX = 64, BY = 4;

template <bool IS3D, bool QUIRKS>
__global__ __launch_bounds__(BX* BY) void jacobi_sweep_kernel(GridDims g, const float* __restrict__ flags,
                                                              const float* __restrict__ div,
                                                              const float* __restrict__ p_in,
                                                              float* __restrict__ p_out, bool from_zero,
                                                              float* __restrict__ sumsq) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
  const int bk = blockIdx.z;
  const int b = IS3D ? bk / g.D : bk, k = IS3D ? bk - b * g.D : 0;
  float d2 = 0.f;
  if (i < g.W && j < g.H) {
    const size_t o = (size_t)b * g.DHW + (size_t)k * g.HW + j * g.W + i;
    float v = 0.f;
    const float pc = from_zero ? 0.f : p_in[o];
    if (!is_border<IS3D>(g, i, j, k) && flags[o] != FNX_OBST) {
      if (from_zero) {
        float sum = 0.f + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f; sum = sum + 0.f;
        v = (sum + div[o]) / (IS3D ? 6.f : 4.f);
      } else {
        const float n1 = flags[o - 1] == FNX_OBST ? pc : p_in[o - 1];
        const float n2 = flags[o + 1] == FNX_OBST ? pc : p_in[o + 1];
        const float n3 = flags[o - g.W] == FNX_OBST ? pc : p_in[o - g.W];
        const float n4 = flags[o + g.W] == FNX_OBST ? pc : p_in[o + g.W];
        float sum = n1 + n2;
        sum = sum + n3;
        sum = sum + n4;
        if (IS3D) {
          // reference applies no Neumann substitution in z (fluids_init.cpp:935-943) -> QUIRKS
          const float n5 = (!QUIRKS && flags[o - g.HW] == FNX_OBST) ? pc : p_in[o - g.HW];
          const float n6 = (!QUIRKS && flags[o + g.HW] == FNX_OBST) ? pc : p_in[o + g.HW];
          sum = sum + n5;
          sum = sum + n6;
        } else {
          sum = sum + 0.f;
          sum = sum + 0.f;
        }
        v = (sum + div[o]) / (IS3D ? 6.f : 4.f);
      }
    }
    p_out[o] = v;
    const float d = v - pc;
    d2 = d * d;
  }
  if (sumsq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d2 += __shfl_down(d2, off, 64);
    if (threadIdx.x == 0) atomicAdd(&sumsq[b], d2);      // blockDim.x == 64: one wave per row of the block
  }
}

__global__ void residual_finish_kernel(int B, const float* __restrict__ sumsq, float* __restrict__ res) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) m = fmaxf(m, sqrtf(sumsq[b]));
    *res = m;
  }
}

__global__ __launch_bounds__(256) void residual_kernel(GridDims g, const float* __restrict__ a,
                                                       const float* __restrict__ bq, float* __restrict__ sumsq) {
  const int b = blockIdx.y;
  float acc = 0.f;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < (size_t)g.DHW; q += (size_t)gridDim.x * 256) {
    const float d = a[(size_t)b * g.DHW + q] - bq[(size_t)b * g.DHW + q];
    acc += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sumsq[b], acc);
}

template <int K>
void launch_tb(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out,
               bool from_zero, float* sumsq, hipStream_t s) {
  constexpr int R = TILE + 2 * K;
  constexpr size_t lds = (size_t)R * R * (3 * sizeof(float) + 1) + 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi2d_tb_kernel<K>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((g.W + TILE - 1) / TILE, (g.H + TILE - 1) / TILE, g.B);
  jacobi2d_tb_kernel<K><<<grid, TB_THREADS, lds, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
}

}  // namespace

namespace fnx {

constexpr int KMAX_2D = 8;

int jacobi_max_sweeps_per_launch(const GridDims& g, bool is3d) {
  if (is3d || g.D != 1) return 1;
  return KMAX_2D;
}

// nsweeps in [1, jacobi_max_sweeps_per_launch]; sumsq (B floats, pre-zeroed) receives ||p_n - p_{n-1}||^2 of the
// LAST sweep of this launch when non-null.
void launch_jacobi(const GridDims& g, bool is3d, bool quirks, const float* flags, const float* div, const float* p_in,
                   float* p_out, int nsweeps, bool from_zero, float* sumsq, hipStream_t s) {
  if (!is3d && g.D == 1 && nsweeps > 1 && nsweeps <= KMAX_2D) {
    switch (nsweeps) {
      case 2: launch_tb<2>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 3: launch_tb<3>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 4: launch_tb<4>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 5: launch_tb<5>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 6: launch_tb<6>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 7: launch_tb<7>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
      case 8: launch_tb<8>(g, flags, div, p_in, p_out, from_zero, sumsq, s); return;
    }
  }
  // generic path: exactly one sweep
  const dim3 grid((g.W + BX - 1) / BX, (g.H + BY - 1) / BY, g.B * g.D), block(BX, BY);
  if (is3d) {
    if (quirks) jacobi_sweep_kernel<true, true><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
    else jacobi_sweep_kernel<true, false><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
  } else {
    jacobi_sweep_kernel<false, false><<<grid, block, 0, s>>>(g, flags, div, p_in, p_out, from_zero, sumsq);
  }
}

void launch_residual_finish(int B, const float* sumsq, float* res, hipStream_t s) {
  residual_finish_kernel<<<1, 64, 0, s>>>(B, sumsq, res);
}

void launch_residual(const GridDims& g, const float* a, const float* b, float* sumsq, float* res, hipStream_t s) {
  int blocks = (g.DHW + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  residual_kernel<<<dim3(blocks, g.B), 256, 0, s>>>(g, a, b, sumsq);
  residual_finish_kernel<<<1, 64, 0, s>>>(g.B, sumsq, res);
}

}  // namespace fnx
