// The whole Jacobi-method time step of a SMALL 2D grid (lib/simulate.py:28-171 at the reference's own 128 x 128 default) in ONE
// launch for gfx950.
//
// Why.  At 128^2 the step is five launches (advection forward, advection backward, staging + divergence, the 28-sweep solve,
// post-projection) of 4-5 us each around a 22-us solve: every launch but the solve is a kernel boundary (2.8 us,
// tools/ubench/grid_barrier.hip) plus ramp-up around ~1 us of work.  A grid-wide barrier INSIDE a launch costs 1.4 us when only 16
// workgroups take part (3.6 us with 64, ~10 us with 256: the arrivals serialise at one L2 line) -- so the step is run by FEW, LARGE
// workgroups: one 1024-thread workgroup per 32 x 32 block of the grid, the phases of the step separated by grid barriers.
//
//   phase A  forward passes of both advections      (sl_scalar_cell + sl_mac_cell_flat: fnx_advect_cells.h)
//   phase B  backward + correct + clamp passes      (sl_scalar_bwd_clamp_cell + sl_mac_bwd_clamp_cell_flat)
//   phase C  BCs, buoyancy, gravity, wall BCs, BCs, -div  (stage2d_div_cell: fnx_step_cells.h)
//   phase D  the Jacobi solve: the workgroup's 64 x 64 register tile (its 32 x 32 block + a halo of 16) runs up to 16 sweeps per
//            round (jacobi2d_wg_tile: fnx_jacobi2d_tile.h), rounds separated by a barrier, p ping-ponging between two arrays
//   phase E  velocityUpdate, wall BCs, BCs           (post_projection_cell)
//
// Every phase calls the per-cell / per-tile function the separate launches call, on the same arrays: the same bits
// (tests/test_small_step.py compares the two paths bit for bit).  Which thread computes which cell differs per phase (waves take
// 64-cell row segments in the cell phases, the solver's tile layout in phase D); the barrier is what makes that legal.
//
// The barrier: one 64-bit arrival counter in the caller's workspace.  A workgroup's ticket t = fetch_add(1) belongs to barrier
// number t / N (N workgroups; nobody can arrive at barrier n + 1 before everybody has arrived at barrier n), so it waits for the
// counter to reach (t / N + 1) * N: no reset, no generation argument -- the counter only has to be a multiple of N when the launch
// starts, which it is after a complete launch and after the zeroing the multi-launch path does (fnx_simulate_step).  Thread 0 of
// a workgroup arrives with agent-scope release (after the workgroup barrier: the L2 write-back covers every thread's stores) and
// polls with agent-scope acquire (L1 / L2 invalidate for the whole workgroup: it lives on one CU).  All workgroups must be
// resident at once: at most SMALL_MAX_WG of them, one per CU.
#include "fnx_device.h"
#include "fnx_kernels.h"

namespace {

#include "fnx_advect_cells.h"
#include "fnx_step_cells.h"
#include "fnx_jacobi2d_tile.h"

constexpr int SB = 32;                  // owned block edge
constexpr int S_RW = 4, S_NW = 16;      // the solver tile: 16 waves x 4 rows x 64 lanes
constexpr int S_HALO = (64 - SB) / 2;   // 16: sweeps per round at most
constexpr int SMALL_MAX_WG = 32;

__device__ __forceinline__ void grid_barrier(unsigned long long* ctr, unsigned n) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (t / n + 1ull) * n;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// -DFNX_SMALL_STAMPS (tools/small_step_phases.py): workgroup 0 leaves the 100-MHz clock at every phase boundary in the words
// behind the arrival counter (the counter's workspace slot is 256 bytes)
#ifdef FNX_SMALL_STAMPS
#define SMALL_STAMP(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.barrier[1 + (n)] = wall_clock64(); } while (0)
#else
#define SMALL_STAMP(n) do { } while (0)
#endif

template <bool SAMPLE_OUTSIDE>
__global__ __launch_bounds__(64 * S_NW) void small_step2d_kernel(GridDims g, fnx::SmallStep2D a, StepPtrs P, int tiles_x, int tiles_y) {
  __shared__ float edge[2][2][S_NW][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned nwg = gridDim.x;
  // cell phases: a wave takes 64-cell row segments (b, j, seg) round-robin
  const int nseg = (g.W + 63) >> 6, nunits = nseg * g.H * g.B;
  const int u0 = blockIdx.x * S_NW + w, ustep = (int)nwg * S_NW;

  SMALL_STAMP(0);
  // ---- A: forward passes
  for (int u = u0; u < nunits; u += ustep) {
    const int seg = u % nseg, rj = u / nseg;
    const CellId c{ rj / g.H, 0, rj % g.H, seg * 64 + lane, true };
    if (c.i < g.W) {
      sl_scalar_cell<false, false, SAMPLE_OUTSIDE>(g, c, a.dt, a.rho, a.U, a.flags, a.rho_fwd, a.cell);
      sl_mac_cell_flat<false>(g, c, a.dt, a.U, a.U, a.flags, a.U_fwd);
    }
  }
  SMALL_STAMP(1);
  grid_barrier(a.barrier, nwg);
  SMALL_STAMP(2);
  // ---- B: backward passes, MacCormack correction, clamp
  for (int u = u0; u < nunits; u += ustep) {
    const int seg = u % nseg, rj = u / nseg;
    const CellId c{ rj / g.H, 0, rj % g.H, seg * 64 + lane, true };
    if (c.i < g.W) {
      sl_scalar_bwd_clamp_cell<false, false, SAMPLE_OUTSIDE>(g, c, a.dt, a.half_s, a.rho, a.rho_fwd, a.cell, a.U, a.flags, nullptr, a.rho2);
      sl_mac_bwd_clamp_cell_flat<false>(g, c, a.dt, a.half_s, a.U, a.U_fwd, a.U, a.flags, a.U2);
    }
  }
  SMALL_STAMP(3);
  grid_barrier(a.barrier, nwg);
  SMALL_STAMP(4);
  // ---- C: the stages between advection and projection, and -div
  for (int u = u0; u < nunits; u += ustep) {
    const int seg = u % nseg, rj = u / nseg;
    const int i = seg * 64 + lane;
    if (i < g.W) stage2d_div_cell<true>(g, P, i, rj % g.H, rj / g.H, a.buoyancy, a.sx, a.sy, a.rho_star);
  }
  SMALL_STAMP(5);
  grid_barrier(a.barrier, nwg);
  SMALL_STAMP(6);
  // ---- D: the solve
  {
    const int t = blockIdx.x, bx = t % tiles_x, t1 = t / tiles_x, by = t1 % tiles_y, b = t1 / tiles_y;
    const int nr = (a.jacobi_iter + S_HALO - 1) / S_HALO, kmax = (a.jacobi_iter + nr - 1) / nr;
    int left = a.jacobi_iter;
    for (int r = 0; r < nr; ++r) {
      const int K = left < kmax ? left : kmax;
      left -= K;
      float* out = ((nr - 1 - r) & 1) ? a.p_tmp : a.p;                  // the last round writes p
      const float* in = ((nr - 1 - r) & 1) ? a.p : a.p_tmp;
      jacobi2d_wg_tile<S_RW, S_NW>(g, a.flags, a.div, in, out, r == 0 ? 1 : 0, b, bx * SB - S_HALO, by * SB - S_HALO, K, S_HALO,
                                   S_HALO + SB, S_HALO, S_HALO + SB, edge);
      SMALL_STAMP(7 + 2 * r);
      grid_barrier(a.barrier, nwg);
      SMALL_STAMP(8 + 2 * r);
    }
  }
  // ---- E: velocityUpdate, setWallBcs, setConstVals
  for (int u = u0; u < nunits; u += ustep) {
    const int seg = u % nseg, rj = u / nseg;
    const int i = seg * 64 + lane;
    if (i < g.W)
      post_projection_cell<false, false>(g, i, rj % g.H, 0, rj / g.H, a.p, a.U, a.rho, a.flags, a.UBC, a.UBCInvMask, a.rhoBC,
                                         a.rhoBCInvMask, a.cls, a.cls ? 1 : 0, nullptr, nullptr);
  }
  SMALL_STAMP(30);
}

}  // namespace

namespace fnx {

static void small_tiles(const GridDims& g, int& tx, int& ty) { tx = (g.W + SB - 1) / SB; ty = (g.H + SB - 1) / SB; }

bool small_step2d_fits(const GridDims& g) {
  if (g.D != 1 || g.W >= 65536 || g.H >= 65536) return false;
  int tx, ty; small_tiles(g, tx, ty);
  return (long)tx * ty * g.B <= SMALL_MAX_WG;
}

void launch_small_step2d(const GridDims& g, const SmallStep2D& a, hipStream_t s) {
  int tx, ty; small_tiles(g, tx, ty);
  StepPtrs P{a.U2, a.rho2, a.flags, a.UBC, a.UBCInvMask, a.rhoBC, a.rhoBCInvMask, a.U, a.rho, a.div, a.cls,
             a.grav, a.gx, a.gy, 0.f, 1, nullptr};
  const dim3 grid((unsigned)(tx * ty * g.B)), block(64 * S_NW);
  if (a.sample_outside) small_step2d_kernel<true><<<grid, block, 0, s>>>(g, a, P, tx, ty);
  else small_step2d_kernel<false><<<grid, block, 0, s>>>(g, a, P, tx, ty);
}

}  // namespace fnx
