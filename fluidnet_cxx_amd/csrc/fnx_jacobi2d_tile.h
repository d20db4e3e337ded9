// The 2D Jacobi solver's tile code (solveLinearSystemJacobi, cpp/fluids_init.cpp:809-1004; see fnx_jacobi.hip): the lock-step row
// update and one workgroup tile of K sweeps.  Included INSIDE the anonymous namespace of the units that run it -- fnx_jacobi.hip
// and fnx_small.hip (the single-launch step of small 2D grids) -- after fnx_device.h.
// ---------------------------------------------------------------------------------------------------
// 2D: register-resident temporal blocking.
//   lane  <-> one grid column (64 columns per wave, OX = 64-2K of them are output)
//   regs  <-> RW rows of p and div per lane
//   x neighbours: DPP wave_shr:1 / wave_shl:1 (one VALU op, no LDS); y neighbours: the adjacent registers.
// Ring s of a tile goes stale at sweep s and the 2K-wide halo is recomputed by the neighbouring tiles.
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ float dpp_from_left(float v) {    // value held by lane-1 (0 into lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {   // value held by lane+1 (0 into lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned dpp_from_left_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned dpp_from_right_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

__device__ __forceinline__ float bfi_blend(int m, float a, float b) {   // m ? a : b for m in {0,-1}
  // The empty asm hides that m is a sign-extended bit, otherwise LLVM rewrites the blend into
  // shift+compare+cndmask (5 ops); as written it selects the single v_bfi_b32.
  asm("" : "+v"(m));
  return __builtin_bit_cast(float, (__builtin_bit_cast(int, a) & m) | (__builtin_bit_cast(int, b) & ~m));
}

// One Jacobi update of rows [R0, R0+N) of the register tile, the N rows advanced in LOCKSTEP: a single wave
// issues a dependent VALU op only every ~4.5 cycles but independent ones every ~2.3 (tools/ubench), and hipcc
// does not interleave the per-row chains on its own.  MASKED = obstacle-aware path.
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void jacobi_rows(float (&p)[V], const float (&d)[V], float& carry, const unsigned (&mL)[2],
                                            const unsigned (&mR)[2], const unsigned (&mD)[2], const unsigned (&mU)[2],
                                            const unsigned (&mC)[2], float (&delta)[N], float top = 0.f) {
  float pc[N], pl[N], pr[N], sum[N], v[N];
#pragma unroll
  for (int n = 0; n < N; ++n) pc[n] = p[R0 + n];
  const float up_last = (R0 + N < V) ? p[(R0 + N < V) ? R0 + N : 0] : top;
#pragma unroll
  for (int n = 0; n < N; ++n) pl[n] = dpp_from_left(pc[n]);
#pragma unroll
  for (int n = 0; n < N; ++n) pr[n] = dpp_from_right(pc[n]);
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n) pl[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mL[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], pl[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) pr[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mR[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], pr[n]);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = pl[n] + pr[n];
  float dn[N], un[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { dn[n] = n == 0 ? carry : pc[n > 0 ? n - 1 : 0]; un[n] = n == N - 1 ? up_last : pc[n < N - 1 ? n + 1 : 0]; }
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n) dn[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mD[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], dn[n]);
#pragma unroll
    for (int n = 0; n < N; ++n) un[n] = bfi_blend(__builtin_amdgcn_sbfe((int)mU[(R0 + n) >> 5], (R0 + n) & 31, 1), pc[n], un[n]);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + dn[n];
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + un[n];
  // the reference adds the two missing z neighbours as zeros: ((s + 0) + 0) == (s + 0) for every s (the first addition
  // already turns a -0 into +0, NaN and inf pass through), so ONE addition reproduces the bits of both
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) sum[n] = sum[n] + d[R0 + n];
#pragma unroll
  for (int n = 0; n < N; ++n) v[n] = sum[n] / 4.f;
  if (MASKED) {
#pragma unroll
    for (int n = 0; n < N; ++n)
      v[n] = __builtin_bit_cast(float, __builtin_bit_cast(int, v[n]) & __builtin_amdgcn_sbfe((int)mC[(R0 + n) >> 5], (R0 + n) & 31, 1));
  }
  carry = pc[N - 1];
#pragma unroll
  for (int n = 0; n < N; ++n) { delta[n] = v[n] - pc[n]; p[R0 + n] = v[n]; }
}

// ---------------------------------------------------------------------------------------------------
// 2D, workgroup tiles: NW waves stacked in y form ONE tile of 64 x (NW*RW) cells and hand each other their edge rows
// through LDS once per sweep (one barrier per sweep, two LDS row images alternating), so only the workgroup's outer ring
// is recomputed halo: the work per K sweeps is 64*NW*RW/((64-2K)*(NW*RW-2K)) times the field (1.8x at K = 8 with 8 waves
// of 8 rows; one independent 64 x (16+2K) tile per wave, the round-1 kernel, did 2.7x), and a wave's serial chain per
// sweep is RW rows whatever K.
// ---------------------------------------------------------------------------------------------------
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void wg_sweep_rows(float (&p)[V], const float (&d)[V], float& carry, float top,
                                              const unsigned (&mL)[2], const unsigned (&mR)[2], const unsigned (&mD)[2],
                                              const unsigned (&mU)[2], const unsigned (&mC)[2]) {
  if constexpr (R0 < V) {
    constexpr int M = (V - R0 >= N) ? N : (V - R0);
    float delta[M];
    jacobi_rows<V, R0, M, MASKED>(p, d, carry, mL, mR, mD, mU, mC, delta, top);
    wg_sweep_rows<V, R0 + M, N, MASKED>(p, d, carry, top, mL, mR, mD, mU, mC);
  }
}

// K (sweeps of the launch = halo width) is a launch argument: it only enters the tile origin, the output window and the trip
// count.  Small grids run DEEP launches (K up to 28: a 128^2 solve of 28 sweeps is ONE launch of 256 tiles with an 8 x 8
// output window each, one per CU, instead of four launches of 7 sweeps: the grid is launch-latency bound, the recomputed halo
// costs idle CUs nothing).
// the same, skipping row groups that lie wholly in the tile's stale rings (rows < s + 1 or > rows_total - 2 - s at sweep s)
template <int V, int R0, int N, bool MASKED>
__device__ __forceinline__ void wg_sweep_live(float (&p)[V], const float (&d)[V], float& carry, float top,
                                              const unsigned (&mL)[2], const unsigned (&mR)[2], const unsigned (&mD)[2],
                                              const unsigned (&mU)[2], const unsigned (&mC)[2], int tr0, int s, int rows_total) {
  if constexpr (R0 < V) {
    constexpr int M = (V - R0 >= N) ? N : (V - R0);
    const bool stale = (tr0 + R0 + M - 1 < s + 1) | (tr0 + R0 > rows_total - 2 - s);
    if (stale) {
      carry = p[R0 + M - 1];                               // what the next group sees below it: this group's last row, unchanged
    } else {
      float delta[M];
      jacobi_rows<V, R0, M, MASKED>(p, d, carry, mL, mR, mD, mU, mC, delta, top);
    }
    wg_sweep_live<V, R0 + M, N, MASKED>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, rows_total);
  }
}

// One tile: the 64 x (NW*RW) cells whose first cell is (x_org, y_org) of sample b run K sweeps (edge: [sweep parity][0: first
// row, 1: last row][wave][lane] in LDS) and the cells of tile columns [ox_lo, ox_hi) x tile rows [oy_lo, oy_hi) that lie in the
// grid are written to p_out -- the caller chooses a window at least K cells inside the tile.
template <int RW, int NW>
__device__ __forceinline__ void jacobi2d_wg_tile(const GridDims& g, const float* __restrict__ flags, const float* __restrict__ div,
                                                 const float* __restrict__ p_in, float* __restrict__ p_out, int from_zero, int b,
                                                 int x_org, int y_org, int K, int ox_lo, int ox_hi, int oy_lo, int oy_hi,
                                                 float (&edge)[2][2][NW][64]) {
  constexpr int V = RW;
  constexpr int NI = 4;
  static_assert(V <= 32, "tile shape");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = x_org + lane, y0 = y_org + w * RW;                   // this wave's first row
  const bool xin = (x >= 0) & (x < g.W), xint = (x >= 1) & (x <= g.W - 2);
  const size_t base = (size_t)b * g.DHW;
  const int xc = x < 0 ? 0 : (x > g.W - 1 ? g.W - 1 : x);

  float p[V], d[V];
  unsigned ob = 0, cont = 0;
  auto is_ob = [&](int y) {                              // obstacle (or outside the grid) at (x, y)
    const bool yin = (y >= 0) & (y < g.H);
    const int yc = y < 0 ? 0 : (y > g.H - 1 ? g.H - 1 : y);
    const float f = (flags + base + (size_t)yc * g.W)[xc];
    return !(xin & yin) | (f == FNX_OBST);
  };
#pragma unroll
  for (int r = 0; r < V; ++r) {
    const int y = y0 + r;
    const bool yin = (y >= 0) & (y < g.H);
    const int yc = y < 0 ? 0 : (y > g.H - 1 ? g.H - 1 : y);
    const size_t row = base + (size_t)yc * g.W;
    const float f = (flags + row)[xc];
    const float dv = (div + row)[xc];
    float pv = 0.f;
    if (!from_zero) pv = (p_in + row)[xc];
    const bool in = xin & yin;
    d[r] = in ? dv : 0.f;
    p[r] = in ? pv : 0.f;
    const bool isob = !in | (f == FNX_OBST);
    ob |= (unsigned)isob << r;
    cont |= (unsigned)(xint & (y >= 1) & (y <= g.H - 2) & !isob) << r;
  }
  const unsigned ob_below = is_ob(y0 - 1), ob_above = is_ob(y0 + V);
  const unsigned obL = dpp_from_left_u(ob), obR = dpp_from_right_u(ob);
  const unsigned obD = (ob << 1) | ob_below, obU = (ob >> 1) | (ob_above << (V - 1));
  // the workgroup's outermost ring is never evaluated (its neighbours are not in the tile)
  if (w == 0) cont &= ~1u;
  if (w == NW - 1) cont &= ~(1u << (V - 1));
  if (lane == 0 || lane == 63) cont = 0;
  unsigned mL[2] = {obL, 0}, mR[2] = {obR, 0}, mD[2] = {obD, 0}, mU[2] = {obU, 0}, mC[2] = {cont, 0};
  // output rows of this wave: tile rows [oy_lo, oy_hi) that lie in the grid
  int out_lo = oy_lo - w * RW, out_hi = oy_hi - w * RW;
  if (out_lo < 0) out_lo = 0;
  if (out_hi > V) out_hi = V;
  if (out_hi > g.H - y0) out_hi = g.H - y0;
  const bool lane_ok = (lane >= ox_lo) & (lane < ox_hi) & xin;
  constexpr unsigned ALL = V < 32 ? ((1u << V) - 1) : ~0u;
  const unsigned ring = ALL & ~((w == 0 ? 1u : 0u) | (w == NW - 1 ? 1u << (V - 1) : 0u));
  const bool edge_lane = (lane == 0) | (lane == 63);
  const bool plain = edge_lane | ((cont == ring) & (((obL | obR | obD | obU) & ring) == 0));
  const bool all_plain = __all(plain);
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    float* e = &edge[s & 1][0][0][0];
    e[w * 64 + lane] = p[0];
    e[NW * 64 + w * 64 + lane] = p[V - 1];
    __syncthreads();
    float carry = w > 0 ? e[NW * 64 + (w - 1) * 64 + lane] : 0.f;           // last row of the wave below
    const float top = w < NW - 1 ? e[(w + 1) * 64 + lane] : 0.f;            // first row of the wave above
    // Sweep s + 1 makes the tile's ring s stale (its neighbours are not in the tile); nothing inside the output window ever
    // reads a stale cell, so a group of rows that lies wholly in rings 0..s is not evaluated (wave-uniform): a deep launch
    // (K = 28) skips 40 % of its row updates, a K = 8 launch the outer groups of its first and last wave.
    const int tr0 = w * RW;                                // first tile row of this wave
    if (all_plain) wg_sweep_live<V, 0, NI, false>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, NW * RW);
    else {
      asm volatile("" : "+v"(mL[0]), "+v"(mR[0]), "+v"(mD[0]), "+v"(mU[0]), "+v"(mC[0]));
      wg_sweep_live<V, 0, NI, true>(p, d, carry, top, mL, mR, mD, mU, mC, tr0, s, NW * RW);
    }
  }
  if (lane_ok) {
#pragma unroll
    for (int r = 0; r < V; ++r)
      if (r >= out_lo && r < out_hi) (p_out + base + (size_t)(y0 + r) * g.W)[x] = p[r];
  }
}

