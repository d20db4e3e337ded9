// Host-side launch wrappers of the gfx950 kernels (internal to libfluidnet_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include "fnx_device.h"

namespace fnx {

// sets the calling thread's fnx_last_error() message and returns `code` (fnx_api.hip)
int set_error(int code, const char* fmt, ...);

// event-pair timing of kernel classes (fnx_api.hip); no-ops unless fnx_profile_enable(1)
bool prof_begin(int tag, hipStream_t s);      // true: a roctx range was pushed (prof_end pops only then)
void prof_end(int tag, hipStream_t s, bool pushed);
void prof_add_work(int tag, double amount);   // adds to the class's work counter while a recorded launch of it is open
struct ProfScope {
  int tag; hipStream_t s; bool pushed;
  ProfScope(int t, hipStream_t st) : tag(t), s(st), pushed(prof_begin(t, st)) {}
  ~ProfScope() { prof_end(tag, s, pushed); }
};

// advection (fnx_advect.hip)
void launch_sl_scalar(const GridDims& g, bool is3d, bool quirks, bool sample_outside, float dt, const float* src,
                      const float* U, const float* flags, float* dst, int* cell_out, hipStream_t s);
void launch_sl_scalar_bwd_clamp(const GridDims& g, bool is3d, bool quirks, bool sample_outside, float dt, float half_s,
                                const float* src, const float* fwd, const int* cell_in, const float* U,
                                const float* flags, const float* box, float* dst, hipStream_t s);
// `fix`: 4 * advect_fix_words(g) 64-bit words (3D: the tile kernels' fix-up bitmaps; unused in 2D)
size_t advect_fix_words(const GridDims& g);
void launch_advect_fused(const GridDims& g, const GridDims& gfwd, bool is3d, bool quirks, bool sample_outside, float dt,
                         float half_s, const float* rho, const float* U, const float* flags, float* rho_fwd, int* cell,
                         float* U_fwd, float* box, float* rho_dst, float* U_dst, unsigned long long* fix, hipStream_t s,
                         int plan = 0, int what = 3);   // plan: FNX_ADVECT_PLAN_*; what: 3 both, 1 density only, 2 velocity only (tiles only)
// which kernels launch_advect_fused takes: 0 = one thread per cell; 2D: 2 = LDS tiles; 3D: bit 0 / bit 2 = z-marching tiles for the
// forward / backward pass
int advect_tile_plan(const GridDims& g, const GridDims& gfwd, bool is3d, bool quirks, int plan);
void launch_box_minmax(const GridDims& g, bool sample_outside, const float* src, const float* flags, float* box,
                       hipStream_t s);
void launch_sl_mac(const GridDims& g, bool is3d, bool quirks, float dt, const float* src, const float* U,
                   const float* flags, float* dst, hipStream_t s);
void launch_sl_mac_bwd_clamp(const GridDims& g, bool is3d, bool quirks, float dt, float half_s, const float* orig,
                             const float* fwd, const float* U, const float* flags, float* dst, hipStream_t s);

// stencils (fnx_stencils.hip)
void launch_divergence(const GridDims& g, bool is3d, const float* U, const float* flags, float* div, hipStream_t s);
void launch_velocity_update(const GridDims& g, bool is3d, const float* p, float* U, const float* flags, hipStream_t s);
void launch_add_gravity(const GridDims& g, bool is3d, float* U, const float* flags, float fx, float fy, float fz,
                        hipStream_t s);
void launch_correct_scalar(const GridDims& g, bool is3d, float half_dt, float* src, const float* div, const float* flags,
                           hipStream_t s);
void launch_add_viscosity(const GridDims& g, const float* Uin, float* Uout, const float* flags, float coef,
                          hipStream_t s);
void launch_add_buoyancy(const GridDims& g, bool is3d, bool quirks, float* U, const float* flags, const float* rho,
                         float sx, float sy, float sz, float rho_star, hipStream_t s);
void launch_set_wall_bcs(const GridDims& g, bool is3d, float* U, const float* flags, hipStream_t s);
void launch_set_wall_bcs_stick(const GridDims& g, const float* Uin, float* Uout, const float* flags, const float* stick,
                               hipStream_t s);
void launch_set_const_vals(size_t n, float* x, const float* bc, const float* inv_mask, hipStream_t s);
void launch_flags_to_occupancy(size_t n, const float* flags, float* occ, hipStream_t s);
void launch_max_abs(size_t n, const float* x, float* out, hipStream_t s);
void launch_empty_domain(const GridDims& g, bool is3d, float* flags, int bnd, hipStream_t s);
void launch_create_cylinder(const GridDims& g, float* flags, float cx, float cy, float r2, hipStream_t s);
void launch_create_box2d(const GridDims& g, float* flags, float x0, float x1, float y0, float y1, hipStream_t s);
void launch_get_centered(const GridDims& g, bool is3d, const float* U, float* out, hipStream_t s);
void launch_divergence_bwd(const GridDims& g, bool is3d, const float* gdiv, const float* flags, float* gU, hipStream_t s);
void launch_velocity_update_bwd(const GridDims& g, bool is3d, const float* gout, const float* flags, float* gU, float* gp,
                                hipStream_t s);

// fused step stages (fnx_step.hip)
void launch_pre_projection(const GridDims& g, bool is3d, bool quirks, const float* U_adv, const float* rho_adv,
                           const float* flags, const float* UBC, const float* UBCInvMask, const float* rhoBC,
                           const float* rhoBCInvMask, float* U, float* rho, float* div, bool buoyancy, float sx,
                           float sy, float sz, float rho_star, bool wall_bcs, hipStream_t s,
                           const unsigned char* cls = nullptr, const float* gravity = nullptr, bool second_bcs = true,
                           int div_k_end = 0);
// gravity: 3 host floats (gravity * dt) or null; second_bcs = false leaves out the setConstVals of simulate.py:133; 3D with `div`:
// the divergence of the staged field is written for the planes [g.K0, div_k_end) of the staged range (the staged value of a cell's
// +1 neighbours is re-derived in the same pass; their advected inputs must be valid)
// periodic patches of the Jacobi branch (simulate.py:121-128, :157-164); `save`: periodic_save_bytes(g), mode 0 = save the
// source row / column before the post-projection pass, 1 = write the destinations after it
void launch_periodic_pre(const GridDims& g, bool is3d, const float* U_adv, const float* UBC, const float* UBCInvMask, float* U,
                         bool px, bool py, hipStream_t s);
size_t periodic_save_bytes(const GridDims& g);
void launch_periodic_post(const GridDims& g, bool is3d, float* U, float* save, const float* UBC, const float* UBCInvMask,
                          bool px, bool py, int mode, hipStream_t s);
void launch_post_projection(const GridDims& g, bool is3d, const float* p, float* U, float* rho, const float* flags,
                            const float* UBC, const float* UBCInvMask, const float* rhoBC, const float* rhoBCInvMask,
                            hipStream_t s, const unsigned char* cls = nullptr, bool rho_bc_applied = false,
                            const float* scale = nullptr, float* p_scaled = nullptr);
// scale (B device floats) / p_scaled: the tail of FluidNet.forward in the same pass (u = U / s into the update, u * s and
// p_scaled = p * s out of it); p_scaled must not alias p (a cell reads p of its -1 neighbours)
void launch_bc_classify(const GridDims& g, bool is3d, const float* UBC, const float* UBCInvMask, const float* rhoBC,
                        const float* rhoBCInvMask, unsigned char* cls, hipStream_t s);

// Jacobi (fnx_jacobi.hip)
// 2D: `nsweeps` sweeps (1..jacobi_max_sweeps_per_launch) from p_in into p_out; from_zero: p_in is all zeros and is not read
void launch_jacobi(const GridDims& g, const float* flags, const float* div, const float* p_in, float* p_out, int nsweeps,
                   bool from_zero, hipStream_t s);
int  jacobi_max_sweeps_per_launch(const GridDims& g, bool is3d, int total_sweeps);   // total_sweeps: what the solve still has to run
// 3D: flags -> 7-bit neighbour mask (once per solve), then z-marching passes of two sweeps (one for an odd remainder)
size_t jacobi3d_mask_bytes(const GridDims& g);   // bytes of the `mask` allocation of the 3D launches below (byte mask + the same bytes in row groups of four)
void launch_jacobi3d_mask(const GridDims& g, bool quirks, const float* flags, unsigned char* mask, hipStream_t s);
// kb/ke: restrict the OUTPUT to planes [kb, ke) (0,0 = all planes); inputs are read from kb-1 (kb-2 for x2) on
void launch_jacobi3d(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                     bool from_zero, hipStream_t s, int kb = 0, int ke = 0);
// mirror of a two-sweep launch's output: planes [k[r], k[r] + n) of plane range r also go to out[r][q] + sample * bstride (floats),
// q = (*sel[r] + 1) & 1 read on the device (sel NULL: q = 0)
struct JacobiMirror { float* out[2][2]; const unsigned* sel[2]; int k[2]; int n; unsigned long long bstride; unsigned long long* clock; };
bool jacobi3d_mirror_ok(const GridDims& g, int np, bool two_ranges, bool from_zero, int lay);
void launch_jacobi3d_x2(const GridDims& g, const unsigned char* mask, const float* div, const float* p_in, float* p_out,
                        hipStream_t s, int kb = 0, int ke = 0, bool from_zero = false, int kb2 = -1, int lay = 0,
                        const JacobiMirror* mirror = nullptr);
bool jacobi3d_quad_ok(const GridDims& g);   // may two-sweep passes hand each other p in the row-quad layout (lay bits 0 / 1 = p_in / p_out)?
// Reproducible residual (no atomics): per sample b the squared differences of a[b*per_sample + first + q] - b[...] (b == null:
// zeros), q < count, summed in a fixed order in fp64 through `partials` (residual_scratch_bytes(B)); sumsq (B floats, may be
// null) receives the sums, res (1 float, may be null) max_b sqrt(sum)
size_t residual_scratch_bytes(int B);
void launch_residual(int B, size_t per_sample, size_t first, size_t count, const float* a, const float* b, double* partials,
                     float* sumsq, float* res, hipStream_t s);
void launch_residual_root(int B, const float* sumsq, float* res, hipStream_t s);   // res = max_b sqrt(sumsq[b])

}  // namespace fnx
