/*
 * fluidnet_hip.h -- C ABI of libfluidnet_hip.so, the MI355X (gfx950) fluid time-step library.
 *
 * This is the drop-in boundary for fluidnet_cxx's operator surface.  Each entry point cites the
 * reference interface it replaces (paths relative to the reference repo root).  The torch
 * cpp-extension `fluidnet_cpp` (fluidnet_cxx_amd/csrc/fluidnet_cpp.cpp) binds these 1:1 behind
 * the reference's three pybind names (pytorch/lib/fluid/cpp/fluids_init.cpp:1009-1014) plus the
 * operators the reference implements in Python; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every tensor is a caller-owned DEVICE pointer on the current HIP device, fp32, contiguous
 *    (B,C,D,H,W) with x fastest; 2D == D=1 with 2 velocity channels, 3D has 3.  `flags` is fp32
 *    holding Manta cell types exactly as the reference passes them (cpp/cell_type.h:7-18).
 *  - nothing is allocated, retained or freed; scratch comes from the caller's `ws` buffer
 *    (size from fnx_workspace_bytes); outputs are caller-allocated.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t, may be NULL = default stream) and
 *    returns without synchronising, except fnx_jacobi with p_tol > 0 (the reference's own
 *    per-sweep host test, fluids_init.cpp:973).
 *  - return value: FNX_OK or an FNX_E* code; fnx_last_error() gives the message (thread local).
 *    There is no CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef FLUIDNET_HIP_H
#define FLUIDNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNX_ABI_VERSION 19

enum {
  FNX_OK = 0,
  FNX_EINVAL = 1,    /* bad argument (shape, bnd != 1, null pointer, max_iter < 1 ...) */
  FNX_EMETHOD = 2,   /* "Advection method not supported" (cpp/advect_type.cpp:14) */
  FNX_EWORKSPACE = 3,/* workspace too small */
  FNX_EHIP = 4,      /* HIP runtime error / no device */
  FNX_ECFL = 5,      /* z-slab step: max |U| dt > 1 cell (the decomposition's ghost widths rest on CFL <= 1) */
  FNX_ECOMM = 6      /* communicator error (RCCL call failed, librccl not loadable, peer missing) */
};

/* Cell types, cpp/cell_type.h:7-18 and lib/fluid/cell_type.py:5-14 */
enum { FNX_TYPE_NONE = 0, FNX_TYPE_FLUID = 1, FNX_TYPE_OBSTACLE = 2, FNX_TYPE_EMPTY = 4,
       FNX_TYPE_INFLOW = 8, FNX_TYPE_OUTFLOW = 16, FNX_TYPE_OPEN = 32, FNX_TYPE_STICK = 128 };

/* Advection methods, cpp/advect_type.cpp:5-16 ("eulerFluidNet", "maccormackFluidNet") */
enum { FNX_ADVECT_EULER = 0, FNX_ADVECT_MACCORMACK = 1 };

typedef struct FnxGrid {
  int B, D, H, W;   /* batch, depth (1 for 2D), height, width */
  int is3D;         /* 0: U has 2 channels and D must be 1; 1: U has 3 channels */
  int ref_quirks;   /* 3D only: 1 reproduces the reference's 3D defects bit-for-bit,
                       0 (default) gives the intended 3D semantics.  Ignored in 2D. */
  /* z-slab decomposition (3D, multi-GPU): the arrays hold planes [z_offset, z_offset + D) of a domain that is
     D_global planes deep; only the domain-border test uses them.  0 / 0 = a whole domain (the default). */
  int z_offset, D_global;
  /* Compute window (z-slab driver): the cell-parallel operators -- advection, the BC/buoyancy/wall/divergence stage,
     velocity update and the stand-alone stencils -- only produce local planes [k_begin, k_end); planes outside keep
     their old contents.  They still READ outside the window.  MacCormack advection evaluates its forward pass on the
     window widened by 2 planes, which covers what the backward pass and the clamp read while |U dt| <= 1 cell (the
     same bound the ghost width of the decomposition rests on).  0 / 0 = all planes (the default).  The Jacobi entry
     points take their own plane range. */
  int k_begin, k_end;
} FnxGrid;

/* Workspace sizing. */
enum { FNX_OP_ADVECT_SCALAR = 0, FNX_OP_ADVECT_VEL = 1, FNX_OP_JACOBI = 2, FNX_OP_STEP = 3, FNX_OP_FLUIDNET = 4,
       FNX_OP_ADVECT_STEP = 5 };
size_t fnx_workspace_bytes(const FnxGrid* g, int op);

const char* fnx_last_error(void);
int fnx_abi_version(void);
/* name of the HIP device the library would run on, or NULL (and FNX_EHIP set) when there is none */
const char* fnx_device_name(void);

/* advectScalar: pybind `advect_scalar`, cpp/fluids_init.cpp:265-382 (wrapper cpp/advection.py:14-66).
 * dst must not alias src.  MacCormack in 3D default semantics, and in 2D from 1.5 M cells, runs the LDS tile kernels of the fused pair
 * below with its density part only (ABI 19; FNX_OP_ADVECT_SCALAR grew by the tiles' fix-up bitmaps); quirks mode, Euler and small 2D
 * grids run one thread per cell.  Same bits either way. */
int fnx_advect_scalar(const FnxGrid* g, float dt, const float* src, const float* U, const float* flags,
                      float* dst, int method, int bnd, int sample_outside_fluid, float maccormack_strength,
                      void* ws, size_t ws_bytes, void* stream);

/* advectVel: pybind `advect_vel`, cpp/fluids_init.cpp:656-807 (wrapper cpp/advection.py:68-118).
 * orig may alias U (self-advection); dst must alias neither.  Self-advection (orig == U, what lib/simulate.py:93 passes unless
 * viscosity > 0) takes the LDS tile kernels under the same conditions as fnx_advect_scalar (ABI 19); same bits either way. */
int fnx_advect_vel(const FnxGrid* g, float dt, const float* orig, const float* U, const float* flags,
                   float* dst, int method, int bnd, float maccormack_strength,
                   void* ws, size_t ws_bytes, void* stream);

/* The two advections of one time step (lib/simulate.py:75-93): density by U and U by itself, both MacCormack, both
 * from the OLD U, as one fused pair of launches (forward passes together, backward/clamp passes together).  Results
 * are bit-identical to fnx_advect_scalar + fnx_advect_vel.  dst buffers must not alias the inputs. */
int fnx_advect_step(const FnxGrid* g, float dt, const float* density, const float* U, const float* flags,
                    float* density_dst, float* U_dst, int sample_outside, float strength, void* ws, size_t ws_bytes,
                    void* stream);
/* The same with the kernel family chosen by the caller instead of by grid size (same bits either way): */
enum { FNX_ADVECT_PLAN_AUTO = 0,      /* what fnx_advect_step does: LDS tile kernels in 3D and on 2D grids of >= 1.5 M cells */
       FNX_ADVECT_PLAN_TILES = 1,     /* LDS tile kernels + fix-up launches (CFL < 1 is their fast path, any CFL is correct) */
       FNX_ADVECT_PLAN_CELLS = 2,     /* one thread per cell, gathers from global memory */
       FNX_ADVECT_PLAN_TILES_SPLIT = 3 };  /* TILES with the 3D backward pass of fnx_advect_step as two marches (density, velocity) instead of the
                                              fused one (same bits, 1 % slower: a timing aid) */
int fnx_advect_step_plan(const FnxGrid* g, float dt, const float* density, const float* U, const float* flags,
                         float* density_dst, float* U_dst, int sample_outside, float strength, int plan, void* ws,
                         size_t ws_bytes, void* stream);
/* ... and the stand-alone operators with the kernel family chosen by the caller (ABI 19; TILES is honoured where tiles exist: MacCormack,
 * default 3D semantics or 2D, fnx_advect_vel with orig == U -- elsewhere the call runs one thread per cell): */
int fnx_advect_scalar_plan(const FnxGrid* g, float dt, const float* src, const float* U, const float* flags,
                           float* dst, int method, int bnd, int sample_outside_fluid, float maccormack_strength, int plan,
                           void* ws, size_t ws_bytes, void* stream);
int fnx_advect_vel_plan(const FnxGrid* g, float dt, const float* orig, const float* U, const float* flags,
                        float* dst, int method, int bnd, float maccormack_strength, int plan,
                        void* ws, size_t ws_bytes, void* stream);

/* velocityDivergence, lib/fluid/velocity_divergence.py:4-74 */
int fnx_velocity_divergence(const FnxGrid* g, const float* U, const float* flags, float* div, void* stream);

/* solveLinearSystemJacobi: pybind `solve_linear_system`, cpp/fluids_init.cpp:809-1004.
 * p: output (B,1,D,H,W).  residual: DEVICE pointer to 1 float (max over batch of ||p - p_prev||_2 of the
 * last sweep) or NULL.  iters_done: HOST pointer or NULL.  p_tol <= 0 never synchronises.
 * The residual is reproducible: squared differences summed in a fixed order in fp64 (no atomics), so the sweep at which a
 * p_tol > 0 solve stops is the same run to run.  With residual != NULL the last sweep runs as its own launch (both
 * iterates must be in memory); fnx_simulate_step passes NULL. */
int fnx_jacobi(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual,
               float p_tol, int max_iter, int* iters_done, void* ws, size_t ws_bytes, void* stream);
/* The same with `verbose` of the reference (cpp/fluids_init.cpp:968-987): "Jacobi iteration N: residual R" on stdout after
 * EVERY sweep and the two termination messages -- one sweep per launch and one host synchronisation per sweep. */
int fnx_jacobi_verbose(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual,
                       float p_tol, int max_iter, int* iters_done, void* ws, size_t ws_bytes, void* stream);
/* ||a - b||_2 per sample over the compute window of g (b == NULL: zeros), reproducible as above: sumsq (B DEVICE floats, may be
 * NULL) receives the sums of squares, residual (1 DEVICE float, may be NULL) max_b sqrt(sum).  ws: B * 4096 bytes.  What the
 * z-slab driver all-reduces over the ranks for p_tol > 0. */
int fnx_residual(const FnxGrid* g, const float* a, const float* b, float* sumsq, float* residual, void* ws, size_t ws_bytes, void* stream);

/* `nsweeps` more Jacobi sweeps starting from the pressure already in `p` (in place).  Same per-sweep arithmetic as
 * fnx_jacobi (cpp/fluids_init.cpp:858-994); used by the z-slab driver, which exchanges ghost planes of p between
 * blocks of sweeps.  No residual. */
int fnx_jacobi_sweeps(const FnxGrid* g, const float* flags, const float* div, float* p, int nsweeps,
                      void* ws, size_t ws_bytes, void* stream);
/* Same, for callers that keep `ws` alive between calls on the SAME flags: bit 0 of reuse_mask skips rebuilding the 3D
 * neighbour mask that an earlier call (without it) left in `ws`.  Bit 1: the solve starts from p = 0 -- `p` is not read
 * (it need not be zeroed), the first launch is the from-zero instantiation: a whole solve without the residual of
 * fnx_jacobi, which is what the z-slab driver runs on a single rank. */
int fnx_jacobi_sweeps_ex(const FnxGrid* g, const float* flags, const float* div, float* p, int nsweeps,
                         void* ws, size_t ws_bytes, int reuse_mask, void* stream);

/* One pass of 1 or 2 sweeps (3D) from p_in into p_out restricted to the output planes [k_begin, k_end)
 * (0,0 = all); explicit buffers, no ping-pong.  p_in == NULL: the pressure is 0 everywhere (first pass of a solve).  Lets the z-slab driver compute the planes its neighbours need
 * first, start the ghost exchange, and compute the interior while the exchange is in flight.  `ws` as for
 * fnx_jacobi_sweeps_ex (holds the neighbour mask). */
int fnx_jacobi_pass(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                    int nsweeps, int k_begin, int k_end, void* ws, size_t ws_bytes, int reuse_mask, void* stream);
/* The same for TWO disjoint plane ranges of equal length in one launch: [k_begin, k_end) and [k_begin2, k_begin2 +
 * k_end - k_begin) (k_begin2 < 0: one range).  The slab driver's edge parts at the lower and the upper internal face. */
int fnx_jacobi_pass2(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                     int nsweeps, int k_begin, int k_end, int k_begin2, void* ws, size_t ws_bytes, int reuse_mask,
                     void* stream);
/* The same with the pressure arrays in the solver's ROW-QUAD layout p[b][k][j/4][i][j%4] (a plane is the same H*W floats as
 * in the row layout p[b][k][j][i]: ghost-plane exchanges do not care): layout bit 0 = p_in, bit 1 = p_out is row-quad.
 * Two-sweep passes only, on grids fnx_jacobi_quad_ok() accepts (3D, H % 4 == 0).  A chain of passes reads the pressure with
 * 3 instead of 8 and writes it with 1 instead of 4 vector-memory instructions per step of the march when its passes hand each
 * other this layout; the first pass of a solve has no input, the last one writes rows (layout 1).  fnx_jacobi and
 * fnx_jacobi_sweeps do this internally. */
int fnx_jacobi_quad_ok(const FnxGrid* g);
int fnx_jacobi_pass_layout(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                           int nsweeps, int k_begin, int k_end, int k_begin2, int layout, void* ws, size_t ws_bytes,
                           int reuse_mask, void* stream);

/* A two-sweep pass that ALSO stores some of the planes it finishes to a second destination ("mirror"): the planes
 * [k_first[r], k_first[r] + planes) of plane range r (0: [k_begin, k_end); 1: the second range) go to out[r][q] + sample * sample_stride
 * floats, plane after plane, each in the layout of p_out.  The destination is double buffered: q = (*slot_select[r] + 1) & 1, read ON
 * THE DEVICE when the launch starts (slot_select[r] NULL: q = 0) -- whose turn it is may depend on how often a captured graph has been
 * replayed.  What the z-slab driver's last edge part of a sweep block uses to write the planes its neighbours need next straight into
 * their mapped mailboxes (FnxSlabComm.direct_begin): the transport then has nothing left to copy on the sending side.  layout 0 (rows in
 * and out) or 3 (row-quad in and out); p_in != NULL; only launches fnx_jacobi_pass_mirror_ok accepts (every (tile, plane chunk) wave
 * resident at once). */
typedef struct FnxPlaneMirror {
  float* out[2][2]; const unsigned* slot_select[2]; int k_first[2]; int planes; size_t sample_stride;
  unsigned long long* start_clock;   /* optional DEVICE word: the launch stores the device clock (wall_clock64) there when it starts */
} FnxPlaneMirror;
int fnx_jacobi_pass_mirror_ok(const FnxGrid* g, int planes_per_range, int two_ranges, int layout);
int fnx_jacobi_pass_mirror(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out, int k_begin,
                           int k_end, int k_begin2, int layout, const FnxPlaneMirror* mirror, void* ws, size_t ws_bytes,
                           int reuse_mask, void* stream);

/* velocityUpdate (in place on U), lib/fluid/velocity_update.py:6-162 */
int fnx_velocity_update(const FnxGrid* g, const float* p, float* U, const float* flags, void* stream);

/* addBuoyancy (in place on U), lib/fluid/source_terms.py:6-116.  gravity: 3 HOST floats. */
int fnx_add_buoyancy(const FnxGrid* g, float* U, const float* flags, const float* density,
                     const float gravity[3], float rho_star, float dt, void* stream);

/* addGravity (in place on U), lib/fluid/source_terms.py:122-219.  gravity: 3 HOST floats. */
int fnx_add_gravity(const FnxGrid* g, float* U, const float* flags, const float gravity[3], float dt, void* stream);

/* correctScalar (in place on src), lib/fluid/cpp/advection.py:9-12: src += (dt * 0.5) * src * div on fluid cells. */
int fnx_correct_scalar(const FnxGrid* g, float dt, float* src, const float* div, const float* flags, void* stream);

/* addViscosity, lib/fluid/viscosity.py:7-70 (2D only, like the reference).  The reference updates U in place from
 * a fully evaluated right-hand side; here U_in is the old field and U_out (a distinct buffer) receives the result. */
int fnx_add_viscosity(const FnxGrid* g, float dt, const float* U_in, float* U_out, const float* flags,
                      float viscosity, void* stream);

/* setWallBcs (in place on U), lib/fluid/set_wall_bcs.py:4-86 */
int fnx_set_wall_bcs(const FnxGrid* g, float* U, const float* flags, void* stream);

/* setWallBcsStick, lib/fluid/set_wall_bcs_stick.py:5-157 (no-slip walls: `flags_stick` is a copy of flags with the
 * no-slip obstacle cells set to 128, cylinder.py:76).  2D only.  As shipped the reference function raises NameError
 * on three unbound names (:62 ff.); with them bound its 2D body runs, and that is what this reproduces bit for bit
 * (tests/golden/stick.npz).  The reference updates U in place from gathered copies; here U_in is the old field and
 * U_out (a distinct buffer) receives the result. */
int fnx_set_wall_bcs_stick(const FnxGrid* g, const float* U_in, float* U_out, const float* flags,
                           const float* flags_stick, void* stream);

/* setConstVals (in place), lib/simulate.py:4-26.  Either triple may be NULL (key absent in batch_dict). */
int fnx_set_const_vals(const FnxGrid* g, float* U, const float* UBC, const float* UBCInvMask,
                       float* density, const float* densityBC, const float* densityBCInvMask, void* stream);

/* flagsToOccupancy, lib/fluid/flags_to_occupancy.py:6-19 */
int fnx_flags_to_occupancy(const FnxGrid* g, const float* flags, float* occupancy, void* stream);

/* max |x| over a (B,channels,D,H,W) field into *out_max (DEVICE float; overwritten).  No reference counterpart: the
 * z-slab driver's CFL guard (max |U| dt <= 1 is what its ghost widths rest on) and any caller that wants the CFL number
 * without a host-side reduction.  NaNs are ignored. */
int fnx_max_abs(const FnxGrid* g, const float* x, int channels, float* out_max, void* stream);

/* emptyDomain (writes flags), lib/fluid/util.py:5-47 */
int fnx_empty_domain(const FnxGrid* g, float* flags, int boundary_width, void* stream);

/* createCylinder, lib/fluid/geometry_utils.py:4-34: cells with (x - center_x)^2 + (y - center_y)^2 <= radius^2 (integer
 * cell indices against the float centre, fp32 arithmetic as the reference's tensor expression evaluates it) become
 * obstacles on every z plane.  The scalars are doubles like the python floats the reference takes: radius is squared in
 * double and then rounded to fp32, the centre is rounded to fp32 (what the tensor iterator does).  In place on flags. */
int fnx_create_cylinder(const FnxGrid* g, float* flags, double center_x, double center_y, double radius, void* stream);
/* createBox2D, lib/fluid/geometry_utils.py:36-63: cells with x0 <= x < x1 and y0 <= y < y1 become obstacles.  The
 * reference's body cannot run (it tests Y against y1 twice and names an undefined mask, :59-62); this is the box its
 * docstring describes.  In place on flags. */
int fnx_create_box2d(const FnxGrid* g, float* flags, float x0, float x1, float y0, float y1, void* stream);
/* getCentered, lib/fluid/grid.py:7-32: (B,2|3,D,H,W) MAC velocity -> (B,3,D,H,W) cell-centred velocity (the z channel
 * is 0 in 2D); what the drivers' field dumps plot (plume.py:243,336). */
int fnx_get_centered(const FnxGrid* g, const float* U, float* centered, void* stream);

/* One whole time step, lib/simulate.py:28-171 (the keys simulate() reads from mconf). */
typedef struct FnxStepParams {
  float dt;                   /* mconf['dt'] */
  float maccormack_strength;  /* mconf['maccormackStrength'] */
  int   sample_outside_fluid; /* mconf['sampleOutsideFluid'] */
  float buoyancy_scale;       /* mconf['buoyancyScale'] ; <= 0 skips addBuoyancy */
  float gravity_vec[3];       /* mconf['gravityVec'] x,y,z (scaled by -buoyancyScale inside, simulate.py:101-105) */
  float operating_density;    /* mconf['operatingDensity'] */
  float p_tol;                /* mconf['pTol'] */
  int   jacobi_iter;          /* mconf['jacobiIter'] */
  int   method;               /* 0 = 'jacobi', 1 = 'convnet' */
  float normalize_threshold;  /* mconf['normalizeInputThreshold'] (convnet) */
  int   precision_mode;       /* convnet: FNX_PRECISION_FP32 (0, the default), _FP32_DIRECT, _BF16X6 or _BF16X3, see fnx_multiscale_forward */
  int   static_flags;         /* promises about the previous fnx_simulate_step on this workspace (no reference key; every
                                 reference simulation keeps its flags and BC arrays fixed):
                                 bit 0: `flags` is unchanged -> the 3D Jacobi solver reuses the obstacle mask it left there;
                                 bit 1: UBC / UBCInvMask / densityBC / densityBCInvMask are unchanged -> the BC stages use a
                                        1-byte-per-cell class map kept in the workspace (see FnxState.bc_class);
                                 bit 2: that class map was already built by an earlier call with bit 1 set */
  /* Optional stages of lib/simulate.py (all off when zero; fnx_slab_step refuses them): */
  float viscosity;            /* mconf['viscosity'] > 0 (2D only, like addViscosity): the velocity advected is
                                 addViscosity(U.clone()), advected by U (simulate.py:66-69, :85-93) */
  float gravity_scale;        /* mconf['gravityScale'] > 0: addGravity(gravityVec * -gravityScale) after the buoyancy, only with
                                 a density (simulate.py:107-114) */
  int   correct_scalar;       /* mconf['correctScalar']: density += dt/2 * density * div(U) on fluid cells after its
                                 advection (simulate.py:79-81, cpp/advection.py:9-12) */
  int   periodic;             /* bit 0: mconf has BOTH 'periodic-x' and 'periodic-y'; bit 1 / bit 2: their values.  Method 0
                                 only (simulate.py:121-128, :157-164): U[:,1,:,:,1] = U_temp[:,1,:,:,W-1],
                                 U[:,0,:,1] = U_temp[:,0,:,H-1] after each setWallBcs, U_temp the field before it */
} FnxStepParams;

typedef struct FnxState {
  float* p;        /* (B,1,D,H,W) in/out */
  float* U;        /* (B,2|3,D,H,W) in/out */
  float* density;  /* (B,1,D,H,W) in/out, may be NULL (simulate.py:71-83: no 'density' key) */
  const float* flags;
  const float* UBC; const float* UBCInvMask;                 /* may be NULL */
  const float* densityBC; const float* densityBCInvMask;     /* may be NULL */
  const void*  net;  /* packed ScaleNet weights from fnx_scalenet_pack (method 1), else NULL */
  /* Optional (B,1,D,H,W) bytes from fnx_bc_classify, or NULL: bit 0 = setConstVals is x*1+0 for every velocity component
   * of the cell, bit 1 = the same for the density.  The BC stages then skip the 8 BC loads of such a cell (32 of its 64-68
   * bytes) and apply t = x*1, t + 0 directly: same bits.  Only valid while the four BC arrays do not change. */
  const unsigned char* bc_class;
  /* Optional (B,1,1,H,W) copy of flags with the no-slip cells set to 128 ('flags_stick' in batch_dict, cylinder.py:76), or
   * NULL.  Method 1 (convnet), 2D only: setWallBcsStick before the second setConstVals and after the net
   * (simulate.py:129-130, :165-166).  Method 0 ignores it, as the reference does. */
  const float* flags_stick;
  /* Optional promise to fnx_post_projection (0 = none): `density` already went through setConstVals with these BC arrays
   * since it was last written -- fnx_pre_projection leaves it that way.  On a cell of bc_class whose density BC is the
   * identity (x*1 + 0) a further setConstVals then changes no bit (the first one has already turned a -0 into +0), and the
   * pass neither reads nor writes the density there (8 of the ~41 bytes a cell of that pass moves).  Ignored without bc_class. */
  int density_bc_applied;
} FnxState;

/* Classifies every cell for FnxState.bc_class (reads the four BC arrays once; st->bc_class itself is ignored). */
int fnx_bc_classify(const FnxGrid* g, const FnxState* st, unsigned char* bc_class, void* stream);

int fnx_simulate_step(const FnxGrid* g, const FnxStepParams* prm, const FnxState* st,
                      void* ws, size_t ws_bytes, void* stream);

/* The two fused stages of fnx_simulate_step, exposed for drivers that interleave their own work (halo exchange):
 *   pre : simulate.py:96-133 + :144  U_adv, rho_adv (advection outputs) -> st->U, st->density, div
 *         (setConstVals, addBuoyancy, addGravity [prm->gravity_scale > 0], setWallBcs + periodic patches [method 0 only],
 *         setConstVals [not with st->flags_stick in method 1: the caller runs setWallBcsStick first], velocityDivergence
 *         [div != NULL])
 *   post: simulate.py:154-168        velocityUpdate(st->p), setWallBcs, setConstVals, in place on st->U / st->density (no
 *         periodic patches: fnx_simulate_step wraps it with them) */
int fnx_pre_projection(const FnxGrid* g, const FnxStepParams* prm, const FnxState* st, const float* U_adv,
                       const float* rho_adv, float* div, void* stream);
int fnx_post_projection(const FnxGrid* g, const FnxState* st, void* stream);

/* Adjoints of the linear stencil operators the training graph differentiates through (lib/model.py:190-227 and
 * fluid_net_train.py:366: velocityUpdate -> setWallBcs -> velocityDivergence; the reference gets them from autograd over
 * its ATen chains).  grad_* are (B,C,D,H,W) like the quantities they belong to; outputs are overwritten.
 *   velocityDivergence: grad_U = J^T grad_div
 *   velocityUpdate:     grad_U, grad_p from grad_U_out (the gradient w.r.t. the updated velocity); grad_U must not alias it
 *   setWallBcs:         zeroes a flag-dependent set of entries, so its adjoint is fnx_set_wall_bcs applied to the gradient */
int fnx_velocity_divergence_backward(const FnxGrid* g, const float* grad_div, const float* flags, float* grad_U, void* stream);
int fnx_velocity_update_backward(const FnxGrid* g, const float* grad_U_out, const float* flags, float* grad_U, float* grad_p,
                                 void* stream);

/* ---- z-slab decomposition of the 3D Jacobi step over the GPUs of one node (the reference is single
 * device, plume.py:131-135).  Rank r owns D_global / nranks planes and keeps `halo` ghost planes towards each
 * z-neighbour; its arrays hold local planes [0, owned + ghosts) = global planes [z_offset, ...).  One fnx_slab_step is
 * lib/simulate.py:28-171 (method 'jacobi', pTol 0) for the rank's owned planes, bit for bit what the single-domain
 * fnx_simulate_step computes there; ghost planes travel through an FnxSlabComm (neighbour exchange only, no collective on
 * the data path).  The driver enqueues on `stream` and on one internal communication stream joined back with events
 * (the whole step is capturable in a HIP graph); nothing synchronises except the optional CFL check. ---- */
typedef struct FnxSlabSeg {           /* one contiguous block of ghost planes of one field and channel */
  const void* send_lo; void* recv_lo;  /* to / from rank - 1 (NULL on rank 0) */
  const void* send_hi; void* recv_hi;  /* to / from rank + 1 (NULL on the last rank) */
  size_t bytes;
} FnxSlabSeg;
typedef struct FnxSlabComm {
  void* ctx;
  /* Stream-ordered exchange of all segments with both neighbours (RCCL: one ncclGroupStart/Send/Recv/GroupEnd). */
  int (*exchange)(void* ctx, const FnxSlabSeg* segs, int nsegs, void* stream);
  /* In-place max / sum over all ranks of n DEVICE floats (CFL guard, pTol residual; control path only). */
  int (*allreduce_max)(void* ctx, float* x, int n, void* stream);
  int (*allreduce_sum)(void* ctx, float* x, int n, void* stream);
  void (*destroy)(void* ctx);
  /* Optional (may be NULL): called by fnx_slab_step on the rank whose step FAILED, so that its neighbours -- which may already
   * be waiting for it inside exchange() -- return FNX_ECOMM instead of hanging (RCCL: ncclCommAbort; loopback: a group flag). */
  void (*abort)(void* ctx);
  /* Optional pair (both NULL or both set): DIRECT sends.  direct_begin(ctx, bytes, nsegs, dst, select, seg_stride, start_clock, stream)
   * names where the producing kernel may store the NEXT exchange's outgoing planes itself: towards rank - 1 (d = 0) / rank + 1 (d = 1;
   * NULL at the ends of the chain) into dst[d][q], q = (*select[d] + 1) & 1 read on the device (FnxPlaneMirror.slot_select; select[d]
   * NULL: q = 0), segment i of `bytes` bytes at + i * *seg_stride bytes; FNX_EINVAL when that exchange does not fit (the driver then
   * sends it the ordinary way).  direct_exchange(ctx, segs, nsegs, stream), enqueued behind the producing kernel, is exchange() for
   * segments whose send sides are already in place: it publishes them and receives.  *start_clock (may come back NULL): a device word
   * the producing kernel is asked to store its start time in (FnxPlaneMirror.start_clock).  The peer-store communicator hands out its
   * neighbours' mailbox slots (two per side, told apart by its device-side chunk counter: nothing in a captured step depends on how
   * many exchanges came before); the link model buffers of its own, and times the transfer from that clock. */
  int (*direct_begin)(void* ctx, size_t bytes, int nsegs, void* dst[2][2], const unsigned* select[2], size_t* seg_stride,
                      void** start_clock, void* stream);
  int (*direct_exchange)(void* ctx, const FnxSlabSeg* segs, int nsegs, void* stream);
} FnxSlabComm;
/* Ordering contract of a communicator: fnx_slab_step issues exchange() on its internal communication stream (the sweep blocks of
 * FNX_SLAB_DEEP_BESIDE: on its internal edge stream -- a communicator must take the stream it is given) and the
 * control-path all-reduces (CFL guard, pTol residual) on the caller's stream; every rank issues the same calls in the same
 * order (the step is deterministic).  An all-reduce is only issued when no exchange is pending (the caller's stream has
 * been made to wait for the last one) and is followed by a host synchronisation before the next exchange is posted, so the
 * two streams never have work of one communicator in flight at the same time. */
/* RCCL communicator (librccl is loaded on first use; no link-time dependency).  unique_id: the 128 bytes of
 * fnx_slab_rccl_unique_id() from rank 0, distributed by the caller (MPI, torch.distributed, a file ...). */
int fnx_slab_rccl_unique_id(void* out128);
int fnx_slab_comm_rccl(FnxSlabComm* out, int rank, int nranks, const void* unique_id128);
/* In-process communicator for `nranks` slabs driven by `nranks` host threads of one process (one device, or several
 * with peer access: each side of a pair records only its own event on its own stream): device-to-device copies ordered by
 * events.  A rank whose peer does not arrive within the group's timeout (120 s unless set) returns FNX_ECOMM from that call
 * and the group stays usable for DIRECT users of the communicator (a slow peer is not a dead one; an all-reduce round in which a
 * rank timed out is abandoned as a whole -- every rank of it fails, a retry starts a fresh round).  fnx_slab_step cannot resume a
 * step half-way: it calls abort() on ANY failure, a timeout included, so through the driver a timeout poisons the group like any
 * other error.  A rank whose group was aborted (FnxSlabComm.abort: some rank's step failed) returns FNX_ECOMM until
 * fnx_slab_loopback_group_reset, which the caller may issue once no rank is inside a call of the group.  Create the group
 * once, then one comm per rank. */
int fnx_slab_loopback_group(void** group, int nranks);
int fnx_slab_loopback_group_set_timeout(void* group, double seconds);
int fnx_slab_loopback_group_reset(void* group);
int fnx_slab_comm_loopback(FnxSlabComm* out, void* group, int rank);
void fnx_slab_loopback_group_free(void* group);
/* Link-model communicator (a rehearsal aid, not a transport): lets ONE process run the step of a middle rank (rank r of n with
 * 0 < r < n - 1) on one GPU.  An exchange is one launch that fills the ghost planes from the slab's own edge planes and occupies its
 * stream for latency_us + bytes per direction / gbytes_per_s (0 GB/s: no transfer time beyond the copy), so launch sequence, message
 * sizes and stream ordering are those of a real run and the time a schedule leaves exposed for an assumed link can be measured; the
 * field values are those of a periodic stack of this slab.  All-reduces return the rank's own value.  latency_us is the transport's:
 * ~20 us for a grouped RCCL send/recv, ~9 us for the peer-store launch (tools/peer_probe.py). */
int fnx_slab_comm_link_model(FnxSlabComm* out, double latency_us, double gbytes_per_s);
/* Peer-store communicator (no RCCL on the data path; the reference is single device, plume.py:131-135: no counterpart).  Every rank
 * owns a REGION of uncached device memory -- flag words and a mailbox of 2 x 2 slots of `mailbox_bytes` -- that its two z-neighbours
 * map (hipIpcOpenMemHandle between processes; ranks driven by threads of one process use the address as it is).  An exchange is ONE
 * short launch on the stream it is given: its first workgroups store this rank's planes straight into the neighbours' mailboxes over
 * xGMI and raise a flag there, its last workgroups wait -- on the device -- for the neighbours' flags and move the planes that arrived
 * into place; exchanges larger than a slot are cut into chunks (two may be in flight per direction).  No proxy thread, no host
 * synchronisation, 64 workgroups of the device while it runs.  The all-reduces of the control path (CFL guard, pTol residual, the CNN
 * projection's sums) travel along the chain of ranks through the same regions, driven by the host, added in RANK ORDER: the same bits on
 * every rank.  A wait that outlasts the timeout (30 s unless set) or sees an abort gives up and the communicator's next call
 * returns FNX_ECOMM.
 *   1. every rank: fnx_slab_peer_create -> its handle (FNX_PEER_HANDLE_BYTES bytes), to be carried to both neighbours by the caller
 *      (MPI, torch.distributed, a file ... like the RCCL unique id)
 *   2. every rank: fnx_slab_comm_peer with the handles of rank - 1 and rank + 1 (NULL at the ends of the chain)
 * The peer object must outlive the communicator (fnx_slab_comm_free, then fnx_slab_peer_free); all ranks use the same mailbox_bytes. */
#define FNX_PEER_HANDLE_BYTES 128
int fnx_slab_peer_create(void** peer, int rank, int nranks, size_t mailbox_bytes, void* handle_out);
int fnx_slab_peer_set_timeout(void* peer, double seconds);
/* FNX_ECOMM once a device-side wait of this rank's exchanges has timed out or the group was aborted (ABI 19).  The exchanges are
 * enqueued, not waited for, so a step or a replayed graph that met a dead neighbour still returns FNX_OK: ask after synchronising the
 * stream, before trusting the step's ghost planes.  (Every later communicator call fails with FNX_ECOMM by itself.) */
int fnx_slab_peer_failed(void* peer);
int fnx_slab_comm_peer(FnxSlabComm* out, void* peer, const void* handle_lo, const void* handle_hi);
void fnx_slab_peer_free(void* peer);
void fnx_slab_comm_free(FnxSlabComm* comm);

#define FNX_SLAB_MAX_HALO 64
/* How a block of sweeps_per_exchange Jacobi sweeps is ordered around its ghost exchange (all three give the same bits):
 *   DEEP_FIRST  every pass is cut a few planes inside each internal face; the deep parts of all passes run first -- they read
 *               no ghost plane, so the previous block's exchange (first block: the exchange of div) is still in flight --,
 *               then the edge parts (short launches), whose last one produces the planes the neighbours need next
 *   EDGE_FIRST  the edge parts of all passes first (shrinking plane ranges), their exchange posted, the interior parts behind it
 *   LAST_PASS   whole passes; only the last pass of a block is split into edge and interior
 *   DEEP_BESIDE the parts of DEEP_FIRST, the edge chain of a block on a second stream BESIDE its deep chain (edge part k waits
 *               for deep part k-1 only).  The exchanges of the sweep blocks are enqueued on that stream itself -- exchange(ctx, ...,
 *               stream) is called with the edge stream, behind the block's edge chain and ahead of the next block's -- so the chain
 *               exchange -> edge chain -> exchange crosses no stream; a block takes max(deep chain, exchange + edge chain) instead
 *               of their sum.  Pays with links of >= 150 GB/s per direction; a tie with DEEP_FIRST at 75 GB/s
 * Slabs thinner than 4 sweep blocks, solves of at most one block and pTol > 0 always run LAST_PASS. */
enum { FNX_SLAB_DEEP_FIRST = 0, FNX_SLAB_EDGE_FIRST = 1, FNX_SLAB_LAST_PASS = 2, FNX_SLAB_DEEP_BESIDE = 3 };
typedef struct FnxSlabConfig {
  int B, H, W, D_global;    /* the whole domain */
  int rank, nranks;         /* D_global % nranks == 0 */
  int halo;                 /* ghost planes per internal face (5 .. FNX_SLAB_MAX_HALO; the pressure solve uses all of them) */
  int sweeps_per_exchange;  /* Jacobi sweeps per ghost exchange of p (temporal blocking in z), clipped to halo */
  int static_flags;         /* 1: flags and BC arrays never change between steps (solver mask and BC class map are kept) */
  int cfl_check_every;      /* every that many steps the step begins with max |U| dt over all ranks (one host sync);
                               > 1 cell returns FNX_ECFL on every rank.  0 = never */
  int schedule;             /* FNX_SLAB_DEEP_FIRST (0, the default) / FNX_SLAB_EDGE_FIRST / FNX_SLAB_LAST_PASS / FNX_SLAB_DEEP_BESIDE */
  int method;               /* 0: Jacobi projection; 1: the driver is also sized for the CNN projection (prm->method 1 in
                               fnx_slab_step): needs halo >= FNX_SLAB_NET_MARGIN + 1, halo % 4 == 0, D_global / nranks % 4 == 0
                               when nranks > 1, and a workspace that holds the net's activations for owned + 2 x 48 planes (sized for the
                               untrimmed window; the towers run on nested crops of it, see FNX_SLAB_NET_MARGIN_FULL) */
  int direct_sends;         /* where the communicator offers direct sends (FnxSlabComm.direct_begin), the last edge part of a sweep block
                               can store the planes the neighbours need next straight into their windows (fnx_jacobi_pass_mirror) and the
                               exchange is then posted as direct_exchange.  0 (default): in DEEP_BESIDE, whose cycle exchange -> edge chain
                               the transport's launch and that part's run time then leave (middle rank modelled at 75 GB/s: 74 -> 78 % of a
                               ghost-free slab); not in DEEP_FIRST, which hides the exchange behind the deep chain anyway and only pays the
                               mirrored stores.  1: never.  2: in both.  Same bits in every case */
} FnxSlabConfig;
/* ghost planes of the MultiScaleNet's input a rank evaluates beyond its owned planes: the net's receptive field (< 48 cells at
 * full resolution), a multiple of 4 so that the rank's quarter- and half-resolution grids coincide with the global ones */
#define FNX_SLAB_NET_MARGIN 48
/* ... of which the towers need less: the full-resolution tower's receptive radius is 8 planes (5^3, four 3^3, 5^3), the
 * half-resolution tower's 14 (+ 2 for the resampling above it, + the 8), the quarter-resolution tower's 16 (+ 4, + the 24 = 44).  A
 * rank therefore runs each tower only on owned +- its own margin (nested crops, fnx_multiscale_forward_crop): 1.3 x the FLOPs of
 * its owned planes at 64 planes per rank instead of 2.5 x.  The pressure is then exact on the owned planes; the one plane below them
 * that velocityUpdate reads comes from the neighbour (a one-plane exchange of p). */
#define FNX_SLAB_NET_MARGIN_FULL 8
#define FNX_SLAB_NET_MARGIN_HALF 24
typedef struct FnxSlab FnxSlab;
/* Local geometry of a rank (what to allocate): planes it owns, ghosts below / above, global plane of local plane 0. */
int fnx_slab_layout(const FnxSlabConfig* cfg, int* owned, int* ghost_lo, int* ghost_hi, int* z_offset);
size_t fnx_slab_workspace_bytes(const FnxSlabConfig* cfg);
/* comm is borrowed (must outlive the slab).  nranks == 1 needs no communicator (comm may be NULL). */
int fnx_slab_create(FnxSlab** out, const FnxSlabConfig* cfg, const FnxSlabComm* comm);
void fnx_slab_destroy(FnxSlab* s);
/* One time step.  st: the rank's local arrays (with ghost planes), st->density required.  prm->method 0: the Jacobi projection
 * (st->net unused).  prm->method 1 (cfg.method 1, st->net = the packed weights): the CNN projection, lib/model.py:118-227 on
 * z-slabs -- _ScaleNet's std over the whole domain from per-rank fp64 (sum, sumsq) gathered and added in RANK ORDER (the same
 * bits on every rank; one host synchronisation), FNX_SLAB_NET_MARGIN + 1 ghost planes of U exchanged once, the net evaluated on
 * owned +- FNX_SLAB_NET_MARGIN planes, velocityUpdate / un-normalise / setWallBcs / setConstVals on the owned planes: p and U
 * within the CNN tolerance (1e-5 |ref|max) of the single-domain step, density bit for bit.
 * Jacobi:  prm->p_tol > 0 runs the reference's convergence test (fluids_init.cpp:961-979): one sweep per ghost exchange,
 * the squared differences over the owned planes all-reduced over the ranks, one host sync per sweep (as in fnx_jacobi);
 * every rank's part reproducible (fnx_residual).  The bit-for-bit statement above holds for p_tol == 0; with p_tol > 0 every
 * sweep still has the single-domain bits, but the residual is summed per rank and then over the ranks (fp32 all-reduce), so a
 * tolerance within rounding (~1e-7 relative) of a sweep's residual can stop one sweep earlier or later than fnx_jacobi does.
 * prm->static_flags is ignored (FnxSlabConfig.static_flags).  ws: fnx_slab_workspace_bytes, kept between steps.
 * On failure (other than FNX_ECFL, which every rank reports together) the communicator's abort() is called -- also when the
 * failure is a peer timeout of the loopback communicator -- and the caller's stream is made to wait for whatever the failed step
 * had already forked onto the driver's internal streams. */
int fnx_slab_step(FnxSlab* s, const FnxStepParams* prm, const FnxState* st, void* ws, size_t ws_bytes, void* stream);

/* Communication statistics of a rank (off by default: the event pairs they need cannot be recorded inside a graph capture).
 * enable(1) clears them; read() synchronises the recorded events, adds them up and clears the event list. */
typedef struct FnxSlabStats {
  double bytes_per_neighbour;  /* bytes posted towards EACH neighbour (and received from it) since enable() */
  double wait_ms;              /* time the compute stream stood in front of posted exchanges (HIP events around each wait) */
  long long exchanges;         /* ghost exchanges posted */
} FnxSlabStats;
int fnx_slab_stats_enable(FnxSlab* s, int on);
int fnx_slab_stats_read(FnxSlab* s, FnxSlabStats* out);
/* One-off probe of a communicator: `reps` exchanges of `bytes` bytes with each neighbour, back to back on `stream`, after
 * one untimed exchange; *ms_per_exchange = average duration (HIP events; synchronises).  scratch: 4 * bytes of device
 * memory.  Every rank of the communicator must call it with the same arguments. */
int fnx_slab_comm_probe(const FnxSlabComm* comm, void* scratch, size_t bytes, int reps, float* ms_per_exchange, void* stream);

/* MultiScaleNet / FluidNet.forward, lib/multi_scale_net.py:118-127 and lib/model.py:76-227 (ScaleNet variant).
 * weights_blob: 17 convs in the order convN_4[0..3], convN_2[0..5], convN_1[0..5], final; for each conv the
 * torch-layout weight (Cout,Cin,[kd,]kh,kw) then bias (Cout); fp32, DEVICE pointer.
 * fnx_scalenet_pack repacks it for the MFMA kernels into `packed` (size fnx_scalenet_packed_bytes). */
size_t fnx_scalenet_weight_floats(int is3D);
size_t fnx_scalenet_packed_bytes(int is3D);
int fnx_scalenet_pack(int is3D, const float* weights_blob, void* packed, void* stream);
/* precision_mode of the CNN entry points (and FnxStepParams.precision_mode).  The first two are exact-fp32 arithmetic on
 * v_mfma_f32_*_f32; the last two are opt-in and carry their own labels in every report (bench.py never puts them in the headline):
 *   FNX_PRECISION_FP32         the default: 3x3(x3) layers of launches that fill the chip run in the Winograd domain -- F(2x2,3x3)
 *                              (2.25x fewer multiplies) and, since round 6, F(4x4,3x3) in (y, x) (4x fewer) for the 64- and
 *                              128-output-channel layers; not the summation order of a direct convolution; within 1e-5 |ref|max of the
 *                              torch reference (tests/test_parity_gpu.py)
 *   FNX_PRECISION_FP32_DIRECT  every convolution as a direct sum over its taps (implicit GEMM), no Winograd
 *   FNX_PRECISION_BF16X6       FNX_PRECISION_FP32 with the Winograd-domain GEMMs of the 64- and 128-output-channel 3x3(x3) layers on
 *                              the bf16 matrix cores: every fp32 operand cut EXACTLY into three bf16 pieces (8 + 8 + 8 significand
 *                              bits), a product evaluated as six bf16 x bf16 MFMAs with fp32 accumulation (the three dropped
 *                              cross terms are below 2^-23 of the product: the size of one fp32 rounding).  Same tolerance as the
 *                              other modes in the tests (1e-5 |ref|max against oracle and goldens); every other layer as in
 *                              FNX_PRECISION_FP32
 *   FNX_PRECISION_BF16X3       FNX_PRECISION_BF16X6 with only the three products that involve no low piece (ah*bh + ah*bm + am*bh): half
 *                              the bf16 MFMAs and two thirds of the operand traffic; what is dropped is below 2^-15 of a product.  Its own
 *                              label and its own tolerance: 1e-4 |ref|max against oracle and goldens in the tests (measured ~1e-5:
 *                              profiles/r05).  SURVEY.md section 7's "accurate bf16" mode; the reference's own convolutions run on
 *                              torch.nn.Conv2d, whose CUDA default admits TF32 (lib/multi_scale_net.py:21-127)
 *   FNX_PRECISION_FP32_F4      (round 6) the 64- and 128-output-channel 3x3(x3) layers in the Winograd F(4x4,3x3) domain (conv3_wino4_kernel: 36
 *                              multiplies per 16 outputs instead of F(2x2)'s 64; 3D: in (y, x), the z taps as stages): exact-fp32 MFMAs,
 *                              the transforms round more (multipliers 4, 5, 8, 1/6, 1/24) -- measured 2x F(2x2)'s error, 0.07 of the
 *                              tests' 1e-5 |ref|max.  What FNX_PRECISION_FP32 runs since round 6 (256^3 CNN step 92.2 -> 81.0 ms,
 *                              1024^2 2.29 -> 2.14 ms)
 *   FNX_PRECISION_FP32_F2      F(2x2,3x3) for every Winograd layer: the default of rounds 2-5 (kept for A/B timing) */
enum { FNX_PRECISION_FP32 = 0, FNX_PRECISION_FP32_DIRECT = 1, FNX_PRECISION_BF16X6 = 2, FNX_PRECISION_BF16X3 = 3, FNX_PRECISION_FP32_F4 = 4,
       FNX_PRECISION_FP32_F2 = 5 };
/* x: (B,2,D,H,W) [div/s, occupancy] -> p (B,1,D,H,W) */
int fnx_multiscale_forward(const FnxGrid* g, const void* packed, const float* x, float* p, int precision_mode,
                           void* ws, size_t ws_bytes, void* stream);
/* The same pass on NESTED z-crops (what the z-slab driver's CNN projection runs on a rank's window, fnx_slab_step): x covers D
 * planes; the quarter-resolution tower runs on all of them, the half-resolution tower on planes [trim[2], D - trim[3]) and the
 * full-resolution tower on [trim[0], D - trim[1]) -- full-resolution plane counts, multiples of 4 (D too), trim[2] <= trim[0],
 * trim[3] <= trim[1].  p: (B,1,D - trim[0] - trim[1],H,W), the planes [trim[0], D - trim[1]).  Where a window ends at an artificial
 * face the tower's output is only meaningful beyond its receptive radius from that face (8 / 14 / 16 full-resolution planes for the
 * full / half / quarter tower, + 2 / 4 for the resampling between them): the caller sizes the windows (FNX_SLAB_NET_MARGIN and
 * FNX_SLAB_NET_MARGIN_FULL / _HALF above).  trim = {0,0,0,0} is fnx_multiscale_forward.  Workspace: as fnx_multiscale_forward for D planes. */
int fnx_multiscale_forward_crop(const FnxGrid* g, const void* packed, const float* x, float* p, int precision_mode,
                                const int trim[4], void* ws, size_t ws_bytes, void* stream);
/* input: (B,5|6,D,H,W) = [p, U, flags, density] -> p_out (B,1,..), U_out (B,2|3,..) */
int fnx_fluidnet_forward(const FnxGrid* g, const void* packed, const float* input, float normalize_threshold,
                         float* p_out, float* U_out, int precision_mode, void* ws, size_t ws_bytes, void* stream);

/* Optional timing of the dominant kernels with HIP events on the launch stream (used by bench.py for the roofline
 * figures).  While enabled, every launch of the tagged kernel class is bracketed by an event pair (up to 16384 pairs,
 * later launches are not recorded).  fnx_profile_read synchronises the recorded events and returns the summed
 * kernel time and the number of launches of that class; fnx_profile_enable(1) also clears earlier records.
 * fnx_profile_enable(2) (round 6): ONE pair around each RUN of consecutive launches of a class on a stream (the 50 passes of a
 * Jacobi-100 solve, the consecutive Winograd layers of a tower) -- the launches then run back to back as they do in a step; a pair
 * per launch holds each kernel until the one before it has drained and reads 3-10 % long (rocprofv3's per-kernel durations of a
 * replayed step agree with the run figure: profiles/r06/c_wino4_versions.txt, tools/debug/trace_step_span.sh). */
enum { FNX_PROF_JACOBI = 0, FNX_PROF_CONV_MFMA = 1, FNX_PROF_ADVECT = 2, FNX_PROF_STAGE = 3, FNX_PROF_CONV_DIRECT = 4,
       FNX_PROF_CONV_MFMA16 = 5, FNX_PROF_CONV_BF16 = 6 /* FNX_PRECISION_BF16X6 launches; work = bf16 MFMA FLOPs issued */, FNX_PROF_NTAGS = 7 };
int fnx_profile_enable(int on);
/* roctx ranges ("fnx:jacobi", "fnx:advect", "fnx:stage", "fnx:conv_*") around the enqueue of the same kernel classes, for
 * `rocprofv3 --marker-trace`.  The marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is loaded by this call,
 * not linked; off by default.  The reference has no tracing hooks. */
int fnx_roctx_enable(int on);
int fnx_profile_read(int tag, double* total_ms, int* launches);
/* What the recorded launches of a class ISSUED: for the conv classes the multiply-add FLOPs actually sent to the matrix
 * cores (a Winograd F(2x2,3x3) launch issues 16/36 of the direct convolution's), so that issued / time / peak is the
 * MFMA utilisation rather than a direct-convolution-equivalent rate.  0 for the other classes. */
int fnx_profile_read_work(int tag, double* work);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDNET_HIP_H */
