// C-ABI of libfluidnet_hip.so (include/fluidnet_hip.h): argument checks, workspace carving and the launch
// sequences.  No allocation, no synchronisation (except fnx_jacobi with p_tol > 0), no CPU fallback.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fluidnet_hip.h"
#include "fnx_cnn.h"
#include "fnx_kernels.h"
#include <atomic>
#include <dlfcn.h>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_OK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(FNX_EHIP, "HIP error: %s (%s)", hipGetErrorString(e_), #expr); \
  } while (0)

int check_grid(const FnxGrid* g) {
  if (!g) return fail(FNX_EINVAL, "grid descriptor is NULL");
  if (g->B < 1 || g->D < 1 || g->H < 3 || g->W < 3) return fail(FNX_EINVAL, "Dimension mismatch: B=%d D=%d H=%d W=%d", g->B, g->D, g->H, g->W);
  if (!g->is3D && g->D != 1) return fail(FNX_EINVAL, "2D velocity field but zdepth > 1");
  if (g->is3D && g->D < 3) return fail(FNX_EINVAL, "3D domain needs D >= 3");
  if ((long long)g->D * g->H * g->W >= (1ll << 31)) return fail(FNX_EINVAL, "more than 2^31 cells per sample");
  if ((long long)g->B * g->D > 65535) return fail(FNX_EINVAL, "B*D > 65535 not supported");
  if (!g->is3D && (g->W > 65535 || g->H > 65535)) return fail(FNX_EINVAL, "2D grids wider or taller than 65535 are not supported");
  if (g->k_end != 0 || g->k_begin != 0) {
    if (!g->is3D || g->k_begin < 0 || g->k_end > g->D || g->k_end <= g->k_begin)
      return fail(FNX_EINVAL, "bad compute window [%d, %d) for D=%d", g->k_begin, g->k_end, g->D);
  }
  if (g->D_global != 0 && (!g->is3D || g->z_offset < 0 || g->z_offset + g->D > g->D_global))
    return fail(FNX_EINVAL, "bad z-slab: z_offset=%d D=%d D_global=%d", g->z_offset, g->D, g->D_global);
  return FNX_OK;
}

inline GridDims dims(const FnxGrid* g) {
  GridDims d = make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global);
  if (g->is3D && g->k_end > g->k_begin) { d.K0 = g->k_begin; d.KN = g->k_end - g->k_begin; }
  return d;
}
// the same grid with the compute window widened by `by` planes (clipped to the array)
inline GridDims widened(GridDims d, int by) {
  const int a = d.K0 - by < 0 ? 0 : d.K0 - by, b = d.K0 + d.KN + by > d.D ? d.D : d.K0 + d.KN + by;
  d.K0 = a; d.KN = b - a;
  return d;
}
inline bool quirks(const FnxGrid* g) { return g->is3D && g->ref_quirks; }
inline size_t ncell(const FnxGrid* g) { return (size_t)g->B * g->D * g->H * g->W; }
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
  char* base; size_t off, cap;
  Carver(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
  void* take(size_t bytes) { void* r = base ? base + off : nullptr; off += al(bytes); return r; }
  bool ok() const { return base != nullptr && off <= cap; }
};

size_t ws_advect_scalar_fields(const FnxGrid* g) { return al(ncell(g) * 4) + al(ncell(g) * 4) + (g->is3D ? al(ncell(g) * 8) : 0); }   // fwd, traced cell, 3D clamp bounds
size_t ws_advect_vel_fields(const FnxGrid* g) { return al(ncell(g) * 4 * (g->is3D ? 3 : 2)); }
// fix-up bitmaps of the tile advection kernels (fnx_advect_march.h, fnx_advect_tile2d.h): 4 x one 64-bit word per 64-cell row segment
size_t ws_advect_fix(const FnxGrid* g) { return al(4 * 8 * (size_t)g->B * g->D * g->H * ((g->W + 63) / 64)); }
// the stand-alone operators take the tile kernels too (ABI 19): their fields + the bitmaps
size_t ws_advect_scalar(const FnxGrid* g) { return ws_advect_scalar_fields(g) + ws_advect_fix(g); }
size_t ws_advect_vel(const FnxGrid* g) { return ws_advect_vel_fields(g) + ws_advect_fix(g); }
size_t ws_mask(const FnxGrid* g) { return g->is3D ? al(fnx::jacobi3d_mask_bytes(dims(g))) : 0; }   // 3D solver: neighbour-mask bytes, twice (rows / row groups)
// Jacobi workspace: ping-pong pressure, the residual's fixed-order partial sums, one result float, the 3D neighbour mask
size_t ws_jacobi(const FnxGrid* g) { return al(ncell(g) * 4) + al(fnx::residual_scratch_bytes(g->B)) + al(4) + ws_mask(g); }
struct JacobiWs { float* tmp; double* partials; float* res; unsigned char* mask; };
bool carve_jacobi(const FnxGrid* g, void* ws, size_t ws_bytes, JacobiWs* out, size_t* need) {
  Carver c(ws, ws_bytes);
  out->tmp = (float*)c.take(ncell(g) * 4);
  out->partials = (double*)c.take(fnx::residual_scratch_bytes(g->B));
  out->res = (float*)c.take(4);
  out->mask = g->is3D ? (unsigned char*)c.take(ws_mask(g)) : nullptr;
  *need = c.off;
  return c.ok();
}
size_t ws_step(const FnxGrid* g) {
  const size_t nc = g->is3D ? 3 : 2;
  // 2D: the fused advection launches keep both forward fields at once
  size_t adv = ws_advect_scalar_fields(g) + ws_advect_vel_fields(g) + ws_advect_fix(g);
  size_t solve = ws_jacobi(g);
  size_t cnn = fnx::fluidnet_ws_bytes(dims(g), g->is3D);
  size_t tail = adv > solve ? adv : solve;
  if (cnn > tail) tail = cnn;
  return al(ncell(g) * 4) /*rho2*/ + al(ncell(g) * 4 * nc) /*U2*/ + al(ncell(g) * 4) /*div*/ +
         ws_mask(g) /*Jacobi obstacle mask, kept between steps*/ +
         al(ncell(g)) /*BC class map, kept between steps*/ +
         (g->is3D ? 0 : al(ncell(g) * 4 * nc)) /*2D: the viscous velocity that is advected (FnxStepParams.viscosity)*/ + tail;
}

}  // namespace

namespace fnx {
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace fnx

// ---- event-pair profiler ----------------------------------------------------------------------------
namespace fnx {
namespace {
constexpr int PROF_MAX = 16384;
struct Prof {
  bool on = false;
  bool runs = false;               // fnx_profile_enable(2): ONE event pair around a run of consecutive launches of a class on a stream
  int n = 0;
  hipEvent_t ev[PROF_MAX][2];
  int tag[PROF_MAX];
  int count[PROF_MAX];             // launches between the pair
  hipStream_t stream[PROF_MAX];
  int created = 0;
  int open_idx[FNX_PROF_NTAGS];
  double work[FNX_PROF_NTAGS];     // what the recorded launches of a class issued (MFMA FLOPs for the conv classes)
} g_prof;
}  // namespace

// roctx ranges around the same phases (the reference has no tracing hooks; rocprofv3 --marker-trace shows
// them).  The marker library is resolved on request (fnx_roctx_enable), never linked: without it the ranges are no-ops.
namespace {
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; std::atomic<bool> on{false}; } g_roctx;
const char* const kProfNames[FNX_PROF_NTAGS] = {"fnx:jacobi", "fnx:conv_mfma", "fnx:advect", "fnx:stage", "fnx:conv_direct", "fnx:conv_mfma16", "fnx:conv_bf16"};
}  // namespace

// A scope pops the range it pushed and no other: fnx_roctx_enable may be toggled (from another thread, or inside an open scope
// such as the pTol loop of a solve) between a scope's begin and end.
bool prof_begin(int tag, hipStream_t s) {
  const bool pushed = g_roctx.on.load(std::memory_order_acquire);
  if (pushed) g_roctx.push(kProfNames[tag]);
  if (!g_prof.on || g_prof.n >= PROF_MAX) { if (g_prof.on) g_prof.open_idx[tag] = -1; return pushed; }
  if (g_prof.runs && g_prof.n > 0 && g_prof.tag[g_prof.n - 1] == tag && g_prof.stream[g_prof.n - 1] == s) {
    // the launch before this one on the stream was of the same class: extend its pair (prof_end records the end event again) -- events between
    // back-to-back launches keep the next kernel from starting behind the one before it, and each then reads 3-5 % long
    g_prof.open_idx[tag] = g_prof.n - 1;
    ++g_prof.count[g_prof.n - 1];
    return pushed;
  }
  const int i = g_prof.n++;
  if (i >= g_prof.created) {
    hipEventCreate(&g_prof.ev[i][0]); hipEventCreate(&g_prof.ev[i][1]);
    g_prof.created = i + 1;
  }
  g_prof.tag[i] = tag;
  g_prof.count[i] = 1;
  g_prof.stream[i] = s;
  g_prof.open_idx[tag] = i;
  hipEventRecord(g_prof.ev[i][0], s);
  return pushed;
}

void prof_add_work(int tag, double amount) {
  if (g_prof.on && g_prof.open_idx[tag] >= 0) g_prof.work[tag] += amount;
}

void prof_end(int tag, hipStream_t s, bool pushed) {
  if (pushed) g_roctx.pop();
  if (!g_prof.on) return;
  const int i = g_prof.open_idx[tag];
  if (i >= 0) hipEventRecord(g_prof.ev[i][1], s);
}
}  // namespace fnx

extern "C" {

int fnx_profile_enable(int on) {
  fnx::g_prof.on = on != 0;
  fnx::g_prof.runs = on == 2;
  if (on) { fnx::g_prof.n = 0; for (int t = 0; t < FNX_PROF_NTAGS; ++t) { fnx::g_prof.open_idx[t] = -1; fnx::g_prof.work[t] = 0.0; } }
  return FNX_OK;
}

int fnx_roctx_enable(int on) {
  if (!on) { fnx::g_roctx.on.store(false, std::memory_order_release); return FNX_OK; }
  if (!fnx::g_roctx.push) {
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return fail(FNX_EINVAL, "roctx_enable: no roctx library found (librocprofiler-sdk-roctx.so / libroctx64.so)");
    fnx::g_roctx.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    fnx::g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!fnx::g_roctx.push || !fnx::g_roctx.pop) { fnx::g_roctx.push = nullptr; return fail(FNX_EINVAL, "roctx_enable: roctxRangePushA / roctxRangePop not found"); }
  }
  fnx::g_roctx.on.store(true, std::memory_order_release);
  return FNX_OK;
}

int fnx_profile_read_work(int tag, double* work) {
  if (tag < 0 || tag >= FNX_PROF_NTAGS || !work) return fail(FNX_EINVAL, "profile_read_work: bad tag");
  *work = fnx::g_prof.work[tag];
  return FNX_OK;
}

int fnx_profile_read(int tag, double* total_ms, int* launches) {
  double tot = 0.0; int cnt = 0;
  for (int i = 0; i < fnx::g_prof.n; ++i) {
    if (fnx::g_prof.tag[i] != tag) continue;
    if (hipEventSynchronize(fnx::g_prof.ev[i][1]) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, fnx::g_prof.ev[i][0], fnx::g_prof.ev[i][1]) == hipSuccess) { tot += ms; cnt += fnx::g_prof.count[i]; }
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return FNX_OK;
}

const char* fnx_last_error(void) { return g_err; }
int fnx_abi_version(void) { return FNX_ABI_VERSION; }

const char* fnx_device_name(void) {
  static thread_local char name[256];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { fail(FNX_EHIP, "no HIP device"); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { fail(FNX_EHIP, "no HIP device"); return nullptr; }
  snprintf(name, sizeof(name), "%s (%s)", prop.name, prop.gcnArchName);
  return name;
}

size_t fnx_workspace_bytes(const FnxGrid* g, int op) {
  if (check_grid(g) != FNX_OK) return 0;
  switch (op) {
    case FNX_OP_ADVECT_SCALAR: return ws_advect_scalar(g);
    case FNX_OP_ADVECT_VEL: return ws_advect_vel(g);
    case FNX_OP_JACOBI: return ws_jacobi(g);
    case FNX_OP_STEP: return ws_step(g);
    case FNX_OP_FLUIDNET: return fnx::fluidnet_ws_bytes(dims(g), g->is3D);
    case FNX_OP_ADVECT_STEP: return ws_advect_scalar_fields(g) + ws_advect_vel_fields(g) + ws_advect_fix(g);
  }
  fail(FNX_EINVAL, "unknown op %d", op);
  return 0;
}

int fnx_advect_scalar(const FnxGrid* g, float dt, const float* src, const float* U, const float* flags, float* dst,
                      int method, int bnd, int sample_outside, float strength, void* ws, size_t ws_bytes,
                      void* stream) {
  return fnx_advect_scalar_plan(g, dt, src, U, flags, dst, method, bnd, sample_outside, strength, FNX_ADVECT_PLAN_AUTO, ws, ws_bytes, stream);
}

int fnx_advect_scalar_plan(const FnxGrid* g, float dt, const float* src, const float* U, const float* flags, float* dst,
                           int method, int bnd, int sample_outside, float strength, int plan, void* ws, size_t ws_bytes,
                           void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (plan < FNX_ADVECT_PLAN_AUTO || plan > FNX_ADVECT_PLAN_TILES_SPLIT) return fail(FNX_EINVAL, "advect_scalar: unknown plan %d", plan);
  if (!src || !U || !flags || !dst) return fail(FNX_EINVAL, "advect_scalar: NULL tensor");
  if (dst == src) return fail(FNX_EINVAL, "advect_scalar: dst must not alias src");
  if (method != FNX_ADVECT_EULER && method != FNX_ADVECT_MACCORMACK) return fail(FNX_EMETHOD, "Advection method not supported");
  if (bnd != 1) return fail(FNX_EINVAL, "advect_scalar: only boundary_width == 1 is supported (the reference's MAC sampling strips exactly one border cell)");
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  if (method == FNX_ADVECT_EULER) {
    fnx::launch_sl_scalar(d, g->is3D, quirks(g), sample_outside != 0, dt, src, U, flags, dst, nullptr, s);
  } else {
    Carver c(ws, ws_bytes);
    float* fwd = (float*)c.take(ncell(g) * 4);
    int* cell = (int*)c.take(ncell(g) * 4);
    float* box = g->is3D ? (float*)c.take(ncell(g) * 8) : nullptr;
    unsigned long long* fix = (unsigned long long*)c.take(ws_advect_fix(g));
    if (!c.ok()) return fail(FNX_EWORKSPACE, "advect_scalar: workspace too small (%zu < %zu)", ws_bytes, c.off);
    const GridDims dw = widened(d, 2);                  // what the backward pass / clamp read at |U dt| <= 1
    // the LDS tile kernels of the fused pair, density part only (3D default semantics; 2D from 1.5 M cells): same bits as the
    // per-cell launches below (fnx_advect_march.h), ~1.5x faster in 3D
    const int tp = fnx::advect_tile_plan(d, dw, g->is3D, quirks(g), plan);
    if (g->is3D ? tp == 5 : tp == 2) {
      fnx::ProfScope ps(FNX_PROF_ADVECT, s);
      fnx::launch_advect_fused(d, dw, g->is3D, false, sample_outside != 0, dt, strength * 0.5f, src, U, flags, fwd, cell, nullptr, box,
                               dst, nullptr, fix, s, plan, 1);
      HIP_OK(hipGetLastError());
      return FNX_OK;
    }
    { fnx::ProfScope ps(FNX_PROF_ADVECT, s); fnx::launch_sl_scalar(dw, g->is3D, quirks(g), sample_outside != 0, dt, src, U, flags, fwd, cell, s); }
    if (g->is3D) { fnx::ProfScope ps(FNX_PROF_ADVECT, s); fnx::launch_box_minmax(dw, sample_outside != 0, src, flags, box, s); }
    else box = nullptr;                                  // 2D: the 3x3 clamp box is walked in the backward kernel
    fnx::ProfScope ps2(FNX_PROF_ADVECT, s);
    fnx::launch_sl_scalar_bwd_clamp(d, g->is3D, quirks(g), sample_outside != 0, dt, strength * 0.5f, src, fwd, cell, U,
                                    flags, box, dst, s);
  }
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_advect_vel(const FnxGrid* g, float dt, const float* orig, const float* U, const float* flags, float* dst,
                   int method, int bnd, float strength, void* ws, size_t ws_bytes, void* stream) {
  return fnx_advect_vel_plan(g, dt, orig, U, flags, dst, method, bnd, strength, FNX_ADVECT_PLAN_AUTO, ws, ws_bytes, stream);
}

int fnx_advect_vel_plan(const FnxGrid* g, float dt, const float* orig, const float* U, const float* flags, float* dst,
                        int method, int bnd, float strength, int plan, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (plan < FNX_ADVECT_PLAN_AUTO || plan > FNX_ADVECT_PLAN_TILES_SPLIT) return fail(FNX_EINVAL, "advect_vel: unknown plan %d", plan);
  if (!orig || !U || !flags || !dst) return fail(FNX_EINVAL, "advect_vel: NULL tensor");
  if (dst == orig || dst == U) return fail(FNX_EINVAL, "advect_vel: dst must not alias orig or U");
  if (method != FNX_ADVECT_EULER && method != FNX_ADVECT_MACCORMACK) return fail(FNX_EMETHOD, "Advection method not supported");
  if (bnd != 1) return fail(FNX_EINVAL, "advect_vel: only boundary_width == 1 is supported");
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  if (method == FNX_ADVECT_EULER) {
    fnx::launch_sl_mac(d, g->is3D, quirks(g), dt, orig, U, flags, dst, s);
  } else {
    Carver c(ws, ws_bytes);
    float* fwd = (float*)c.take(ncell(g) * 4 * (g->is3D ? 3 : 2));
    unsigned long long* fix = (unsigned long long*)c.take(ws_advect_fix(g));
    if (!c.ok()) return fail(FNX_EWORKSPACE, "advect_vel: workspace too small (%zu < %zu)", ws_bytes, c.off);
    // self-advection (orig is U: every call of simulate.py:93 without viscosity): the LDS tile kernels, velocity part only
    const int tp = orig == U ? fnx::advect_tile_plan(d, widened(d, 2), g->is3D, quirks(g), plan) : 0;
    if (g->is3D ? tp == 5 : tp == 2) {
      fnx::ProfScope ps(FNX_PROF_ADVECT, s);
      fnx::launch_advect_fused(d, widened(d, 2), g->is3D, false, false, dt, strength * 0.5f, nullptr, U, flags, nullptr, nullptr, fwd,
                               nullptr, nullptr, dst, fix, s, plan, 2);
      HIP_OK(hipGetLastError());
      return FNX_OK;
    }
    { fnx::ProfScope ps(FNX_PROF_ADVECT, s); fnx::launch_sl_mac(widened(d, 2), g->is3D, quirks(g), dt, orig, U, flags, fwd, s); }
    fnx::ProfScope ps2(FNX_PROF_ADVECT, s);
    fnx::launch_sl_mac_bwd_clamp(d, g->is3D, quirks(g), dt, strength * 0.5f, orig, fwd, U, flags, dst, s);
  }
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_advect_step(const FnxGrid* g, float dt, const float* density, const float* U, const float* flags,
                    float* density_dst, float* U_dst, int sample_outside, float strength, void* ws, size_t ws_bytes,
                    void* stream) {
  return fnx_advect_step_plan(g, dt, density, U, flags, density_dst, U_dst, sample_outside, strength, FNX_ADVECT_PLAN_AUTO, ws, ws_bytes, stream);
}

int fnx_advect_step_plan(const FnxGrid* g, float dt, const float* density, const float* U, const float* flags,
                         float* density_dst, float* U_dst, int sample_outside, float strength, int plan, void* ws,
                         size_t ws_bytes, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (plan < FNX_ADVECT_PLAN_AUTO || plan > FNX_ADVECT_PLAN_TILES_SPLIT) return fail(FNX_EINVAL, "advect_step: unknown plan %d", plan);
  if (!density || !U || !flags || !density_dst || !U_dst) return fail(FNX_EINVAL, "advect_step: NULL tensor");
  if (density_dst == density || U_dst == U) return fail(FNX_EINVAL, "advect_step: dst must not alias the inputs");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = ncell(g), nc = g->is3D ? 3 : 2;
  Carver c(ws, ws_bytes);
  float* rho_fwd = (float*)c.take(n * 4);
  int* cell = (int*)c.take(n * 4);
  float* box = g->is3D ? (float*)c.take(n * 8) : nullptr;
  float* U_fwd = (float*)c.take(n * 4 * nc);
  unsigned long long* fix = (unsigned long long*)c.take(ws_advect_fix(g));
  if (!c.ok()) return fail(FNX_EWORKSPACE, "advect_step: workspace too small (%zu < %zu)", ws_bytes, c.off);
  const GridDims d = dims(g);
  fnx::ProfScope ps(FNX_PROF_ADVECT, s);
  fnx::launch_advect_fused(d, widened(d, 2), g->is3D, quirks(g), sample_outside != 0, dt, strength * 0.5f, density, U, flags,
                           rho_fwd, cell, U_fwd, box, density_dst, U_dst, fix, s, plan);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_velocity_divergence(const FnxGrid* g, const float* U, const float* flags, float* div, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U || !flags || !div) return fail(FNX_EINVAL, "velocity_divergence: NULL tensor");
  fnx::launch_divergence(dims(g), g->is3D, U, flags, div, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

static int jacobi_solve(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual, float p_tol,
                        int max_iter, int* iters_done, void* ws, size_t ws_bytes, unsigned char* kept_mask, bool reuse_mask,
                        void* stream, int verbose);

int fnx_jacobi(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual, float p_tol,
               int max_iter, int* iters_done, void* ws, size_t ws_bytes, void* stream) {
  return jacobi_solve(g, flags, div, p, residual, p_tol, max_iter, iters_done, ws, ws_bytes, nullptr, false, stream, 0);
}

int fnx_jacobi_verbose(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual, float p_tol,
                       int max_iter, int* iters_done, void* ws, size_t ws_bytes, void* stream) {
  return jacobi_solve(g, flags, div, p, residual, p_tol, max_iter, iters_done, ws, ws_bytes, nullptr, false, stream, 1);
}

static int jacobi_solve(const FnxGrid* g, const float* flags, const float* div, float* p, float* residual, float p_tol,
                        int max_iter, int* iters_done, void* ws, size_t ws_bytes, unsigned char* kept_mask, bool reuse_mask,
                        void* stream, int verbose) {
  if (int rc = check_grid(g)) return rc;
  if (!flags || !div || !p) return fail(FNX_EINVAL, "solve_linear_system: NULL tensor");
  if (max_iter < 1) return fail(FNX_EINVAL, "At least 1 iteration is needed (maxIter < 1)");
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  JacobiWs W; size_t need;
  if (!carve_jacobi(g, ws, ws_bytes, &W, &need)) return fail(FNX_EWORKSPACE, "solve_linear_system: workspace too small (%zu < %zu)", ws_bytes, need);
  float* tmp = W.tmp;
  unsigned char* mask = W.mask;
  if (g->is3D && kept_mask) mask = kept_mask;           // a slot nothing else in the step scribbles on
  else reuse_mask = false;
  const bool q = quirks(g);
  if (g->is3D && !reuse_mask) fnx::launch_jacobi3d_mask(d, q, flags, mask, s);
  auto sweep = [&](const float* in, float* out, int k, bool from_zero, int lay = 0) {
    fnx::ProfScope ps(FNX_PROF_JACOBI, s);
    if (g->is3D) {
      if (k == 2) fnx::launch_jacobi3d_x2(d, mask, div, in, out, s, 0, 0, from_zero, -1, lay);
      else fnx::launch_jacobi3d(d, mask, div, in, out, from_zero, s);
    } else {
      fnx::launch_jacobi(d, flags, div, in, out, k, from_zero, s);
    }
  };
  // ||a - b||_2 per sample, max over the batch, reproducible (fixed-order fp64 partial sums, no atomics)
  const size_t per = (size_t)g->D * g->H * g->W;
  auto residual_of = [&](const float* a, const float* b, float* out) { fnx::launch_residual(g->B, per, 0, per, a, b, W.partials, nullptr, out, s); };
  const bool per_sweep = p_tol > 0.f || verbose;        // the reference's per-sweep host test / per-sweep print
  if (!per_sweep) {
    // Sweeps per launch: 2D up to kmax (register temporal blocking); 3D pairs (the first one knows p = 0).  When the caller
    // wants the residual ||p_n - p_(n-1)||, the last sweep runs on its own so that both iterates are in memory.
    const int fused = residual ? max_iter - 1 : max_iter;
    int plan[1024]; int nl = 0, left = fused;
    const int kmax = g->is3D ? 2 : fnx::jacobi_max_sweeps_per_launch(d, false, fused);
    while (left > 0 && nl < 1022) {
      const int k = left < kmax ? left : kmax;
      plan[nl++] = k; left -= k;
    }
    if (left > 0) return fail(FNX_EINVAL, "solve_linear_system: max_iter too large for one call (%d)", max_iter);
    const int ntot = nl + (residual ? 1 : 0);
    const float* in = nullptr;
    // 3D: consecutive two-sweep passes hand each other p in the row-quad layout (fewer, wider vector-memory
    // instructions: launch_jacobi3d_x2); the last of them writes rows
    const bool quad = g->is3D && fnx::jacobi3d_quad_ok(d);
    bool in_quad = false;
    for (int l = 0; l < nl; ++l) {
      float* out = ((ntot - 1 - l) % 2 == 0) ? p : tmp;    // the last launch writes p
      const bool out_quad = quad && plan[l] == 2 && l + 1 < nl && plan[l + 1] == 2;
      sweep(in, out, plan[l], l == 0, (in_quad ? 1 : 0) | (out_quad ? 2 : 0));
      in = out; in_quad = out_quad;
    }
    if (residual) {
      sweep(in, p, 1, nl == 0);
      residual_of(p, in, residual);
    }
    if (iters_done) *iters_done = max_iter;
  } else {
    // the reference's own per-sweep convergence test (fluids_init.cpp:961-979): one host read per sweep
    const float* in = nullptr;
    float* bufs[2] = { p, tmp };
    int sweeps = 0;
    float r = 0.f;
    for (;;) {
      float* out = bufs[sweeps & 1];
      sweep(in, out, 1, sweeps == 0);
      residual_of(out, in, W.res);
      HIP_OK(hipMemcpyAsync(&r, W.res, 4, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      in = out;
      ++sweeps;
      if (verbose) printf("Jacobi iteration %d: residual %g\n", sweeps, (double)r);      // fluids_init.cpp:968-971
      if (r < p_tol) {
        if (verbose) printf("Jacobi max residual fell below p_tol (%g) (terminating)\n", (double)p_tol);          // :973-978
        break;
      }
      if (sweeps >= max_iter) {
        if (verbose) printf("Jacobi max iteration count (%d) reached (terminating)\n", max_iter);                 // :982-987
        break;
      }
    }
    if (verbose) fflush(stdout);
    if (in != p) HIP_OK(hipMemcpyAsync(p, in, ncell(g) * 4, hipMemcpyDeviceToDevice, s));
    if (residual) HIP_OK(hipMemcpyAsync(residual, W.res, 4, hipMemcpyDeviceToDevice, s));
    if (iters_done) *iters_done = sweeps;
  }
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_jacobi_sweeps(const FnxGrid* g, const float* flags, const float* div, float* p, int nsweeps, void* ws,
                      size_t ws_bytes, void* stream) {
  return fnx_jacobi_sweeps_ex(g, flags, div, p, nsweeps, ws, ws_bytes, 0, stream);
}

int fnx_jacobi_sweeps_ex(const FnxGrid* g, const float* flags, const float* div, float* p, int nsweeps, void* ws,
                         size_t ws_bytes, int reuse_mask, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!flags || !div || !p) return fail(FNX_EINVAL, "jacobi_sweeps: NULL tensor");
  if (nsweeps < 1) return fail(FNX_EINVAL, "At least 1 iteration is needed (maxIter < 1)");
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  JacobiWs W; size_t need;
  if (!carve_jacobi(g, ws, ws_bytes, &W, &need)) return fail(FNX_EWORKSPACE, "jacobi_sweeps: workspace too small (%zu < %zu)", ws_bytes, need);
  float* tmp = W.tmp;
  unsigned char* mask = W.mask;
  const bool from_zero = (reuse_mask & 2) != 0;
  if (g->is3D && !(reuse_mask & 1)) fnx::launch_jacobi3d_mask(d, quirks(g), flags, mask, s);
  const int kmax = g->is3D ? 2 : fnx::jacobi_max_sweeps_per_launch(d, false, nsweeps);
  // ping-pong p -> tmp -> p ...; an odd number of launches ends in tmp and is copied back
  const float* in = from_zero ? nullptr : p;
  int done = 0;
  const bool quad = g->is3D && fnx::jacobi3d_quad_ok(d);      // see jacobi_solve
  bool in_quad = false;
  for (int l = 0; done < nsweeps; ++l) {
    const int k = nsweeps - done < kmax ? nsweeps - done : kmax;
    float* out = (l % 2 == 0) ? tmp : p;
    const bool out_quad = quad && k == 2 && nsweeps - done - k >= 2;        // the launch after this one is a two-sweep pass too
    const int lay = (in_quad ? 1 : 0) | (out_quad ? 2 : 0);
    in_quad = out_quad;
    { fnx::ProfScope ps(FNX_PROF_JACOBI, s);
      const bool fz = from_zero && l == 0;
      if (g->is3D) {
        if (k == 2) fnx::launch_jacobi3d_x2(d, mask, div, in, out, s, 0, 0, fz, -1, lay);
        else fnx::launch_jacobi3d(d, mask, div, in, out, fz, s);
      } else {
        fnx::launch_jacobi(d, flags, div, in, out, k, fz, s);
      } }
    in = out; done += k;
  }
  if (in != p) HIP_OK(hipMemcpyAsync(p, in, ncell(g) * 4, hipMemcpyDeviceToDevice, s));
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_jacobi_pass2(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                     int nsweeps, int k_begin, int k_end, int k_begin2, void* ws, size_t ws_bytes, int reuse_mask,
                     void* stream) {
  return fnx_jacobi_pass_layout(g, flags, div, p_in, p_out, nsweeps, k_begin, k_end, k_begin2, 0, ws, ws_bytes, reuse_mask, stream);
}

int fnx_jacobi_quad_ok(const FnxGrid* g) {
  if (check_grid(g) != FNX_OK || !g->is3D) return 0;
  return fnx::jacobi3d_quad_ok(dims(g)) ? 1 : 0;
}

int fnx_jacobi_pass_layout(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                           int nsweeps, int k_begin, int k_end, int k_begin2, int layout, void* ws, size_t ws_bytes,
                           int reuse_mask, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (layout < 0 || layout > 3) return fail(FNX_EINVAL, "jacobi_pass: layout must be 0..3");
  if (layout != 0 && (nsweeps != 2 || !g->is3D || !fnx::jacobi3d_quad_ok(dims(g))))
    return fail(FNX_EINVAL, "jacobi_pass: the row-quad layout needs a two-sweep pass on a grid fnx_jacobi_quad_ok accepts");
  if (!flags || !div || !p_out || p_in == p_out) return fail(FNX_EINVAL, "jacobi_pass: NULL or aliased tensor");
  if (!g->is3D) return fail(FNX_EINVAL, "jacobi_pass: 3D only (2D uses fnx_jacobi_sweeps)");
  const bool from_zero = p_in == nullptr;                 // the first pass of a solve: p = 0 everywhere, nothing to read
  if (nsweeps < 1 || nsweeps > 2) return fail(FNX_EINVAL, "jacobi_pass: nsweeps must be 1 or 2");
  if (k_begin < 0 || k_end > g->D || (k_end != 0 && k_end <= k_begin)) return fail(FNX_EINVAL, "jacobi_pass: bad plane range");
  if (k_begin2 >= 0) {
    const int n = k_end - k_begin;
    if (k_end == 0 || k_begin2 + n > g->D || (k_begin2 < k_end && k_begin < k_begin2 + n))
      return fail(FNX_EINVAL, "jacobi_pass: bad or overlapping second plane range");
  }
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  JacobiWs W; size_t need;
  if (!carve_jacobi(g, ws, ws_bytes, &W, &need)) return fail(FNX_EWORKSPACE, "jacobi_pass: workspace too small (%zu < %zu)", ws_bytes, need);
  unsigned char* mask = W.mask;
  if (!reuse_mask) fnx::launch_jacobi3d_mask(d, quirks(g), flags, mask, s);
  fnx::ProfScope ps(FNX_PROF_JACOBI, s);
  if (nsweeps == 2) fnx::launch_jacobi3d_x2(d, mask, div, p_in, p_out, s, k_begin, k_end, from_zero, k_begin2, layout);
  else {
    fnx::launch_jacobi3d(d, mask, div, p_in, p_out, from_zero, s, k_begin, k_end);
    if (k_begin2 >= 0) fnx::launch_jacobi3d(d, mask, div, p_in, p_out, from_zero, s, k_begin2, k_begin2 + (k_end - k_begin));
  }
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_jacobi_pass_mirror_ok(const FnxGrid* g, int planes, int two_ranges, int layout) {
  if (check_grid(g) != FNX_OK || !g->is3D || planes < 1) return 0;
  return fnx::jacobi3d_mirror_ok(dims(g), planes, two_ranges != 0, false, layout) ? 1 : 0;
}

int fnx_jacobi_pass_mirror(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out, int k_begin,
                           int k_end, int k_begin2, int layout, const FnxPlaneMirror* mirror, void* ws, size_t ws_bytes,
                           int reuse_mask, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!g->is3D || !flags || !div || !p_in || !p_out || p_in == p_out || !mirror) return fail(FNX_EINVAL, "jacobi_pass_mirror: NULL or aliased argument");
  if (k_begin < 0 || k_end > g->D || k_end <= k_begin) return fail(FNX_EINVAL, "jacobi_pass_mirror: bad plane range");
  const int np = k_end - k_begin;
  if (k_begin2 >= 0 && (k_begin2 + np > g->D || (k_begin2 < k_end && k_begin < k_begin2 + np)))
    return fail(FNX_EINVAL, "jacobi_pass_mirror: bad or overlapping second plane range");
  if ((layout != 0 && layout != 3) || (layout == 3 && !fnx::jacobi3d_quad_ok(dims(g))) || !fnx::jacobi3d_mirror_ok(dims(g), np, k_begin2 >= 0, false, layout))
    return fail(FNX_EINVAL, "jacobi_pass_mirror: this launch cannot mirror its output (fnx_jacobi_pass_mirror_ok)");
  if (mirror->planes < 1 || !mirror->out[0][0] || (mirror->slot_select[0] && !mirror->out[0][1]) ||
      (k_begin2 >= 0 && (!mirror->out[1][0] || (mirror->slot_select[1] && !mirror->out[1][1]))))
    return fail(FNX_EINVAL, "jacobi_pass_mirror: bad mirror");
  hipStream_t s = (hipStream_t)stream;
  const GridDims d = dims(g);
  JacobiWs W; size_t need;
  if (!carve_jacobi(g, ws, ws_bytes, &W, &need)) return fail(FNX_EWORKSPACE, "jacobi_pass_mirror: workspace too small (%zu < %zu)", ws_bytes, need);
  if (!reuse_mask) fnx::launch_jacobi3d_mask(d, quirks(g), flags, W.mask, s);
  fnx::ProfScope ps(FNX_PROF_JACOBI, s);
  fnx::JacobiMirror m{};
  for (int r = 0; r < 2; ++r) { m.out[r][0] = mirror->out[r][0]; m.out[r][1] = mirror->out[r][1]; m.sel[r] = mirror->slot_select[r]; }
  m.k[0] = mirror->k_first[0]; m.k[1] = mirror->k_first[1]; m.n = mirror->planes;
  m.bstride = mirror->sample_stride; m.clock = mirror->start_clock;
  fnx::launch_jacobi3d_x2(d, W.mask, div, p_in, p_out, s, k_begin, k_end, false, k_begin2, layout, &m);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_jacobi_pass(const FnxGrid* g, const float* flags, const float* div, const float* p_in, float* p_out,
                    int nsweeps, int k_begin, int k_end, void* ws, size_t ws_bytes, int reuse_mask, void* stream) {
  return fnx_jacobi_pass2(g, flags, div, p_in, p_out, nsweeps, k_begin, k_end, -1, ws, ws_bytes, reuse_mask, stream);
}

int fnx_residual(const FnxGrid* g, const float* a, const float* b, float* sumsq, float* residual, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!a || (!sumsq && !residual)) return fail(FNX_EINVAL, "residual: NULL tensor");
  if (!ws || ws_bytes < fnx::residual_scratch_bytes(g->B)) return fail(FNX_EWORKSPACE, "residual: workspace too small (%zu < %zu)", ws_bytes, fnx::residual_scratch_bytes(g->B));
  const GridDims d = dims(g);
  const size_t per = (size_t)g->D * g->H * g->W;
  fnx::launch_residual(g->B, per, (size_t)d.K0 * d.HW, (size_t)d.KN * d.HW, a, b, (double*)ws, sumsq, residual, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_velocity_update(const FnxGrid* g, const float* p, float* U, const float* flags, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!p || !U || !flags) return fail(FNX_EINVAL, "velocity_update: NULL tensor");
  fnx::launch_velocity_update(dims(g), g->is3D, p, U, flags, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_add_gravity(const FnxGrid* g, float* U, const float* flags, const float gravity[3], float dt, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U || !flags || !gravity) return fail(FNX_EINVAL, "add_gravity: NULL tensor");
  fnx::launch_add_gravity(dims(g), g->is3D, U, flags, gravity[0] * dt, gravity[1] * dt, gravity[2] * dt,
                          (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_correct_scalar(const FnxGrid* g, float dt, float* src, const float* div, const float* flags, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!src || !div || !flags) return fail(FNX_EINVAL, "correct_scalar: NULL tensor");
  fnx::launch_correct_scalar(dims(g), g->is3D, dt * 0.5f, src, div, flags, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_add_viscosity(const FnxGrid* g, float dt, const float* U_in, float* U_out, const float* flags, float viscosity,
                      void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U_in || !U_out || !flags) return fail(FNX_EINVAL, "add_viscosity: NULL tensor");
  if (g->is3D) return fail(FNX_EINVAL, "add_viscosity: 2D only (reference viscosity.py:5)");
  if (U_in == U_out) return fail(FNX_EINVAL, "add_viscosity: U_out must not alias U_in");
  if (!(viscosity >= 0.f)) return fail(FNX_EINVAL, "add_viscosity: viscosity must be positive");
  const float coef = (float)((double)dt * (double)viscosity);
  fnx::launch_add_viscosity(dims(g), U_in, U_out, flags, coef, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_add_buoyancy(const FnxGrid* g, float* U, const float* flags, const float* density, const float gravity[3],
                     float rho_star, float dt, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U || !flags || !density || !gravity) return fail(FNX_EINVAL, "add_buoyancy: NULL tensor");
  const float sx = gravity[0] * dt, sy = gravity[1] * dt, sz = gravity[2] * dt;   // strength = gravity * dt
  fnx::launch_add_buoyancy(dims(g), g->is3D, quirks(g), U, flags, density, sx, sy, sz, rho_star, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_set_wall_bcs(const FnxGrid* g, float* U, const float* flags, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U || !flags) return fail(FNX_EINVAL, "set_wall_bcs: NULL tensor");
  fnx::launch_set_wall_bcs(dims(g), g->is3D, U, flags, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_set_wall_bcs_stick(const FnxGrid* g, const float* U_in, float* U_out, const float* flags, const float* flags_stick,
                           void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!U_in || !U_out || !flags || !flags_stick) return fail(FNX_EINVAL, "set_wall_bcs_stick: NULL tensor");
  if (g->is3D || g->D != 1) return fail(FNX_EINVAL, "set_wall_bcs_stick: 2D only (the reference's 3D branch cannot run, set_wall_bcs_stick.py:85-86)");
  if (U_in == U_out) return fail(FNX_EINVAL, "set_wall_bcs_stick: U_out must not alias U_in");
  fnx::launch_set_wall_bcs_stick(dims(g), U_in, U_out, flags, flags_stick, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_set_const_vals(const FnxGrid* g, float* U, const float* UBC, const float* UBCInvMask, float* density,
                       const float* densityBC, const float* densityBCInvMask, void* stream) {
  if (int rc = check_grid(g)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (U && UBC && UBCInvMask) fnx::launch_set_const_vals(ncell(g) * (g->is3D ? 3 : 2), U, UBC, UBCInvMask, s);
  if (density && densityBC && densityBCInvMask) fnx::launch_set_const_vals(ncell(g), density, densityBC, densityBCInvMask, s);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_flags_to_occupancy(const FnxGrid* g, const float* flags, float* occupancy, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!flags || !occupancy) return fail(FNX_EINVAL, "flags_to_occupancy: NULL tensor");
  fnx::launch_flags_to_occupancy(ncell(g), flags, occupancy, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_max_abs(const FnxGrid* g, const float* x, int channels, float* out_max, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!x || !out_max || channels < 1) return fail(FNX_EINVAL, "max_abs: NULL tensor or channels < 1");
  if (((size_t)x & 15) != 0) return fail(FNX_EINVAL, "max_abs: the field must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  HIP_OK(hipMemsetAsync(out_max, 0, 4, s));
  fnx::launch_max_abs(ncell(g) * (size_t)channels, x, out_max, s);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_empty_domain(const FnxGrid* g, float* flags, int boundary_width, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!flags) return fail(FNX_EINVAL, "empty_domain: NULL tensor");
  if (boundary_width < 1) return fail(FNX_EINVAL, "Boundary width must be greater than zero!");
  fnx::launch_empty_domain(dims(g), g->is3D, flags, boundary_width, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_create_cylinder(const FnxGrid* g, float* flags, double center_x, double center_y, double radius, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!flags) return fail(FNX_EINVAL, "create_cylinder: NULL tensor");
  // the reference squares the radius as a python double and the comparison casts it to fp32 (geometry_utils.py:31)
  const float r2 = (float)(radius * radius);
  fnx::launch_create_cylinder(dims(g), flags, (float)center_x, (float)center_y, r2, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_create_box2d(const FnxGrid* g, float* flags, float x0, float x1, float y0, float y1, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!flags) return fail(FNX_EINVAL, "create_box2d: NULL tensor");
  fnx::launch_create_box2d(dims(g), flags, x0, x1, y0, y1, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_get_centered(const FnxGrid* g, const float* U, float* centered, void* stream) {
  // a pure per-cell average: any block of a field is a valid input (the drivers pass interior blocks, plume.py:351), so
  // only the launch limits are checked, not the operators' minimum domain size
  if (!g) return fail(FNX_EINVAL, "grid descriptor is NULL");
  if (g->B < 1 || g->D < 1 || g->H < 1 || g->W < 1) return fail(FNX_EINVAL, "Dimension mismatch: B=%d D=%d H=%d W=%d", g->B, g->D, g->H, g->W);
  if (!g->is3D && g->D != 1) return fail(FNX_EINVAL, "2D velocity field but zdepth > 1");
  if ((long long)g->D * g->H * g->W >= (1ll << 31)) return fail(FNX_EINVAL, "more than 2^31 cells per sample");
  if ((long long)g->B * g->D > 65535 || (g->H + 3) / 4 > 65535) return fail(FNX_EINVAL, "B*D or H/4 > 65535 not supported");
  if (!U || !centered) return fail(FNX_EINVAL, "get_centered: NULL tensor");
  if ((const float*)centered == U) return fail(FNX_EINVAL, "get_centered: the output must not alias U");
  fnx::launch_get_centered(make_dims(g->B, g->D, g->H, g->W), g->is3D, U, centered, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_velocity_divergence_backward(const FnxGrid* g, const float* grad_div, const float* flags, float* grad_U, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!grad_div || !flags || !grad_U) return fail(FNX_EINVAL, "velocity_divergence_backward: NULL tensor");
  fnx::launch_divergence_bwd(make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global), g->is3D, grad_div, flags, grad_U, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_velocity_update_backward(const FnxGrid* g, const float* grad_U_out, const float* flags, float* grad_U, float* grad_p,
                                 void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!grad_U_out || !flags || !grad_U || !grad_p) return fail(FNX_EINVAL, "velocity_update_backward: NULL tensor");
  if (grad_U == grad_U_out) return fail(FNX_EINVAL, "velocity_update_backward: grad_U must not alias grad_U_out (a cell reads its +1 neighbours)");
  fnx::launch_velocity_update_bwd(make_dims(g->B, g->D, g->H, g->W, g->z_offset, g->D_global), g->is3D, grad_U_out, flags, grad_U, grad_p, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_pre_projection(const FnxGrid* g, const FnxStepParams* prm, const FnxState* st, const float* U_adv,
                       const float* rho_adv, float* div, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!prm || !st || !st->U || !st->flags || !U_adv) return fail(FNX_EINVAL, "pre_projection: NULL state");
  if (rho_adv && !st->density) return fail(FNX_EINVAL, "pre_projection: rho_adv given but state has no density");
  // a cell reads the advected fields of its +1 / -1 neighbours while other threads write theirs: no in-place use
  if (U_adv == st->U || (rho_adv && rho_adv == st->density)) return fail(FNX_EINVAL, "pre_projection: the advected fields must not alias the state");
  const bool has_rho = rho_adv != nullptr;
  const bool buoy = has_rho && prm->buoyancy_scale > 0.f;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (buoy) {
    const float ns = -prm->buoyancy_scale;                       // gravity.mul_(-buoyancyScale), simulate.py:101-105
    const float gx = prm->gravity_vec[0] * ns, gy = prm->gravity_vec[1] * ns, gz = prm->gravity_vec[2] * ns;
    sx = gx * prm->dt; sy = gy * prm->dt; sz = gz * prm->dt;     // strength = gravity * dt, source_terms.py:45
  }
  const bool ubc = st->UBC && st->UBCInvMask, rbc = st->densityBC && st->densityBCInvMask;
  // simulate.py:107-114: addGravity(gravityVec * -gravityScale) after the buoyancy, inside the density branch
  const bool grav = has_rho && prm->gravity_scale > 0.f;
  float gv[3] = {0.f, 0.f, 0.f};
  if (grav) {
    const float ns = -prm->gravity_scale;
    for (int a = 0; a < 3; ++a) gv[a] = (prm->gravity_vec[a] * ns) * prm->dt;       // force = gravity * dt, source_terms.py
  }
  // simulate.py:119-130: setWallBcs and the periodic patches in the Jacobi branch only; with 'flags_stick' the convnet branch
  // runs setWallBcsStick between the stages and the second setConstVals, which are then the caller's (fnx_simulate_step)
  const bool wall = prm->method == 0;
  const bool periodic = wall && (prm->periodic & 1);
  const bool second_bcs = !(prm->method == 1 && st->flags_stick);
  fnx::ProfScope ps(FNX_PROF_STAGE, (hipStream_t)stream);
  // One fused pass writes the staged fields and the divergence (each cell re-derives the staged component of its +1 neighbours),
  // unless the periodic patches have to go between the stages and the divergence: then staging, patches, a divergence pass.
  const bool split = periodic;
  // the staging covers one plane more than the divergences asked for where there is one (3D ranges): the divergence of the last
  // plane reads the advected fields of that plane, and the callers of plane ranges expect it staged as well
  GridDims ds = dims(g);
  const int div_k_end = ds.K0 + ds.KN;
  if (g->is3D && div && ds.K0 + ds.KN < ds.D) ds.KN += 1;
  fnx::launch_pre_projection(ds, g->is3D, quirks(g), U_adv, rho_adv, st->flags, ubc ? st->UBC : nullptr,
                             ubc ? st->UBCInvMask : nullptr, rbc ? st->densityBC : nullptr,
                             rbc ? st->densityBCInvMask : nullptr, st->U, st->density, (split && div) ? nullptr : div, buoy, sx, sy, sz,
                             prm->operating_density, wall, (hipStream_t)stream, st->bc_class, grav ? gv : nullptr, second_bcs, div_k_end);
  if (periodic)
    fnx::launch_periodic_pre(ds, g->is3D, U_adv, ubc ? st->UBC : nullptr, ubc ? st->UBCInvMask : nullptr, st->U,
                             (prm->periodic & 2) != 0, (prm->periodic & 4) != 0, (hipStream_t)stream);
  if (split && div) fnx::launch_divergence(dims(g), g->is3D, st->U, st->flags, div, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_post_projection(const FnxGrid* g, const FnxState* st, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!st || !st->U || !st->flags || !st->p) return fail(FNX_EINVAL, "post_projection: NULL state");
  const bool ubc = st->UBC && st->UBCInvMask, rbc = st->densityBC && st->densityBCInvMask;
  fnx::ProfScope ps(FNX_PROF_STAGE, (hipStream_t)stream);
  fnx::launch_post_projection(dims(g), g->is3D, st->p, st->U, st->density, st->flags, ubc ? st->UBC : nullptr,
                              ubc ? st->UBCInvMask : nullptr, rbc ? st->densityBC : nullptr,
                              rbc ? st->densityBCInvMask : nullptr, (hipStream_t)stream, st->bc_class, st->density_bc_applied != 0);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_bc_classify(const FnxGrid* g, const FnxState* st, unsigned char* bc_class, void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!st || !bc_class) return fail(FNX_EINVAL, "bc_classify: NULL argument");
  const bool ubc = st->UBC && st->UBCInvMask, rbc = st->densityBC && st->densityBCInvMask;
  fnx::launch_bc_classify(dims(g), g->is3D, ubc ? st->UBC : nullptr, ubc ? st->UBCInvMask : nullptr,
                          rbc ? st->densityBC : nullptr, rbc ? st->densityBCInvMask : nullptr, bc_class, (hipStream_t)stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

int fnx_simulate_step(const FnxGrid* g, const FnxStepParams* prm, const FnxState* st, void* ws, size_t ws_bytes,
                      void* stream) {
  if (int rc = check_grid(g)) return rc;
  if (!prm || !st || !st->p || !st->U || !st->flags) return fail(FNX_EINVAL, "simulate_step: NULL state");
  if (prm->method != 0 && prm->method != 1) return fail(FNX_EINVAL, "Simulation method not supported. Choose either convnet or jacobi.");
  if (prm->method == 1 && !st->net) return fail(FNX_EINVAL, "simulate_step: convnet method needs packed weights");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = ncell(g), nc = g->is3D ? 3 : 2;
  Carver c(ws, ws_bytes);
  float* rho2 = (float*)c.take(n * 4);
  float* U2 = (float*)c.take(n * 4 * nc);
  float* div = (float*)c.take(n * 4);
  unsigned char* kept_mask = g->is3D ? (unsigned char*)c.take(ws_mask(g)) : nullptr;
  unsigned char* kept_cls = (unsigned char*)c.take(n);
  float* orig = g->is3D ? nullptr : (float*)c.take(n * 4 * nc);        // 2D: the viscous velocity (prm->viscosity > 0)
  void* tail = c.take(0);
  const size_t tail_bytes = ws_bytes > c.off ? ws_bytes - c.off : 0;
  if (!c.ok() || ws_bytes < ws_step(g)) return fail(FNX_EWORKSPACE, "simulate_step: workspace too small (%zu < %zu)", ws_bytes, ws_step(g));
  const bool has_rho = st->density != nullptr;
  const bool viscous = prm->viscosity > 0.f;
  if (prm->viscosity < 0.f) return fail(FNX_EINVAL, "Viscosity must be positive");
  if (viscous && g->is3D) return fail(FNX_EINVAL, "simulate_step: viscosity is 2D only (reference viscosity.py:5)");
  const bool stick = prm->method == 1 && st->flags_stick != nullptr;           // simulate.py:129-130, :165-166
  if (stick && g->is3D) return fail(FNX_EINVAL, "simulate_step: flags_stick is 2D only (set_wall_bcs_stick.py:85-86)");
  const bool periodic = prm->method == 0 && (prm->periodic & 1);
  // simulate.py:66-93: advect density then velocity (both by the OLD U; the velocity advected is the viscous one)
  if (viscous) {
    if (int rc = fnx_add_viscosity(g, prm->dt, st->U, orig, st->flags, prm->viscosity, stream)) return rc;
    if (has_rho) {
      if (int rc = fnx_advect_scalar(g, prm->dt, st->density, st->U, st->flags, rho2, FNX_ADVECT_MACCORMACK, 1,
                                     prm->sample_outside_fluid, prm->maccormack_strength, tail, tail_bytes, stream)) return rc;
    }
    if (int rc = fnx_advect_vel(g, prm->dt, orig, st->U, st->flags, U2, FNX_ADVECT_MACCORMACK, 1, prm->maccormack_strength,
                                tail, tail_bytes, stream)) return rc;
  } else if (has_rho) {
    // forward passes of both advections in one launch, backward/clamp passes in another (same cell functions)
    if (int rc = fnx_advect_step(g, prm->dt, st->density, st->U, st->flags, rho2, U2, prm->sample_outside_fluid,
                                 prm->maccormack_strength, tail, tail_bytes, stream)) return rc;
  } else {                                     // simulate.py:71-83: no 'density' key in the batch
    if (int rc = fnx_advect_vel(g, prm->dt, st->U, st->U, st->flags, U2, FNX_ADVECT_MACCORMACK, 1,
                                prm->maccormack_strength, tail, tail_bytes, stream)) return rc;
  }
  if (has_rho && prm->correct_scalar) {        // simulate.py:79-81: by the divergence of the OLD U (st->U is still untouched)
    fnx::launch_divergence(dims(g), g->is3D, st->U, st->flags, div, s);
    if (int rc = fnx_correct_scalar(g, prm->dt, rho2, div, st->flags, stream)) return rc;
  }
  // static BC arrays (static_flags bit 1): the BC stages go by the class map kept in the workspace (built once, bit 2)
  FnxState stc = *st;
  stc.bc_class = nullptr;
  if ((prm->static_flags & 2) && (st->UBC || st->densityBC)) {
    if (!(prm->static_flags & 4)) { if (int rc = fnx_bc_classify(g, st, kept_cls, stream)) return rc; }
    stc.bc_class = kept_cls;
  }
  stc.density_bc_applied = 1;                  // by the pre-projection stage below, with these BC arrays
  st = &stc;
  // simulate.py:96-133 (+ :144 divergence) in one pass: BCs, buoyancy, gravity, wall BCs (+ periodic patches), BCs, -div
  if (int rc = fnx_pre_projection(g, prm, st, U2, has_rho ? rho2 : nullptr, prm->method == 0 ? div : nullptr, stream)) return rc;
  const GridDims d = dims(g);
  float* rho = has_rho ? st->density : nullptr;
  // setWallBcsStick is out of place: st->U -> U2 (free since the staging pass) and back
  auto stick_pass = [&]() -> int {
    if (int rc = fnx_set_wall_bcs_stick(g, st->U, U2, st->flags, st->flags_stick, stream)) return rc;
    HIP_OK(hipMemcpyAsync(st->U, U2, n * 4 * nc, hipMemcpyDeviceToDevice, s));
    return FNX_OK;
  };
  if (prm->method == 0) {
    // simulate.py:144-168
    if (int rc = jacobi_solve(g, st->flags, div, st->p, nullptr, prm->p_tol, prm->jacobi_iter, nullptr, tail, tail_bytes, kept_mask, (prm->static_flags & 1) != 0, stream, 0)) return rc;
    if (!periodic) return fnx_post_projection(g, st, stream);
    // simulate.py:157-164: the patches read the field as it was before setWallBcs; their source row / column (border cells,
    // which velocityUpdate leaves alone) is saved ahead of the in-place pass.  The solve is through with the tail.
    if (tail_bytes < fnx::periodic_save_bytes(d)) return fail(FNX_EWORKSPACE, "simulate_step: workspace too small for the periodic patches");
    const bool ubc = st->UBC && st->UBCInvMask;
    const bool px = (prm->periodic & 2) != 0, py = (prm->periodic & 4) != 0;
    fnx::launch_periodic_post(d, g->is3D, st->U, (float*)tail, nullptr, nullptr, px, py, 0, s);
    if (int rc = fnx_post_projection(g, st, stream)) return rc;
    fnx::launch_periodic_post(d, g->is3D, st->U, (float*)tail, ubc ? st->UBC : nullptr, ubc ? st->UBCInvMask : nullptr, px, py, 1, s);
    HIP_OK(hipGetLastError());
    return FNX_OK;
  } else {
    if (stick) {                                // simulate.py:129-133: setWallBcsStick, then the second setConstVals
      if (int rc = stick_pass()) return rc;
      if (int rc = fnx_set_const_vals(g, st->U, st->UBC, st->UBCInvMask, rho, st->densityBC, st->densityBCInvMask, stream)) return rc;
    }
    // simulate.py:136-142: p, U = net(cat(p, U, flags, density)) -- the net only reads U and flags (model.py:104-126),
    // so the concatenation is not materialised: U is projected in place.
    if (tail_bytes < fnx::fluidnet_ws_bytes(d, g->is3D)) return fail(FNX_EWORKSPACE, "simulate_step: workspace too small for the CNN");
    if (prm->precision_mode < FNX_PRECISION_FP32 || prm->precision_mode > FNX_PRECISION_FP32_F2)
      return fail(FNX_EINVAL, "simulate_step: unknown precision_mode %d", prm->precision_mode);
    // without flags_stick the tail of the net's forward and the step's last setConstVals are one pass (fluidnet_core)
    if (int rc = fnx::fluidnet_core(g, st->net, st->flags, prm->normalize_threshold, prm->precision_mode, st->p, st->U, tail, stream,
                                    stick ? nullptr : st)) return rc;
    if (!stick) { HIP_OK(hipGetLastError()); return FNX_OK; }
    if (int rc = stick_pass()) return rc;                                       // simulate.py:165-166
  }
  fnx_set_const_vals(g, st->U, st->UBC, st->UBCInvMask, rho, st->densityBC, st->densityBCInvMask, stream);
  HIP_OK(hipGetLastError());
  return FNX_OK;
}

}  // extern "C"
