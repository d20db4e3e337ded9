#include <memory>
// torch cpp-extension `fluidnet_cpp` for MI355X: the reference's three pybind entry points
// (pytorch/lib/fluid/cpp/fluids_init.cpp:1009-1014) with identical names and positional signatures, bound
// to the C ABI of libfluidnet_hip.so, plus the operators the reference implements in Python
// (lib/fluid/*.py) as additional entry points in the same style.
//
// This file is host plumbing only: shape/contiguity checks mirroring the reference's asserts, output and
// workspace allocation from torch's caching allocator, the current HIP stream, status -> RuntimeError.
// There is no CPU path: tensors must live on a HIP device.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>

#include "../../include/fluidnet_hip.h"

namespace {

using at::Tensor;

// Per-call geometry options (no reference counterpart; the reference is single-device and has one 3D behaviour).  The
// extension keeps NO mutable state: every 3D-capable entry point takes an optional `geom` and is re-entrant like the
// reference's (SURVEY.md 8b "no globals").
struct Geom {
  bool ref_quirks = false;          // 3D only: reproduce the reference's 3D defects bit-for-bit (FnxGrid.ref_quirks)
  int z_offset = 0, D_global = 0;   // z-slab view: the tensors hold planes [z_offset, z_offset + D) of a D_global-deep domain
  int k_begin = 0, k_end = 0;       // compute window: plane-parallel operators produce local planes [k_begin, k_end) only
};

void check_status(int rc) {
  TORCH_CHECK(rc == FNX_OK, fnx_last_error());
}

void check_field(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be on the GPU (libfluidnet_hip has no CPU path)");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
  TORCH_CHECK(t.dim() == 5, "Dimension mismatch");
  TORCH_CHECK(t.is_contiguous(), "Input is not contiguous");
}

FnxGrid grid_of(const Tensor& flags, bool is3D, const Geom* geom) {
  static const Geom none;
  const Geom& go = geom ? *geom : none;
  check_field(flags, "flags");
  TORCH_CHECK(flags.size(1) == 1, "flags is not scalar");
  FnxGrid g{};
  g.B = (int)flags.size(0); g.D = (int)flags.size(2); g.H = (int)flags.size(3); g.W = (int)flags.size(4);
  g.is3D = is3D ? 1 : 0;
  g.ref_quirks = go.ref_quirks ? 1 : 0;
  g.z_offset = is3D ? go.z_offset : 0; g.D_global = is3D ? go.D_global : 0;
  g.k_begin = is3D ? go.k_begin : 0; g.k_end = is3D ? go.k_end : 0;
  if (!is3D) TORCH_CHECK(g.D == 1, "2D velocity field but zdepth > 1");
  return g;
}

void check_vel(const Tensor& U, const FnxGrid& g, const char* name) {
  check_field(U, name);
  TORCH_CHECK(U.size(1) == (g.is3D ? 3 : 2), g.is3D ? "3D velocity field must have 3 channels" : "2D velocity field must have only 2 channels");
  TORCH_CHECK(U.size(0) == g.B && U.size(2) == g.D && U.size(3) == g.H && U.size(4) == g.W, "Size mismatch");
}

void check_scalar(const Tensor& s, const FnxGrid& g, const char* name) {
  check_field(s, name);
  TORCH_CHECK(s.size(1) == 1 && s.size(0) == g.B && s.size(2) == g.D && s.size(3) == g.H && s.size(4) == g.W, "Size mismatch");
}

struct Workspace {
  Tensor t; void* ptr; size_t bytes;
  Workspace(const FnxGrid& g, int op, const Tensor& like) {
    bytes = fnx_workspace_bytes(&g, op);
    t = at::empty({(int64_t)(bytes ? bytes : 1)}, like.options().dtype(at::kByte));
    ptr = t.data_ptr();
  }
};

// In-place entry points bump the written tensor's version counter like torch's own in-place operators do: autograd's
// saved-tensor checks and simulate()'s "have flags / the BC arrays changed since the last step?" test (_simulate.py) rely on it.
void wrote(const Tensor& t) {
  if (t.defined() && !t.is_inference()) t.unsafeGetTensorImpl()->bump_version();
}
void wrote(const c10::optional<Tensor>& t) { if (t.has_value()) wrote(*t); }

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

int method_of(const std::string& m) {
  // cpp/advect_type.cpp:5-16
  if (m == "eulerFluidNet") return FNX_ADVECT_EULER;
  if (m == "maccormackFluidNet") return FNX_ADVECT_MACCORMACK;
  TORCH_CHECK(false, "Advection method not supported: ", m);
  return -1;
}

// ---- the reference's three entry points -------------------------------------------------------------
// `out` (not in the reference): write into an existing tensor -- with a compute window (set_window) the z-slab driver
// fills one advected field from several calls.
int plan_of(const std::string& plan) {
  TORCH_CHECK(plan == "auto" || plan == "tiles" || plan == "cells" || plan == "tiles_split", "plan must be 'auto', 'tiles', 'cells' or 'tiles_split'");
  return plan == "tiles" ? FNX_ADVECT_PLAN_TILES : (plan == "cells" ? FNX_ADVECT_PLAN_CELLS : (plan == "tiles_split" ? FNX_ADVECT_PLAN_TILES_SPLIT : FNX_ADVECT_PLAN_AUTO));
}

Tensor advect_scalar(float dt, Tensor src, Tensor U, Tensor flags, const std::string method, int bnd,
                     const bool sample_outside_fluid, const float maccormack_strength, c10::optional<Tensor> out,
                     const Geom* geom, const std::string& plan) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(src, g, "src");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor dst = (out.has_value() && out->defined()) ? *out : at::empty_like(src);
  check_scalar(dst, g, "out");
  Workspace ws(g, FNX_OP_ADVECT_SCALAR, src);
  check_status(fnx_advect_scalar_plan(&g, dt, src.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(),
                                      dst.data_ptr<float>(), method_of(method), bnd, sample_outside_fluid,
                                      maccormack_strength, plan_of(plan), ws.ptr, ws.bytes, cur_stream(src)));
  return dst;
}

Tensor advect_vel(float dt, Tensor orig, Tensor U, Tensor flags, const std::string method, int bnd,
                  const float maccormack_strength, c10::optional<Tensor> out, const Geom* geom, const std::string& plan) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_vel(orig, g, "orig");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor dst = (out.has_value() && out->defined()) ? *out : at::empty_like(U);
  check_vel(dst, g, "out");
  Workspace ws(g, FNX_OP_ADVECT_VEL, U);
  check_status(fnx_advect_vel_plan(&g, dt, orig.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(),
                                   dst.data_ptr<float>(), method_of(method), bnd, maccormack_strength, plan_of(plan), ws.ptr,
                                   ws.bytes, cur_stream(U)));
  return dst;
}

// the two advections of one step fused (fnx_advect_step): returns {density_adv, U_adv}
std::vector<Tensor> advect_step(float dt, Tensor density, Tensor U, Tensor flags, const bool sample_outside_fluid,
                                const float maccormack_strength, c10::optional<Tensor> out_density,
                                c10::optional<Tensor> out_U, const Geom* geom, const std::string& plan) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(density, g, "density");
  const int pl = plan_of(plan);
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor rd = (out_density.has_value() && out_density->defined()) ? *out_density : at::empty_like(density);
  Tensor ud = (out_U.has_value() && out_U->defined()) ? *out_U : at::empty_like(U);
  check_scalar(rd, g, "out_density"); check_vel(ud, g, "out_U");
  Workspace ws(g, FNX_OP_ADVECT_STEP, U);
  check_status(fnx_advect_step_plan(&g, dt, density.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(),
                                    rd.data_ptr<float>(), ud.data_ptr<float>(), sample_outside_fluid, maccormack_strength, pl,
                                    ws.ptr, ws.bytes, cur_stream(U)));
  return {rd, ud};
}

std::vector<Tensor> solve_linear_system(Tensor flags, Tensor div, const bool is3D, const float p_tol,
                                        const int max_iter, const bool verbose, const Geom* geom) {
  FnxGrid g = grid_of(flags, is3D, geom);
  check_scalar(div, g, "div");
  if (!is3D) TORCH_CHECK(g.D == 1, "d > 1 for a 2D domain");
  TORCH_CHECK(max_iter >= 1, "At least 1 iteration is needed (maxIter < 1)");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor p = at::empty_like(flags);
  Tensor residual = at::zeros({}, flags.options());
  Workspace ws(g, FNX_OP_JACOBI, flags);
  int iters = 0;
  // verbose: the reference prints the residual after every sweep (fluids_init.cpp:968-971), which needs every iterate on the
  // host -- one sweep per launch and a host sync per sweep, like pTol > 0
  check_status((verbose ? fnx_jacobi_verbose : fnx_jacobi)(&g, flags.data_ptr<float>(), div.data_ptr<float>(), p.data_ptr<float>(),
                                                           residual.data_ptr<float>(), p_tol, max_iter, &iters, ws.ptr, ws.bytes,
                                                           cur_stream(flags)));
  return {p, residual};
}

// ---- operators the reference writes in Python -------------------------------------------------------
Tensor velocity_divergence(Tensor U, Tensor flags, const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor div = at::empty_like(flags);
  check_status(fnx_velocity_divergence(&g, U.data_ptr<float>(), flags.data_ptr<float>(), div.data_ptr<float>(), cur_stream(U)));
  return div;
}

void velocity_update_(Tensor pressure, Tensor U, Tensor flags, const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(pressure, g, "pressure");
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_velocity_update(&g, pressure.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(), cur_stream(U)));
  wrote(U);
}

void add_buoyancy_(Tensor U, Tensor flags, Tensor density, std::vector<double> gravity, double rho_star, double dt,
                   const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(density, g, "density");
  TORCH_CHECK(gravity.size() == 3, "Gravity must be a 3D vector (even in 2D)");
  const float gv[3] = {(float)gravity[0], (float)gravity[1], (float)gravity[2]};
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_add_buoyancy(&g, U.data_ptr<float>(), flags.data_ptr<float>(), density.data_ptr<float>(), gv,
                                (float)rho_star, (float)dt, cur_stream(U)));
  wrote(U);
}

void add_gravity_(Tensor U, Tensor flags, std::vector<double> gravity, double dt, const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U");
  TORCH_CHECK(gravity.size() == 3, "Gravity must be a 3D vector (even in 2D)");
  const float gv[3] = {(float)gravity[0], (float)gravity[1], (float)gravity[2]};
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_add_gravity(&g, U.data_ptr<float>(), flags.data_ptr<float>(), gv, (float)dt, cur_stream(U)));
  wrote(U);
}

// correctScalar (cpp/advection.py:9-12), in place on src
void correct_scalar_(double dt, Tensor src, Tensor div, Tensor flags) {
  check_field(src, "src"); check_field(flags, "flags");
  FnxGrid g = grid_of(flags, flags.size(2) > 1, nullptr);
  check_scalar(src, g, "src"); check_scalar(div, g, "div"); check_scalar(flags, g, "flags");
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_correct_scalar(&g, (float)dt, src.data_ptr<float>(), div.data_ptr<float>(), flags.data_ptr<float>(), cur_stream(src)));
  wrote(src);
}

void add_viscosity_(double dt, Tensor U, Tensor flags, double viscosity) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, nullptr);
  check_vel(U, g, "U");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor old = U.clone();
  check_status(fnx_add_viscosity(&g, (float)dt, old.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(),
                                 (float)viscosity, cur_stream(U)));
  wrote(U);
}

// set_wall_bcs_stick.py:5-157 (2D; in place on U like the reference)
void set_wall_bcs_stick_(Tensor U, Tensor flags, Tensor flags_stick) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, nullptr);
  check_vel(U, g, "U");
  check_scalar(flags_stick, g, "flags_stick");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor old = U.clone();
  check_status(fnx_set_wall_bcs_stick(&g, old.data_ptr<float>(), U.data_ptr<float>(), flags.data_ptr<float>(),
                                      flags_stick.data_ptr<float>(), cur_stream(U)));
  wrote(U);
}

void set_wall_bcs_(Tensor U, Tensor flags, const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U");
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_set_wall_bcs(&g, U.data_ptr<float>(), flags.data_ptr<float>(), cur_stream(U)));
  wrote(U);
}

void set_const_vals_(Tensor U, c10::optional<Tensor> UBC, c10::optional<Tensor> UBCInvMask, c10::optional<Tensor> density,
                     c10::optional<Tensor> densityBC, c10::optional<Tensor> densityBCInvMask) {
  check_field(U, "U");
  FnxGrid g{}; g.B = (int)U.size(0); g.D = (int)U.size(2); g.H = (int)U.size(3); g.W = (int)U.size(4);
  g.is3D = U.size(1) == 3; g.ref_quirks = 0; g.z_offset = 0; g.D_global = 0;
  auto ptr = [&](c10::optional<Tensor>& t, bool vel) -> float* {
    if (!t.has_value() || !t->defined()) return nullptr;
    if (vel) check_vel(*t, g, "UBC"); else check_scalar(*t, g, "densityBC");
    return t->data_ptr<float>();
  };
  c10::hip::HIPGuard guard(U.get_device());
  check_status(fnx_set_const_vals(&g, U.data_ptr<float>(), ptr(UBC, true), ptr(UBCInvMask, true), ptr(density, false),
                                  ptr(densityBC, false), ptr(densityBCInvMask, false), cur_stream(U)));
  wrote(U); wrote(density);
}

// max |x| of a 5-D field as a 0-dim device tensor (no host sync) -- CFL guard of the z-slab driver
Tensor max_abs(Tensor x) {
  check_field(x, "x");
  FnxGrid g{}; g.B = (int)x.size(0); g.D = (int)x.size(2); g.H = (int)x.size(3); g.W = (int)x.size(4); g.is3D = g.D > 1;
  c10::hip::HIPGuard guard(x.get_device());
  Tensor out = at::empty({}, x.options());
  check_status(fnx_max_abs(&g, x.data_ptr<float>(), (int)x.size(1), out.data_ptr<float>(), cur_stream(x)));
  return out;
}

Tensor flags_to_occupancy(Tensor flags) {
  FnxGrid g = grid_of(flags, flags.size(2) > 1, nullptr);
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor occ = at::empty_like(flags);
  check_status(fnx_flags_to_occupancy(&g, flags.data_ptr<float>(), occ.data_ptr<float>(), cur_stream(flags)));
  return occ;
}

void empty_domain_(Tensor flags, int boundary_width, const Geom* geom) {
  FnxGrid g = grid_of(flags, flags.size(2) > 1, geom);
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_empty_domain(&g, flags.data_ptr<float>(), boundary_width, cur_stream(flags)));
  wrote(flags);
}

// adjoints of the linear stencil operators (fnx_velocity_divergence_backward, fnx_velocity_update_backward)
Tensor velocity_divergence_backward(Tensor grad_div, Tensor flags, bool is3D, const Geom* geom) {
  FnxGrid g = grid_of(flags, is3D, geom);
  check_scalar(grad_div, g, "grad_div");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor gU = at::empty({flags.size(0), is3D ? 3 : 2, flags.size(2), flags.size(3), flags.size(4)}, flags.options());
  check_status(fnx_velocity_divergence_backward(&g, grad_div.data_ptr<float>(), flags.data_ptr<float>(), gU.data_ptr<float>(), cur_stream(flags)));
  return gU;
}
std::vector<Tensor> velocity_update_backward(Tensor grad_U_out, Tensor flags, const Geom* geom) {
  check_field(grad_U_out, "grad_U_out");
  FnxGrid g = grid_of(flags, grad_U_out.size(1) == 3, geom);
  check_vel(grad_U_out, g, "grad_U_out");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor gU = at::empty_like(grad_U_out), gp = at::empty_like(flags);
  check_status(fnx_velocity_update_backward(&g, grad_U_out.data_ptr<float>(), flags.data_ptr<float>(), gU.data_ptr<float>(),
                                            gp.data_ptr<float>(), cur_stream(flags)));
  return {gU, gp};
}

// geometry written into flags (all z planes), lib/fluid/geometry_utils.py:4-63
void create_cylinder_(Tensor flags, double center_x, double center_y, double radius) {
  FnxGrid g = grid_of(flags, flags.size(2) > 1, nullptr);
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_create_cylinder(&g, flags.data_ptr<float>(), center_x, center_y, radius, cur_stream(flags)));
  wrote(flags);
}

void create_box2d_(Tensor flags, double x0, double x1, double y0, double y1) {
  FnxGrid g = grid_of(flags, flags.size(2) > 1, nullptr);
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_create_box2d(&g, flags.data_ptr<float>(), (float)x0, (float)x1, (float)y0, (float)y1, cur_stream(flags)));
  wrote(flags);
}

// lib/fluid/grid.py:7-32
Tensor get_centered(Tensor U) {
  check_field(U, "U");
  TORCH_CHECK(U.size(1) == 2 || U.size(1) == 3, "velocity field must have 2 or 3 channels");
  FnxGrid g{}; g.B = (int)U.size(0); g.D = (int)U.size(2); g.H = (int)U.size(3); g.W = (int)U.size(4); g.is3D = U.size(1) == 3;
  TORCH_CHECK(g.is3D || g.D == 1, "2D velocity field but zdepth > 1");
  c10::hip::HIPGuard guard(U.get_device());
  Tensor out = at::empty({U.size(0), 3, U.size(2), U.size(3), U.size(4)}, U.options());
  check_status(fnx_get_centered(&g, U.data_ptr<float>(), out.data_ptr<float>(), cur_stream(U)));
  return out;
}

// ---- CNN pressure projection --------------------------------------------------------------------------
Tensor scalenet_pack(Tensor blob, bool is3D) {
  TORCH_CHECK(blob.is_cuda() && blob.scalar_type() == at::kFloat && blob.is_contiguous(), "weights blob must be a contiguous float32 GPU tensor");
  TORCH_CHECK((size_t)blob.numel() == fnx_scalenet_weight_floats(is3D), "weights blob has ", blob.numel(), " floats, expected ",
              fnx_scalenet_weight_floats(is3D));
  c10::hip::HIPGuard guard(blob.get_device());
  Tensor packed = at::zeros({(int64_t)fnx_scalenet_packed_bytes(is3D)}, blob.options().dtype(at::kByte));
  check_status(fnx_scalenet_pack(is3D, blob.data_ptr<float>(), packed.data_ptr(), cur_stream(blob)));
  return packed;
}

// precision_mode: "fp32" (default: Winograd where the launch fills the chip), "fp32_direct", "bf16x6" or "bf16x3" (FNX_PRECISION_*)
static int precision_of(const std::string& m) {
  TORCH_CHECK(m == "fp32" || m == "fp32_direct" || m == "bf16x6" || m == "bf16x3" || m == "fp32_f4" || m == "fp32_f2",
              "precision_mode must be 'fp32', 'fp32_direct', 'fp32_f4', 'fp32_f2', 'bf16x6' or 'bf16x3', got '", m, "'");
  if (m == "fp32_f4") return FNX_PRECISION_FP32_F4;
  if (m == "fp32_f2") return FNX_PRECISION_FP32_F2;
  if (m == "bf16x6") return FNX_PRECISION_BF16X6;
  if (m == "bf16x3") return FNX_PRECISION_BF16X3;
  return m == "fp32_direct" ? FNX_PRECISION_FP32_DIRECT : FNX_PRECISION_FP32;
}

Tensor multiscale_forward(Tensor packed, Tensor x, const std::string& precision_mode, std::vector<int64_t> trim) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous(), "x must be a contiguous float32 GPU tensor");
  TORCH_CHECK((x.dim() == 4 || x.dim() == 5) && x.size(1) == 2, "x must be (B,2,H,W) or (B,2,D,H,W)");
  TORCH_CHECK(trim.empty() || trim.size() == 4, "trim: {full-tower low, high, half-tower low, high} full-resolution planes");
  const bool is3D = x.dim() == 5 && x.size(2) > 1;
  FnxGrid g{}; g.B = (int)x.size(0); g.is3D = is3D; g.ref_quirks = 0; g.z_offset = 0; g.D_global = 0;
  g.D = x.dim() == 5 ? (int)x.size(2) : 1; g.H = (int)x.size(x.dim() - 2); g.W = (int)x.size(x.dim() - 1);
  c10::hip::HIPGuard guard(x.get_device());
  int tr[4] = {0, 0, 0, 0};
  for (size_t a = 0; a < trim.size(); ++a) tr[a] = (int)trim[a];
  std::vector<int64_t> osz = x.sizes().vec(); osz[1] = 1;
  if (x.dim() == 5) osz[2] -= tr[0] + tr[1];
  TORCH_CHECK(x.dim() == 5 || !(tr[0] | tr[1] | tr[2] | tr[3]), "trim needs a (B,2,D,H,W) input");
  TORCH_CHECK(x.dim() != 5 || osz[2] > 0, "trim leaves no planes");
  Tensor p = at::empty(osz, x.options());
  const size_t bytes = fnx_workspace_bytes(&g, FNX_OP_FLUIDNET);
  Tensor ws = at::empty({(int64_t)bytes}, x.options().dtype(at::kByte));
  check_status(fnx_multiscale_forward_crop(&g, packed.data_ptr(), x.data_ptr<float>(), p.data_ptr<float>(), precision_of(precision_mode),
                                           tr, ws.data_ptr(), bytes, cur_stream(x)));
  return p;
}

std::vector<Tensor> fluidnet_forward(Tensor packed, Tensor input, double normalize_threshold, const std::string& precision_mode) {
  check_field(input, "input");
  const bool is3D = input.size(1) == 6;
  TORCH_CHECK(input.size(1) == 5 || input.size(1) == 6, "input must have 5 (2D) or 6 (3D) channels [p, U, flags, density]");
  FnxGrid g{}; g.B = (int)input.size(0); g.D = (int)input.size(2); g.H = (int)input.size(3); g.W = (int)input.size(4);
  g.is3D = is3D; g.ref_quirks = 0; g.z_offset = 0; g.D_global = 0;
  c10::hip::HIPGuard guard(input.get_device());
  Tensor p = at::empty({g.B, 1, g.D, g.H, g.W}, input.options());
  Tensor U = at::empty({g.B, is3D ? 3 : 2, g.D, g.H, g.W}, input.options());
  const size_t bytes = fnx_workspace_bytes(&g, FNX_OP_FLUIDNET);
  Tensor ws = at::empty({(int64_t)bytes}, input.options().dtype(at::kByte));
  check_status(fnx_fluidnet_forward(&g, packed.data_ptr(), input.data_ptr<float>(), (float)normalize_threshold,
                                    p.data_ptr<float>(), U.data_ptr<float>(), precision_of(precision_mode), ws.data_ptr(), bytes,
                                    cur_stream(input)));
  return {p, U};
}

// one whole step of lib/simulate.py:28-171, in place on p, U, density
void simulate_step_(Tensor p, Tensor U, Tensor flags, c10::optional<Tensor> density, c10::optional<Tensor> UBC,
                    c10::optional<Tensor> UBCInvMask, c10::optional<Tensor> densityBC,
                    c10::optional<Tensor> densityBCInvMask, c10::optional<Tensor> net, double dt,
                    double maccormack_strength, bool sample_outside_fluid, double buoyancy_scale,
                    std::vector<double> gravity_vec, double operating_density, double p_tol, int jacobi_iter,
                    const std::string method, double normalize_threshold, c10::optional<Tensor> workspace,
                    int static_flags, const Geom* geom, const std::string& precision_mode, double viscosity,
                    double gravity_scale, bool correct_scalar, int periodic, c10::optional<Tensor> flags_stick) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(p, g, "p");
  TORCH_CHECK(viscosity >= 0, "Viscosity must be positive");
  TORCH_CHECK(method == "jacobi" || method == "convnet", "Simulation method not supported. Choose either convnet or jacobi.");
  TORCH_CHECK(gravity_vec.size() == 3, "gravityVec needs x, y, z");
  FnxStepParams prm{};
  prm.dt = (float)dt; prm.maccormack_strength = (float)maccormack_strength; prm.sample_outside_fluid = sample_outside_fluid;
  prm.buoyancy_scale = (float)buoyancy_scale;
  for (int a = 0; a < 3; ++a) prm.gravity_vec[a] = (float)gravity_vec[a];
  prm.operating_density = (float)operating_density; prm.p_tol = (float)p_tol; prm.jacobi_iter = jacobi_iter;
  prm.method = method == "convnet" ? 1 : 0; prm.normalize_threshold = (float)normalize_threshold;
  prm.precision_mode = precision_of(precision_mode);
  prm.viscosity = (float)viscosity; prm.gravity_scale = (float)gravity_scale; prm.correct_scalar = correct_scalar ? 1 : 0;
  prm.periodic = periodic;
  auto opt = [&](c10::optional<Tensor>& t, bool vel, const char* name) -> float* {
    if (!t.has_value() || !t->defined()) return nullptr;
    if (vel) check_vel(*t, g, name); else check_scalar(*t, g, name);
    return t->data_ptr<float>();
  };
  FnxState st{};
  st.p = p.data_ptr<float>(); st.U = U.data_ptr<float>(); st.flags = flags.data_ptr<float>();
  st.density = opt(density, false, "density");
  st.UBC = opt(UBC, true, "UBC"); st.UBCInvMask = opt(UBCInvMask, true, "UBCInvMask");
  st.densityBC = opt(densityBC, false, "densityBC"); st.densityBCInvMask = opt(densityBCInvMask, false, "densityBCInvMask");
  st.net = (net.has_value() && net->defined()) ? net->data_ptr() : nullptr;
  st.flags_stick = opt(flags_stick, false, "flags_stick");
  c10::hip::HIPGuard guard(flags.get_device());
  const size_t bytes = fnx_workspace_bytes(&g, FNX_OP_STEP);
  Tensor ws;
  const bool given = workspace.has_value() && workspace->defined() && (size_t)workspace->numel() * workspace->element_size() >= bytes;
  if (given) ws = *workspace;
  else ws = at::empty({(int64_t)bytes}, flags.options().dtype(at::kByte));
  // the mask lives in the workspace: only a caller-owned workspace carries it from one step to the next
  prm.static_flags = given ? static_flags : 0;      // (promises about the previous step need a caller-owned workspace)
  check_status(fnx_simulate_step(&g, &prm, &st, ws.data_ptr(), (size_t)ws.numel() * ws.element_size(), cur_stream(U)));
  wrote(p); wrote(U); wrote(density);
}

// ---- native z-slab driver (fnx_slab_*): communicators and the per-rank driver object ---------------------------
struct PyLoopbackGroup {
  void* g = nullptr; int nranks;
  explicit PyLoopbackGroup(int n) : nranks(n) { check_status(fnx_slab_loopback_group(&g, n)); }
  ~PyLoopbackGroup() { fnx_slab_loopback_group_free(g); }
};
struct PyPeer {                                // a rank's peer-store region (fnx_slab_peer_create)
  void* p = nullptr;
  std::string handle;
  PyPeer(int rank, int nranks, int64_t mailbox_bytes) {
    char h[FNX_PEER_HANDLE_BYTES];
    check_status(fnx_slab_peer_create(&p, rank, nranks, (size_t)mailbox_bytes, h));
    handle.assign(h, FNX_PEER_HANDLE_BYTES);
  }
  ~PyPeer() { fnx_slab_peer_free(p); }
};
struct PySlabComm {
  FnxSlabComm c{};
  std::shared_ptr<PyLoopbackGroup> keep;      // a loopback comm keeps its group alive
  std::shared_ptr<PyPeer> keep_peer;          // a peer-store comm its region
  ~PySlabComm() { fnx_slab_comm_free(&c); }
};
py::bytes slab_rccl_unique_id() {
  char id[128];
  check_status(fnx_slab_rccl_unique_id(id));
  return py::bytes(id, 128);
}
std::shared_ptr<PySlabComm> slab_comm_rccl(int rank, int nranks, const std::string& unique_id) {
  TORCH_CHECK(unique_id.size() == 128, "the RCCL unique id has 128 bytes");
  auto c = std::make_shared<PySlabComm>();
  py::gil_scoped_release nogil;               // ncclCommInitRank blocks until every rank has arrived
  check_status(fnx_slab_comm_rccl(&c->c, rank, nranks, unique_id.data()));
  return c;
}
std::shared_ptr<PySlabComm> slab_comm_loopback(std::shared_ptr<PyLoopbackGroup> group, int rank) {
  auto c = std::make_shared<PySlabComm>();
  check_status(fnx_slab_comm_loopback(&c->c, group->g, rank));
  c->keep = group;
  return c;
}
struct PySlab {
  FnxSlab* s = nullptr;
  FnxSlabConfig cfg{};
  std::shared_ptr<PySlabComm> comm;
  PySlab(int B, int H, int W, int D_global, int rank, int nranks, int halo, int sweeps_per_exchange, bool static_flags,
         int cfl_check_every, std::shared_ptr<PySlabComm> comm_, const std::string& schedule, const std::string& method, const std::string& direct_sends) : comm(comm_) {
    cfg.B = B; cfg.H = H; cfg.W = W; cfg.D_global = D_global; cfg.rank = rank; cfg.nranks = nranks; cfg.halo = halo;
    cfg.sweeps_per_exchange = sweeps_per_exchange; cfg.static_flags = static_flags ? 1 : 0; cfg.cfl_check_every = cfl_check_every;
    TORCH_CHECK(schedule == "deep_first" || schedule == "edge_first" || schedule == "last_pass" || schedule == "deep_beside",
                "unknown z-slab schedule '", schedule, "'");
    cfg.schedule = schedule == "deep_first" ? FNX_SLAB_DEEP_FIRST : (schedule == "edge_first" ? FNX_SLAB_EDGE_FIRST :
                   (schedule == "deep_beside" ? FNX_SLAB_DEEP_BESIDE : FNX_SLAB_LAST_PASS));
    TORCH_CHECK(method == "jacobi" || method == "convnet", "unknown z-slab method '", method, "'");
    cfg.method = method == "convnet" ? 1 : 0;
    TORCH_CHECK(direct_sends == "auto" || direct_sends == "never" || direct_sends == "always", "direct_sends: 'auto', 'never' or 'always'");
    cfg.direct_sends = direct_sends == "auto" ? 0 : (direct_sends == "never" ? 1 : 2);
    check_status(fnx_slab_create(&s, &cfg, comm ? &comm->c : nullptr));
  }
  ~PySlab() { fnx_slab_destroy(s); }
  std::vector<int> layout() const {
    int o, l, h, z;
    check_status(fnx_slab_layout(&cfg, &o, &l, &h, &z));
    return {o, l, h, z};
  }
  int64_t workspace_bytes() const { return (int64_t)fnx_slab_workspace_bytes(&cfg); }
  void stats_enable(bool on) { check_status(fnx_slab_stats_enable(s, on ? 1 : 0)); }
  py::dict stats_read() {
    FnxSlabStats t{};
    check_status(fnx_slab_stats_read(s, &t));
    py::dict d;
    d["bytes_per_neighbour"] = t.bytes_per_neighbour; d["wait_ms"] = t.wait_ms; d["exchanges"] = (int64_t)t.exchanges;
    return d;
  }
  void step(Tensor p, Tensor U, Tensor flags, Tensor density, c10::optional<Tensor> UBC, c10::optional<Tensor> UBCInvMask,
            c10::optional<Tensor> densityBC, c10::optional<Tensor> densityBCInvMask, double dt, double maccormack_strength,
            bool sample_outside_fluid, double buoyancy_scale, std::vector<double> gravity_vec, double operating_density,
            double p_tol, int jacobi_iter, Tensor workspace, c10::optional<Tensor> net, const std::string& precision_mode,
            double normalize_threshold) {
    check_field(U, "U");
    Geom none;
    FnxGrid g = grid_of(flags, true, &none);
    check_vel(U, g, "U"); check_scalar(p, g, "p"); check_scalar(density, g, "density");
    const std::vector<int> l = layout();
    TORCH_CHECK(g.B == cfg.B && g.H == cfg.H && g.W == cfg.W && g.D == l[0] + l[1] + l[2], "the tensors are not this rank's slab (",
                cfg.B, " x ", l[0] + l[1] + l[2], " x ", cfg.H, " x ", cfg.W, " with ghost planes)");
    TORCH_CHECK(gravity_vec.size() == 3, "gravityVec needs x, y, z");
    TORCH_CHECK(workspace.is_cuda() && workspace.is_contiguous(), "workspace must be a contiguous GPU tensor");
    FnxStepParams prm{};
    prm.dt = (float)dt; prm.maccormack_strength = (float)maccormack_strength; prm.sample_outside_fluid = sample_outside_fluid;
    prm.buoyancy_scale = (float)buoyancy_scale;
    for (int a = 0; a < 3; ++a) prm.gravity_vec[a] = (float)gravity_vec[a];
    prm.operating_density = (float)operating_density; prm.p_tol = (float)p_tol; prm.jacobi_iter = jacobi_iter; prm.method = 0;
    const bool cnn = net.has_value() && net->defined();
    if (cnn) {                                       // the CNN projection (a driver created with method = "convnet")
      TORCH_CHECK(net->is_cuda() && net->is_contiguous(), "net: the packed weights (scalenet_pack) on the GPU");
      prm.method = 1; prm.precision_mode = precision_of(precision_mode); prm.normalize_threshold = (float)normalize_threshold;
    }
    auto opt = [&](c10::optional<Tensor>& t, bool vel, const char* name) -> float* {
      if (!t.has_value() || !t->defined()) return nullptr;
      if (vel) check_vel(*t, g, name); else check_scalar(*t, g, name);
      return t->data_ptr<float>();
    };
    FnxState st{};
    st.p = p.data_ptr<float>(); st.U = U.data_ptr<float>(); st.flags = flags.data_ptr<float>(); st.density = density.data_ptr<float>();
    st.UBC = opt(UBC, true, "UBC"); st.UBCInvMask = opt(UBCInvMask, true, "UBCInvMask");
    st.densityBC = opt(densityBC, false, "densityBC"); st.densityBCInvMask = opt(densityBCInvMask, false, "densityBCInvMask");
    if (cnn) st.net = net->data_ptr();
    c10::hip::HIPGuard guard(flags.get_device());
    check_status(fnx_slab_step(s, &prm, &st, workspace.data_ptr(), (size_t)workspace.numel() * workspace.element_size(), cur_stream(U)));
  }
};

// `nsweeps` more sweeps on an existing pressure field (in place) -- used by the z-slab driver
void jacobi_sweeps_(Tensor flags, Tensor div, Tensor p, bool is3D, int nsweeps, c10::optional<Tensor> workspace,
                    bool reuse_mask, const Geom* geom, bool from_zero) {
  FnxGrid g = grid_of(flags, is3D, geom);
  check_scalar(div, g, "div"); check_scalar(p, g, "p");
  c10::hip::HIPGuard guard(flags.get_device());
  const size_t bytes = fnx_workspace_bytes(&g, FNX_OP_JACOBI);
  Tensor ws;
  if (workspace.has_value() && workspace->defined() && (size_t)workspace->numel() * workspace->element_size() >= bytes) ws = *workspace;
  else { ws = at::empty({(int64_t)bytes}, flags.options().dtype(at::kByte)); reuse_mask = false; }
  check_status(fnx_jacobi_sweeps_ex(&g, flags.data_ptr<float>(), div.data_ptr<float>(), p.data_ptr<float>(), nsweeps,
                                    ws.data_ptr(), (size_t)ws.numel() * ws.element_size(), (reuse_mask ? 1 : 0) | (from_zero ? 2 : 0),
                                    cur_stream(flags)));
  wrote(p);
}

// one pass (1|2 sweeps) p_in -> p_out on the output planes [k_begin, k_end) -- z-slab driver (overlap with exchange)
void jacobi_pass_(Tensor flags, Tensor div, c10::optional<Tensor> p_in_opt, Tensor p_out, int nsweeps, int k_begin,
                  int k_end, Tensor workspace, bool reuse_mask, int k_begin2, const Geom* geom, int layout) {
  FnxGrid g = grid_of(flags, true, geom);
  const bool zero = !(p_in_opt.has_value() && p_in_opt->defined());     // None: p = 0 (first pass of a solve)
  Tensor p_in = zero ? p_out : *p_in_opt;
  check_scalar(div, g, "div"); check_scalar(p_in, g, "p_in"); check_scalar(p_out, g, "p_out");
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_jacobi_pass_layout(&g, flags.data_ptr<float>(), div.data_ptr<float>(), zero ? nullptr : p_in.data_ptr<float>(),
                                      p_out.data_ptr<float>(), nsweeps, k_begin, k_end, k_begin2, layout, workspace.data_ptr(),
                                      (size_t)workspace.numel() * workspace.element_size(), reuse_mask ? 1 : 0, cur_stream(flags)));
  wrote(p_out);
}

// may two-sweep passes on this 3D grid hand each other the pressure in the row-quad layout (jacobi_pass_'s `layout`)?
bool jacobi_quad_ok(int B, int D, int H, int W) {
  FnxGrid g{B, D, H, W, 1, 0, 0, 0};
  return fnx_jacobi_quad_ok(&g) != 0;
}

// fnx_jacobi_pass_mirror: `mirror` / `mirror2` are flat fp32 tensors of B * planes * H * W floats (sample stride planes * H * W)
void jacobi_pass_mirror_(Tensor flags, Tensor div, Tensor p_in, Tensor p_out, int k_begin, int k_end, Tensor workspace, bool reuse_mask,
                         Tensor mirror, int k_first, int planes, int k_begin2, c10::optional<Tensor> mirror2, int k_first2, const Geom* geom,
                         int layout) {
  FnxGrid g = grid_of(flags, true, geom);
  check_scalar(div, g, "div"); check_scalar(p_in, g, "p_in"); check_scalar(p_out, g, "p_out");
  const int64_t need = (int64_t)g.B * planes * g.H * g.W;
  auto chk = [&](const Tensor& t) { TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat && t.numel() >= need, "mirror: a contiguous fp32 GPU tensor of B * planes * H * W floats"); };
  chk(mirror);
  FnxPlaneMirror m{};
  m.out[0][0] = mirror.data_ptr<float>(); m.k_first[0] = k_first; m.planes = planes; m.sample_stride = (size_t)planes * g.H * g.W;
  if (mirror2.has_value() && mirror2->defined()) { chk(*mirror2); m.out[1][0] = mirror2->data_ptr<float>(); m.k_first[1] = k_first2; }
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_jacobi_pass_mirror(&g, flags.data_ptr<float>(), div.data_ptr<float>(), p_in.data_ptr<float>(), p_out.data_ptr<float>(),
                                      k_begin, k_end, k_begin2, layout, &m, workspace.data_ptr(),
                                      (size_t)workspace.numel() * workspace.element_size(), reuse_mask ? 1 : 0, cur_stream(flags)));
  wrote(p_out); wrote(mirror); wrote(mirror2);
}

int64_t jacobi_workspace_bytes(int B, int D, int H, int W, bool is3D) {
  FnxGrid g{B, D, H, W, is3D ? 1 : 0, 0, 0, 0};
  return (int64_t)fnx_workspace_bytes(&g, FNX_OP_JACOBI);
}

static FnxState make_state(const FnxGrid& g, Tensor& p, Tensor& U, Tensor& flags, c10::optional<Tensor>& density,
                           c10::optional<Tensor>& UBC, c10::optional<Tensor>& UBCInvMask,
                           c10::optional<Tensor>& densityBC, c10::optional<Tensor>& densityBCInvMask) {
  auto opt = [&](c10::optional<Tensor>& t, bool vel, const char* name) -> float* {
    if (!t.has_value() || !t->defined()) return nullptr;
    if (vel) check_vel(*t, g, name); else check_scalar(*t, g, name);
    return t->data_ptr<float>();
  };
  FnxState st{};
  st.p = p.data_ptr<float>(); st.U = U.data_ptr<float>(); st.flags = flags.data_ptr<float>();
  st.density = opt(density, false, "density");
  st.UBC = opt(UBC, true, "UBC"); st.UBCInvMask = opt(UBCInvMask, true, "UBCInvMask");
  st.densityBC = opt(densityBC, false, "densityBC"); st.densityBCInvMask = opt(densityBCInvMask, false, "densityBCInvMask");
  st.net = nullptr;
  st.bc_class = nullptr;
  return st;
}

static const unsigned char* bc_class_ptr(c10::optional<Tensor>& t, const Tensor& flags) {
  if (!t.has_value() || !t->defined()) return nullptr;
  TORCH_CHECK(t->scalar_type() == at::kByte && t->is_contiguous() && t->numel() == flags.numel() && t->device() == flags.device(),
              "bc_class: expected the uint8 tensor bc_classify returned for this grid");
  return t->data_ptr<unsigned char>();
}

// class map of the (static) BC arrays for pre_projection_ / post_projection_ (FnxState.bc_class)
Tensor bc_classify(Tensor flags, bool is3D, c10::optional<Tensor> UBC, c10::optional<Tensor> UBCInvMask,
                   c10::optional<Tensor> densityBC, c10::optional<Tensor> densityBCInvMask) {
  FnxGrid g = grid_of(flags, is3D, nullptr);
  Tensor dummy;
  c10::optional<Tensor> none;
  FnxState st{};
  auto opt = [&](c10::optional<Tensor>& t, bool vel, const char* name) -> const float* {
    if (!t.has_value() || !t->defined()) return nullptr;
    if (vel) check_vel(*t, g, name); else check_scalar(*t, g, name);
    return t->data_ptr<float>();
  };
  st.UBC = opt(UBC, true, "UBC"); st.UBCInvMask = opt(UBCInvMask, true, "UBCInvMask");
  st.densityBC = opt(densityBC, false, "densityBC"); st.densityBCInvMask = opt(densityBCInvMask, false, "densityBCInvMask");
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor cls = at::empty(flags.sizes(), flags.options().dtype(at::kByte));
  check_status(fnx_bc_classify(&g, &st, cls.data_ptr<unsigned char>(), cur_stream(flags)));
  return cls;
}

// simulate.py:96-133 (+ divergence): U_adv/rho_adv -> U, density (written), returns div
Tensor pre_projection_(Tensor U_adv, c10::optional<Tensor> rho_adv, Tensor p, Tensor U, Tensor flags,
                       c10::optional<Tensor> density, c10::optional<Tensor> UBC, c10::optional<Tensor> UBCInvMask,
                       c10::optional<Tensor> densityBC, c10::optional<Tensor> densityBCInvMask, double dt,
                       double buoyancy_scale, std::vector<double> gravity_vec, double operating_density,
                       bool jacobi_method, c10::optional<Tensor> bc_class, const Geom* geom) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_vel(U_adv, g, "U_adv"); check_scalar(p, g, "p");
  FnxStepParams prm{};
  prm.dt = (float)dt; prm.buoyancy_scale = (float)buoyancy_scale;
  for (int a = 0; a < 3; ++a) prm.gravity_vec[a] = (float)gravity_vec[a];
  prm.operating_density = (float)operating_density; prm.method = jacobi_method ? 0 : 1;
  FnxState st = make_state(g, p, U, flags, density, UBC, UBCInvMask, densityBC, densityBCInvMask);
  const float* ra = nullptr;
  if (rho_adv.has_value() && rho_adv->defined()) { check_scalar(*rho_adv, g, "rho_adv"); ra = rho_adv->data_ptr<float>(); }
  st.bc_class = bc_class_ptr(bc_class, flags);
  c10::hip::HIPGuard guard(flags.get_device());
  Tensor div = at::empty_like(flags);
  check_status(fnx_pre_projection(&g, &prm, &st, U_adv.data_ptr<float>(), ra, div.data_ptr<float>(), cur_stream(U)));
  return div;
}

void post_projection_(Tensor p, Tensor U, Tensor flags, c10::optional<Tensor> density, c10::optional<Tensor> UBC,
                      c10::optional<Tensor> UBCInvMask, c10::optional<Tensor> densityBC,
                      c10::optional<Tensor> densityBCInvMask, c10::optional<Tensor> bc_class, const Geom* geom,
                      bool density_bc_applied) {
  check_field(U, "U");
  FnxGrid g = grid_of(flags, U.size(1) == 3, geom);
  check_vel(U, g, "U"); check_scalar(p, g, "p");
  FnxState st = make_state(g, p, U, flags, density, UBC, UBCInvMask, densityBC, densityBCInvMask);
  st.bc_class = bc_class_ptr(bc_class, flags);
  st.density_bc_applied = density_bc_applied ? 1 : 0;
  c10::hip::HIPGuard guard(flags.get_device());
  check_status(fnx_post_projection(&g, &st, cur_stream(U)));
  wrote(U); wrote(density);
}

int64_t step_workspace_bytes(int B, int D, int H, int W, bool is3D) {
  FnxGrid g{B, D, H, W, is3D ? 1 : 0, 0, 0, 0};
  return (int64_t)fnx_workspace_bytes(&g, FNX_OP_STEP);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  // Every compute entry point releases the GIL while it checks, allocates and enqueues: arguments are
  // converted before the guard is taken and results after it is dropped; nothing inside touches Python objects.
  using NoGil = py::call_guard<py::gil_scoped_release>;
  const auto GEOM = py::arg("geom") = py::none();
  py::class_<Geom>(m, "Geom", "per-call 3D geometry options: ref_quirks, z-slab view (z_offset, D_global), compute window [k_begin, k_end)")
      .def(py::init([](bool ref_quirks, int z_offset, int D_global, int k_begin, int k_end) {
             Geom g; g.ref_quirks = ref_quirks; g.z_offset = z_offset; g.D_global = D_global; g.k_begin = k_begin; g.k_end = k_end;
             return g;
           }),
           py::arg("ref_quirks") = false, py::arg("z_offset") = 0, py::arg("D_global") = 0, py::arg("k_begin") = 0,
           py::arg("k_end") = 0)
      .def_readwrite("ref_quirks", &Geom::ref_quirks)
      .def_readwrite("z_offset", &Geom::z_offset)
      .def_readwrite("D_global", &Geom::D_global)
      .def_readwrite("k_begin", &Geom::k_begin)
      .def_readwrite("k_end", &Geom::k_end);
  // reference entry points (fluids_init.cpp:1009-1014): same names, same positional arguments; `out` / `geom` are
  // optional trailing extras
  m.def("advect_scalar", &advect_scalar, "Advect Scalar", py::arg("dt"), py::arg("src"), py::arg("U"), py::arg("flags"),
        py::arg("method"), py::arg("boundary_width"), py::arg("sample_outside_fluid"), py::arg("maccormack_strength"),
        py::arg("out") = py::none(), GEOM, py::arg("plan") = "auto", NoGil());
  m.def("advect_step", &advect_step, py::arg("dt"), py::arg("density"), py::arg("U"), py::arg("flags"),
        py::arg("sample_outside_fluid"), py::arg("maccormack_strength"), py::arg("out_density") = py::none(),
        py::arg("out_U") = py::none(), GEOM, py::arg("plan") = "auto", NoGil());
  m.def("advect_vel", &advect_vel, "Advect Velocity", py::arg("dt"), py::arg("orig"), py::arg("U"), py::arg("flags"),
        py::arg("method"), py::arg("boundary_width"), py::arg("maccormack_strength"), py::arg("out") = py::none(), GEOM,
        py::arg("plan") = "auto", NoGil());
  m.def("solve_linear_system", &solve_linear_system, "Solve Linear System using Jacobi's method", py::arg("flags"),
        py::arg("div"), py::arg("is3D"), py::arg("p_tol"), py::arg("max_iter"), py::arg("verbose"), GEOM, NoGil());
  // operators the reference implements in Python
  m.def("velocity_divergence", &velocity_divergence, py::arg("U"), py::arg("flags"), GEOM, NoGil());
  m.def("velocity_update_", &velocity_update_, py::arg("pressure"), py::arg("U"), py::arg("flags"), GEOM, NoGil());
  m.def("add_buoyancy_", &add_buoyancy_, py::arg("U"), py::arg("flags"), py::arg("density"), py::arg("gravity"),
        py::arg("rho_star"), py::arg("dt"), GEOM, NoGil());
  m.def("correct_scalar_", &correct_scalar_, py::arg("dt"), py::arg("src"), py::arg("div"), py::arg("flags"), NoGil());
  m.def("add_gravity_", &add_gravity_, py::arg("U"), py::arg("flags"), py::arg("gravity"), py::arg("dt"), GEOM, NoGil());
  m.def("add_viscosity_", &add_viscosity_, NoGil());
  m.def("set_wall_bcs_", &set_wall_bcs_, py::arg("U"), py::arg("flags"), GEOM, NoGil());
  m.def("set_wall_bcs_stick_", &set_wall_bcs_stick_, NoGil());
  m.def("set_const_vals_", &set_const_vals_, NoGil());
  m.def("flags_to_occupancy", &flags_to_occupancy, NoGil());
  m.def("max_abs", &max_abs, NoGil());
  m.def("empty_domain_", &empty_domain_, py::arg("flags"), py::arg("boundary_width"), GEOM, NoGil());
  m.def("velocity_divergence_backward", &velocity_divergence_backward, py::arg("grad_div"), py::arg("flags"), py::arg("is3D"), GEOM, NoGil());
  m.def("velocity_update_backward", &velocity_update_backward, py::arg("grad_U_out"), py::arg("flags"), GEOM, NoGil());
  m.def("create_cylinder_", &create_cylinder_, NoGil());
  m.def("create_box2d_", &create_box2d_, NoGil());
  m.def("get_centered", &get_centered, NoGil());
  m.def("scalenet_pack", &scalenet_pack, NoGil());
  m.def("multiscale_forward", &multiscale_forward, py::arg("packed"), py::arg("x"), py::arg("precision_mode") = "fp32",
        py::arg("trim") = std::vector<int64_t>(), NoGil());
  m.def("fluidnet_forward", &fluidnet_forward, py::arg("packed"), py::arg("input"), py::arg("normalize_threshold"),
        py::arg("precision_mode") = "fp32", NoGil());
  m.def("simulate_step_", &simulate_step_, py::arg("p"), py::arg("U"), py::arg("flags"), py::arg("density"), py::arg("UBC"),
        py::arg("UBCInvMask"), py::arg("densityBC"), py::arg("densityBCInvMask"), py::arg("net"), py::arg("dt"),
        py::arg("maccormack_strength"), py::arg("sample_outside_fluid"), py::arg("buoyancy_scale"), py::arg("gravity_vec"),
        py::arg("operating_density"), py::arg("p_tol"), py::arg("jacobi_iter"), py::arg("method"),
        py::arg("normalize_threshold"), py::arg("workspace") = py::none(), py::arg("static_flags") = 0, GEOM,
        py::arg("precision_mode") = "fp32", py::arg("viscosity") = 0.0, py::arg("gravity_scale") = 0.0,
        py::arg("correct_scalar") = false, py::arg("periodic") = 0, py::arg("flags_stick") = py::none(), NoGil());
  m.def("step_workspace_bytes", &step_workspace_bytes);
  m.def("jacobi_sweeps_", &jacobi_sweeps_, py::arg("flags"), py::arg("div"), py::arg("p"), py::arg("is3D"), py::arg("nsweeps"),
        py::arg("workspace") = py::none(), py::arg("reuse_mask") = false, GEOM, py::arg("from_zero") = false, NoGil());
  m.def("jacobi_workspace_bytes", &jacobi_workspace_bytes);
  m.def("jacobi_pass_", &jacobi_pass_, py::arg("flags"), py::arg("div"), py::arg("p_in"), py::arg("p_out"), py::arg("nsweeps"),
        py::arg("k_begin"), py::arg("k_end"), py::arg("workspace"), py::arg("reuse_mask"), py::arg("k_begin2") = -1, GEOM,
        py::arg("layout") = 0, NoGil());
  m.def("jacobi_pass_mirror_", &jacobi_pass_mirror_, py::arg("flags"), py::arg("div"), py::arg("p_in"), py::arg("p_out"), py::arg("k_begin"),
        py::arg("k_end"), py::arg("workspace"), py::arg("reuse_mask"), py::arg("mirror"), py::arg("k_first"), py::arg("planes"),
        py::arg("k_begin2") = -1, py::arg("mirror2") = py::none(), py::arg("k_first2") = 0, GEOM, py::arg("layout") = 0, NoGil(),
        "two sweeps on planes [k_begin, k_end) (+ a second range) whose output planes [k_first, k_first + planes) also go to `mirror` (fnx_jacobi_pass_mirror)");
  m.def("jacobi_pass_mirror_ok", [](int B, int D, int H, int W, int planes_per_range, bool two_ranges, int layout) {
    FnxGrid g{}; g.B = B; g.D = D; g.H = H; g.W = W; g.is3D = 1;
    return fnx_jacobi_pass_mirror_ok(&g, planes_per_range, two_ranges ? 1 : 0, layout) != 0;
  });
  m.def("jacobi_quad_ok", &jacobi_quad_ok);
  m.def("pre_projection_", &pre_projection_, py::arg("U_adv"), py::arg("rho_adv"), py::arg("p"), py::arg("U"), py::arg("flags"),
        py::arg("density"), py::arg("UBC"), py::arg("UBCInvMask"), py::arg("densityBC"), py::arg("densityBCInvMask"),
        py::arg("dt"), py::arg("buoyancy_scale"), py::arg("gravity_vec"), py::arg("operating_density"),
        py::arg("jacobi_method"), py::arg("bc_class") = py::none(), GEOM, NoGil());
  m.def("post_projection_", &post_projection_, py::arg("p"), py::arg("U"), py::arg("flags"), py::arg("density"), py::arg("UBC"),
        py::arg("UBCInvMask"), py::arg("densityBC"), py::arg("densityBCInvMask"), py::arg("bc_class") = py::none(), GEOM,
        py::arg("density_bc_applied") = false, NoGil());
  m.def("bc_classify", &bc_classify, "uint8 class map of static BC arrays (FnxState.bc_class)", NoGil());
  py::class_<PyLoopbackGroup, std::shared_ptr<PyLoopbackGroup>>(m, "SlabLoopbackGroup", "in-process communicator group: n slabs driven by n host threads")
      .def(py::init<int>(), py::arg("nranks"))
      .def("set_timeout", [](PyLoopbackGroup& g, double seconds) { check_status(fnx_slab_loopback_group_set_timeout(g.g, seconds)); }, py::arg("seconds"),
           "how long a rank waits for a peer before its call fails (the group stays usable); default 120 s")
      .def("reset", [](PyLoopbackGroup& g) { check_status(fnx_slab_loopback_group_reset(g.g)); },
           "clear an abort (no rank may be inside a call of the group)");
  py::class_<PySlabComm, std::shared_ptr<PySlabComm>>(m, "SlabComm", "ghost-plane communicator of the native z-slab driver (FnxSlabComm)")
      .def("failed", [](PySlabComm& c) { return c.keep_peer ? fnx_slab_peer_failed(c.keep_peer->p) != FNX_OK : false; },
           "peer-store: an exchange of this rank timed out / the group was aborted (ask after synchronising; other transports report through their calls)")
      // the control-path collectives of the table, callable on their own (every rank of the communicator must call them)
      .def("allreduce_sum_", [](PySlabComm& c, Tensor x) {
             TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.scalar_type() == at::kFloat, "allreduce: a contiguous fp32 GPU tensor");
             c10::hip::HIPGuard guard(x.get_device());
             py::gil_scoped_release nogil;
             check_status(c.c.allreduce_sum(c.c.ctx, x.data_ptr<float>(), (int)x.numel(), cur_stream(x)));
           }, py::arg("x"))
      .def("allreduce_max_", [](PySlabComm& c, Tensor x) {
             TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.scalar_type() == at::kFloat, "allreduce: a contiguous fp32 GPU tensor");
             c10::hip::HIPGuard guard(x.get_device());
             py::gil_scoped_release nogil;
             check_status(c.c.allreduce_max(c.c.ctx, x.data_ptr<float>(), (int)x.numel(), cur_stream(x)));
           }, py::arg("x"));
  m.def("slab_rccl_unique_id", &slab_rccl_unique_id);
  m.def("slab_comm_rccl", &slab_comm_rccl, py::arg("rank"), py::arg("nranks"), py::arg("unique_id"));
  m.def("slab_comm_loopback", &slab_comm_loopback, py::arg("group"), py::arg("rank"));
  py::class_<PyPeer, std::shared_ptr<PyPeer>>(m, "SlabPeer", "a rank's peer-store region: flags + mailbox its z-neighbours map (fnx_slab_peer_create)")
      .def(py::init<int, int, int64_t>(), py::arg("rank"), py::arg("nranks"), py::arg("mailbox_bytes"))
      .def_property_readonly("handle", [](PyPeer& p) { return py::bytes(p.handle); }, "the bytes both neighbours need (FNX_PEER_HANDLE_BYTES)")
      .def("set_timeout", [](PyPeer& p, double seconds) { check_status(fnx_slab_peer_set_timeout(p.p, seconds)); }, py::arg("seconds"))
      .def("failed", [](PyPeer& p) { return fnx_slab_peer_failed(p.p) != FNX_OK; },
           "a device-side wait of this rank's exchanges has timed out, or the group was aborted (ask after synchronising the stream)");
  m.def("slab_comm_peer", [](std::shared_ptr<PyPeer> peer, py::object handle_lo, py::object handle_hi) {
    std::string lo = handle_lo.is_none() ? std::string() : handle_lo.cast<std::string>();
    std::string hi = handle_hi.is_none() ? std::string() : handle_hi.cast<std::string>();
    TORCH_CHECK((lo.empty() || lo.size() == FNX_PEER_HANDLE_BYTES) && (hi.empty() || hi.size() == FNX_PEER_HANDLE_BYTES), "a peer handle has ",
                FNX_PEER_HANDLE_BYTES, " bytes");
    auto c = std::make_shared<PySlabComm>();
    check_status(fnx_slab_comm_peer(&c->c, peer->p, lo.empty() ? nullptr : lo.data(), hi.empty() ? nullptr : hi.data()));
    c->keep_peer = peer;
    return c;
  }, py::arg("peer"), py::arg("handle_lo"), py::arg("handle_hi"), "peer-store communicator: device stores into the neighbours' mapped mailboxes + flags");
  m.def("slab_comm_link_model", [](double latency_us, double gbps) {
    auto c = std::make_shared<PySlabComm>();
    check_status(fnx_slab_comm_link_model(&c->c, latency_us, gbps));
    return c;
  }, py::arg("latency_us"), py::arg("gbytes_per_s"), "rehearsal communicator: one process runs a middle rank's step against an assumed link");
  m.def("slab_comm_probe", [](std::shared_ptr<PySlabComm> comm, int64_t bytes, int reps, Tensor scratch) {
    TORCH_CHECK(scratch.is_cuda() && scratch.is_contiguous() && (int64_t)scratch.numel() * scratch.element_size() >= 4 * bytes,
                "slab_comm_probe: scratch must be a contiguous GPU tensor of at least 4 * bytes");
    c10::hip::HIPGuard guard(scratch.get_device());
    float ms = 0.f;
    { py::gil_scoped_release nogil;
      check_status(fnx_slab_comm_probe(&comm->c, scratch.data_ptr(), (size_t)bytes, reps, &ms, cur_stream(scratch))); }
    return (double)ms;
  }, py::arg("comm"), py::arg("bytes"), py::arg("reps"), py::arg("scratch"),
        "average ms of one ghost exchange of `bytes` bytes with each neighbour (every rank must call it)");
  py::class_<PySlab>(m, "SlabDriver", "native z-slab driver of the 3D Jacobi step (fnx_slab_create / fnx_slab_step)")
      .def(py::init<int, int, int, int, int, int, int, int, bool, int, std::shared_ptr<PySlabComm>, const std::string&, const std::string&, const std::string&>(), py::arg("B"), py::arg("H"),
           py::arg("W"), py::arg("D_global"), py::arg("rank"), py::arg("nranks"), py::arg("halo"), py::arg("sweeps_per_exchange"),
           py::arg("static_flags") = false, py::arg("cfl_check_every") = 0, py::arg("comm") = nullptr, py::arg("schedule") = "deep_first",
           py::arg("method") = "jacobi", py::arg("direct_sends") = "auto")
      .def("stats_enable", &PySlab::stats_enable, py::arg("on"))
      .def("stats_read", &PySlab::stats_read, "dict(bytes_per_neighbour, wait_ms, exchanges) since stats_enable(True); synchronises")
      .def("layout", &PySlab::layout, "(owned planes, ghost planes below, above, global plane of local plane 0)")
      .def("workspace_bytes", &PySlab::workspace_bytes)
      .def("step", &PySlab::step, py::arg("p"), py::arg("U"), py::arg("flags"), py::arg("density"), py::arg("UBC"), py::arg("UBCInvMask"),
           py::arg("densityBC"), py::arg("densityBCInvMask"), py::arg("dt"), py::arg("maccormack_strength"),
           py::arg("sample_outside_fluid"), py::arg("buoyancy_scale"), py::arg("gravity_vec"), py::arg("operating_density"),
           py::arg("p_tol"), py::arg("jacobi_iter"), py::arg("workspace"), py::arg("net") = py::none(), py::arg("precision_mode") = "fp32",
           py::arg("normalize_threshold") = 1e-5, NoGil());
  m.def("device_name", []() { const char* n = fnx_device_name(); return std::string(n ? n : ""); });
  m.def("abi_version", &fnx_abi_version);
  m.def("profile_enable", [](bool on, bool runs) { fnx_profile_enable(on ? (runs ? 2 : 1) : 0); }, py::arg("on"), py::arg("runs") = false);
  m.def("roctx_enable", [](bool on) { check_status(fnx_roctx_enable(on ? 1 : 0)); });
  m.def("profile_read_work", [](int tag) { double w = 0; fnx_profile_read_work(tag, &w); return w; });
  m.def("profile_read", [](int tag) { double ms = 0; int n = 0; fnx_profile_read(tag, &ms, &n); return std::make_pair(ms, n); });
}
