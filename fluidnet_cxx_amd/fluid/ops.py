"""Operator wrappers with the reference's Python signatures.

Each function cites the reference function it replaces; argument checks mirror the reference's asserts
(AssertionError for shape errors raised on the Python side, RuntimeError from the native side).

The 3D-capable operators take one extra keyword-only argument the reference does not have: `geom`, an
`ext.Geom(ref_quirks=..., z_offset=..., D_global=..., k_begin=..., k_end=...)` (see `Geom` in `fluid/__init__.py`).
It is per call -- the extension keeps no mutable state.
"""
import torch

from .._ext import ext


def _check_advection_method(method):
    # cpp/advection.py:4-7
    assert method in ("eulerFluidNet", "maccormackFluidNet"), \
        "Error: Advection method not supported. Options are: maccormackFluidNet, eulerFluidNet"


def _check5(*ts):
    for t in ts:
        assert t.dim() == 5, "Dimension mismatch"
        assert t.is_contiguous(), "Input is not contiguous"


def getDx(self):
    """lib/fluid/grid.py:3-6"""
    return 1.0 / max(self.size(2), self.size(3), self.size(4))


def getCentered(self):
    """lib/fluid/grid.py:7-32: MAC velocity (B,2|3,D,H,W) -> cell-centred (B,3,D,H,W), 0 on the last column/row/plane."""
    assert self.dim() == 5, "Dimension mismatch"
    return ext.get_centered(self.contiguous())


def advectScalar(dt, src, U, flags, method="maccormackFluidNet", boundary_width=1, sample_outside_fluid=False,
                 maccormack_strength=0.75, *, geom=None, plan="auto"):
    """cpp/advection.py:14-66 -> pybind advect_scalar (cpp/fluids_init.cpp:265-382). 3D is supported here.
    `plan` ('auto' | 'tiles' | 'cells'): the kernel family (same bits; 'auto' = LDS tiles in 3D and on 2D grids of >= 1.5 M cells)."""
    _check_advection_method(method)
    _check5(src, U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    is3D = U.size(1) == 3
    if not is3D:
        assert flags.size(2) == 1, "2D velocity field but zdepth > 1"
        assert U.size(1) == 2, "2D velocity field must have only 2 channels"
    assert U.size(0) == flags.size(0) and U.shape[2:] == flags.shape[2:], "Size mismatch"
    return ext.advect_scalar(float(dt), src, U, flags, method, int(boundary_width), bool(sample_outside_fluid),
                             float(maccormack_strength), None, geom, plan)


def advectVelocity(dt, orig, U, flags, method="maccormackFluidNet", boundary_width=1, maccormack_strength=0.75, *,
                   geom=None, plan="auto"):
    """cpp/advection.py:68-118 -> pybind advect_vel (cpp/fluids_init.cpp:656-807).  `plan` as in advectScalar (tiles exist for
    self-advection, `orig is U`, which is what simulate.py:93 passes unless viscosity > 0)."""
    _check_advection_method(method)
    _check5(orig, U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    is3D = U.size(1) == 3
    if not is3D:
        assert flags.size(2) == 1, "2D velocity field but zdepth > 1"
        assert orig.size(1) == 2 and U.size(1) == 2, "2D velocity field must have only 2 channels"
    assert U.shape == orig.shape and U.size(0) == flags.size(0) and U.shape[2:] == flags.shape[2:], "Size mismatch"
    return ext.advect_vel(float(dt), orig, U, flags, method, int(boundary_width), float(maccormack_strength), None, geom, plan)


def correctScalar(dt, src, div, flags):
    """cpp/advection.py:9-12 (off in all shipped configs): src += dt*0.5*src*div on fluid cells, in place.
    The reference's statement is plain tensor arithmetic, so it takes strided views and broadcastable `div` / `flags`; here those are
    brought to the native operator's form (5-D contiguous fp32, one shape) and the result is copied back into `src`'s own storage."""
    for t, name in ((src, "src"), (div, "div"), (flags, "flags")):
        assert t.is_cuda, "fluidnet_cxx_amd has no CPU path: tensors must be on the GPU"
        assert t.dtype == torch.float32, f"{name} must be float32"
    assert src.dim() == 5, "Dimension mismatch"
    shape = src.shape
    assert torch.broadcast_shapes(shape, div.shape, flags.shape) == shape, "Size mismatch"       # (src is written in place: it sets the shape)
    work = src if src.is_contiguous() else src.contiguous()
    ext.correct_scalar_(float(dt), work, div.expand(shape).contiguous(), flags.expand(shape).contiguous())
    if work is not src:
        src.copy_(work)


def solveLinearSystemJacobi(flags, div, is_3d=False, p_tol=1e-5, max_iter=1000, verbose=False, *, geom=None):
    """cpp/solve_linear_sys.py:4-40 -> pybind solve_linear_system (cpp/fluids_init.cpp:809-1004).
    Returns (p, residual) with residual a 0-dim tensor, like the reference."""
    _check5(div, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    assert div.shape == flags.shape, "Size mismatch"
    p, res = ext.solve_linear_system(flags, div, bool(is_3d), float(p_tol), int(max_iter), bool(verbose), geom)
    return p, res


# ---- autograd: the reference's operators are chains of differentiable ATen ops, and its training graph goes through
# velocityUpdate -> setWallBcs -> velocityDivergence (model.py:190-227, fluid_net_train.py:366).  Here each operator is one
# native launch, so its adjoint is one too (fnx_velocity_divergence_backward, fnx_velocity_update_backward; setWallBcs is
# its own adjoint).  The in-place operators stay in place (`mark_dirty`), like the reference's slice assignments.
class _DivergenceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, flags, geom):
        ctx.save_for_backward(flags)
        ctx.geom, ctx.is3d = geom, U.size(1) == 3
        return ext.velocity_divergence(U, flags, geom)

    @staticmethod
    def backward(ctx, g):
        (flags,) = ctx.saved_tensors
        return ext.velocity_divergence_backward(g.contiguous(), flags, ctx.is3d, ctx.geom), None, None


class _VelocityUpdateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pressure, U, flags, geom):
        ctx.save_for_backward(flags)
        ctx.geom = geom
        ext.velocity_update_(pressure, U, flags, geom)
        ctx.mark_dirty(U)
        return U

    @staticmethod
    def backward(ctx, g):
        (flags,) = ctx.saved_tensors
        gU, gp = ext.velocity_update_backward(g.contiguous(), flags, ctx.geom)
        return gp, gU, None, None


class _SetWallBcsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, flags, geom):
        ctx.save_for_backward(flags)
        ctx.geom = geom
        ext.set_wall_bcs_(U, flags, geom)
        ctx.mark_dirty(U)
        return U

    @staticmethod
    def backward(ctx, g):
        (flags,) = ctx.saved_tensors
        g = g.contiguous().clone()
        ext.set_wall_bcs_(g, flags, ctx.geom)              # zeroing a flag-dependent set of entries is its own adjoint
        return g, None, None


def _needs_grad(*ts, geom=None):
    need = torch.is_grad_enabled() and any(t.requires_grad for t in ts)
    # the adjoint entry points cover whole fields; a windowed forward (planes outside [k_begin, k_end) left as they were)
    # has no matching backward
    assert not (need and geom is not None and (geom.k_begin != 0 or geom.k_end != 0)), \
        "autograd through an operator with a compute window (Geom.k_begin / k_end) is not supported"
    return need


def velocityDivergence(U, flags, *, geom=None):
    """lib/fluid/velocity_divergence.py:4-74 (differentiable w.r.t. U)"""
    _check5(U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    if _needs_grad(U, geom=geom):
        return _DivergenceFn.apply(U, flags, geom)
    return ext.velocity_divergence(U, flags, geom)


def velocityUpdate(pressure, U, flags, *, geom=None):
    """lib/fluid/velocity_update.py:6-162 -- in place on U, returns None (differentiable w.r.t. pressure and U)."""
    _check5(pressure, U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    assert pressure.shape == flags.shape, "size mismatch"
    if _needs_grad(pressure, U, geom=geom):
        _VelocityUpdateFn.apply(pressure, U, flags, geom)
        return
    ext.velocity_update_(pressure, U, flags, geom)


def addBuoyancy(U, flags, density, gravity, rho_star, dt, *, geom=None):
    """lib/fluid/source_terms.py:6-116 -- in place on U, returns U.  gravity: 3 floats (tensor or sequence)."""
    _check5(U, flags, density)
    assert flags.size(1) == 1, "flags is not scalar"
    g = gravity.detach().cpu().tolist() if torch.is_tensor(gravity) else [float(x) for x in gravity]
    assert len(g) == 3, "Gravity must be a 3D vector (even in 2D)"
    ext.add_buoyancy_(U, flags, density, g, float(rho_star), float(dt), geom)
    return U


def addGravity(U, flags, gravity, dt, *, geom=None):
    """lib/fluid/source_terms.py:122-219 -- in place on U, returns U."""
    _check5(U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    g = gravity.detach().cpu().tolist() if torch.is_tensor(gravity) else [float(x) for x in gravity]
    assert len(g) == 3, "Gravity must be a 3D vector (even in 2D)"
    ext.add_gravity_(U, flags, g, float(dt), geom)
    return U


def addViscosity(dt, U, flags, viscosity):
    """lib/fluid/viscosity.py:7-70 -- in place on U (2D only, like the reference)."""
    _check5(U, flags)
    assert flags.size(1) == 1, "flags is not scalar"
    assert U.size(1) == 2 and U.size(2) == 1, "addViscosity: ONLY IN 2D"
    ext.add_viscosity_(float(dt), U, flags, float(viscosity))


def setWallBcs(U, flags, *, geom=None):
    """lib/fluid/set_wall_bcs.py:4-86 -- in place on U, returns U (differentiable w.r.t. U)."""
    _check5(U, flags)
    assert flags.size(1) == 1, "flags is not a scalar"
    if _needs_grad(U, geom=geom):
        return _SetWallBcsFn.apply(U, flags, geom)
    ext.set_wall_bcs_(U, flags, geom)
    return U


def setWallBcsStick(U, flags, flags_stick):
    """lib/fluid/set_wall_bcs_stick.py:5-157 -- no-slip walls, in place on U (2D).  `flags_stick` is a copy of flags
    with the no-slip obstacle cells set to CellType.TypeStick (cylinder.py:76).  The reference function raises
    NameError as shipped; this is its body with the three unbound names bound (include/fluidnet_hip.h)."""
    _check5(U, flags)
    assert flags.dim() == 5 and flags_stick.dim() == 5, "Dimension mismatch"
    assert flags.size(1) == 1, "flags is not a scalar"
    assert flags_stick.size(1) == 1, "flags is not a scalar"
    assert flags_stick.is_contiguous(), "Input is not contiguous"
    ext.set_wall_bcs_stick_(U, flags, flags_stick)


def flagsToOccupancy(flags):
    """lib/fluid/flags_to_occupancy.py:6-19"""
    return ext.flags_to_occupancy(flags)


def setConstVals(batch_dict, p, U, flags, density):
    """lib/simulate.py:4-26 (same side effects on batch_dict: stores clones)."""
    has_u = ("UBCInvMask" in batch_dict) and ("UBC" in batch_dict)
    has_r = ("densityBCInvMask" in batch_dict) and ("densityBC" in batch_dict)
    if not (has_u or has_r):
        return
    ext.set_const_vals_(U, batch_dict["UBC"] if has_u else None, batch_dict["UBCInvMask"] if has_u else None,
                        density if has_r else None, batch_dict["densityBC"] if has_r else None,
                        batch_dict["densityBCInvMask"] if has_r else None)
    if has_u:
        batch_dict["U"] = U.clone()
    if has_r:
        batch_dict["density"] = density.clone()
