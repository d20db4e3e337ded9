"""One fluid time step -- drop-in for the reference's `lib.simulate` (pytorch/lib/simulate.py:28-171).

`simulate(mconf, batch_dict, net, sim_method)` keeps the reference's calling convention (in place on
batch_dict, same mconf keys).  Two execution modes:
  fused=True  (default): one call into the native `simulate_step_` (the whole step enqueued by C++, no
              Python between kernels);
  fused=False: operator by operator through the `fluid.*` surface in the reference's order (what the
              parity tests use to compare stage by stage).
The reference's own call -- `simulate(mconf, batch_dict, net, sim_method)`, four arguments (plume.py:237) -- gets the static
path by itself: the layer keeps one step workspace per (device, grid shape) and looks at `(data_ptr, _version, shape, stride)` of
`flags` and of the four BC arrays (torch bumps `_version` on every in-place write, and so do this package's own in-place
operators).  While they are the tensors of the previous call, unwritten, the step reuses what it derived from them (the 3D
solver's obstacle mask; the 1-byte class map of the BC stages, built on the second such call); any change -- a new tensor, an
in-place write, another shape -- drops back to deriving everything again for that call.  The cache holds a reference to those
five tensors, so their addresses cannot be recycled while it trusts them.  What torch's counter does not see: writes through
`.data`, a numpy / dlpack alias or a raw pointer -- after one of those call `forget_static_inputs()` (or pass `static_flags=0`).
`release_workspaces()` frees the cached workspaces (the CNN's is ~1 KB per cell; one per device, stream and grid shape, at most four).
A call captured in a HIP graph bakes the bits of capture time into the graph: replays do not look at the tensors again (and a graph
captured over a cached workspace must not be replayed after `release_workspaces()`; a FIRST call inside a capture keeps nothing).

Explicit control, as before: `workspace` (a uint8 tensor of ext.step_workspace_bytes) with `static_flags` = True / the C ABI's
bit set (FnxStepParams.static_flags): 1 = flags unchanged since the previous call on that workspace (the 3D Jacobi reuses its
obstacle mask), 2 = the four BC arrays unchanged (the BC stages then go by a 1-byte class map kept in the workspace), 4 = that
map was already built by an earlier call with bit 2 -- e.g. 0 for the first step, 3 for the second, 7 from the third on.
`static_flags=None` (the default) with no `workspace` is the automatic mode above; with a caller's workspace it means 0.
`geom` (an `ext.Geom`, 3D only) selects the reference-quirk mode / a z-slab view for this call; it is per call, the
extension keeps no state.
"""
import torch

from . import fluid
from ._ext import ext


def setConstVals(batch_dict, p, U, flags, density):
    fluid.setConstVals(batch_dict, p, U, flags, density)


def _gravity(mconf, scale):
    gv = mconf["gravityVec"]
    return [float(gv["x"]), float(gv["y"]), float(gv["z"])], float(scale)


_BC_KEYS = ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask")
_AUTO = {}            # (device index, stream, B, D, H, W, is3D) -> _AutoStep
_AUTO_MAX = 4         # grid shapes kept (least recently used goes first)


def _ident(t):
    """what identifies a tensor's CONTENT between two calls, or None when torch cannot say (inference tensors have no counter)"""
    if t is None:
        return ()
    try:
        return (t.data_ptr(), t._version, tuple(t.shape), t.stride())
    except RuntimeError:
        return None


class _AutoStep:
    """the step workspace of one grid shape + what it holds that was derived from flags / the BC arrays"""

    def __init__(self, nbytes, like):
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
        self.forget()

    def forget(self):
        self.flags_id = self.bc_id = None     # identities at the previous call
        self.mask_id = None                   # flags identity the kept 3D obstacle mask was built from
        self.cls_id = None                    # BC identity the kept class map was built from
        self.held = None                      # the five tensors themselves: alive -> their addresses cannot be reused

    def static_bits(self, batch_dict, flags, is3D, jacobi):
        fid = _ident(flags)
        bcs = [batch_dict.get(k) for k in _BC_KEYS]
        bid = None if any(_ident(t) is None for t in bcs) else tuple(_ident(t) for t in bcs)
        bits = 0
        if fid is not None and fid == self.flags_id and (not (is3D and jacobi) or self.mask_id == fid):
            bits |= 1                         # flags unchanged (and, where a mask is used, the kept one is theirs)
        if bid is not None and bid == self.bc_id:
            bits |= 2                         # BC arrays unchanged since the previous call: go by the class map ...
            if self.cls_id == bid:
                bits |= 4                     # ... which an earlier call already built
            self.cls_id = bid
        else:
            self.cls_id = None
        if is3D and jacobi:
            self.mask_id = fid                # this call leaves the mask of these flags behind (built now, or reused)
        self.flags_id, self.bc_id = fid, bid
        self.held = (flags, bcs)
        return bits


def _auto_step(flags, is3D):
    """the cached step state of this (device, stream, grid shape), or None where the layer must not keep one: a first call inside a
    HIP-graph capture (its allocation would belong to the graph's private pool and die with the graph)"""
    dev = flags.device.index if flags.device.index is not None else torch.cuda.current_device()
    # per stream: two simulations of one shape on two streams must not share scratch memory, and a workspace is handed back to the
    # allocator on the stream it was used on
    key = (dev, torch.cuda.current_stream(dev).cuda_stream) + tuple(int(flags.size(i)) for i in (0, 2, 3, 4)) + (bool(is3D),)
    a = _AUTO.pop(key, None)
    if a is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        while len(_AUTO) >= _AUTO_MAX:
            _AUTO.pop(next(iter(_AUTO)))
        a = _AutoStep(ext.step_workspace_bytes(key[2], key[3], key[4], key[5], key[6]), flags)
    _AUTO[key] = a                            # (re-inserted: most recently used last)
    return a


def forget_static_inputs():
    """the next simulate() call of every grid shape derives everything from flags / the BC arrays again (after a write torch's
    version counter cannot see: `.data`, a numpy / dlpack alias, a raw pointer)"""
    for a in _AUTO.values():
        a.forget()


def release_workspaces():
    """free the step workspaces the automatic mode keeps (one per grid shape seen, at most four)"""
    _AUTO.clear()


def simulate(mconf, batch_dict, net, sim_method, output_div=False, fused=True, workspace=None, static_flags=None,
             geom=None):
    assert sim_method in ("convnet", "jacobi"), "Simulation method not supported. Choose either convnet or jacobi."
    dt = float(mconf["dt"])
    maccormackStrength = mconf["maccormackStrength"]
    sampleOutsideFluid = mconf["sampleOutsideFluid"]
    buoyancyScale = mconf["buoyancyScale"]
    gravityScale = mconf.get("gravityScale", 0)
    viscosity = mconf.get("viscosity", 0)
    assert viscosity >= 0, "Viscosity must be positive"
    p, U, flags = batch_dict["p"], batch_dict["U"], batch_dict["flags"]
    has_density = "density" in batch_dict

    is3D = U.size(1) == 3
    # the optional stages (viscosity, gravityScale, correctScalar, the periodic patches, flags_stick) are branches of the
    # native step (FnxStepParams / FnxState); what stays on the operator path: output_div (a training-time early return),
    # the stages the reference only has in 2D asked for in 3D (they raise there, as the operators do), and
    # no 'density' key but a density BC: the reference applies setConstVals to its zeros (simulate.py:82-97)
    simple = (not output_div and not (is3D and (viscosity > 0 or "flags_stick" in batch_dict))
              and (has_density or not ("densityBC" in batch_dict and "densityBCInvMask" in batch_dict)))
    if fused and simple:
        periodic = 0
        if "periodic-x" in mconf and "periodic-y" in mconf:                     # simulate.py:121, :157
            periodic = 1 | (2 if mconf["periodic-x"] else 0) | (4 if mconf["periodic-y"] else 0)
        want_gvec = has_density and (buoyancyScale > 0 or gravityScale > 0)
        # gravityVec is only read when buoyancy is applied (simulate.py:99-105)
        gvec = _gravity(mconf, 1.0)[0] if want_gvec else [0.0, 0.0, 0.0]
        density = batch_dict["density"] if has_density else None
        packed = net.packed_for(U.device) if (sim_method == "convnet") else None
        if workspace is None and static_flags is None and flags.is_cuda and geom is None:
            # the reference's four-argument call: the layer's own workspace, static inputs detected (module docstring)
            auto = _auto_step(flags, is3D)
            if auto is not None:
                workspace, static_flags = auto.workspace, auto.static_bits(batch_dict, flags, is3D, sim_method == "jacobi")
        if static_flags is None:
            static_flags = 0
        ext.simulate_step_(p, U, flags, density, batch_dict.get("UBC"), batch_dict.get("UBCInvMask"),
                           batch_dict.get("densityBC"), batch_dict.get("densityBCInvMask"), packed, dt,
                           float(maccormackStrength), bool(sampleOutsideFluid), float(buoyancyScale), gvec,
                           float(mconf.get("operatingDensity", 0.0)), float(mconf.get("pTol", 0.0)),
                           int(mconf.get("jacobiIter", 1)), sim_method,
                           float(mconf.get("normalizeInputThreshold", 1e-5)), workspace, int(static_flags), geom,
                           getattr(net, "precision_mode", "fp32") if sim_method == "convnet" else "fp32",
                           float(viscosity), float(gravityScale) if gravityScale > 0 else 0.0,
                           bool(mconf.get("correctScalar", False)) and has_density, periodic,
                           batch_dict.get("flags_stick") if sim_method == "convnet" else None)
        if not has_density:
            batch_dict["density"] = torch.zeros_like(flags)     # simulate.py:82-83
        return

    stick = "flags_stick" in batch_dict                          # simulate.py:61-64
    # ---- operator-by-operator path, reference order ----
    orig = U
    if viscosity > 0:                                           # simulate.py:66-69
        orig = U.clone()
        fluid.addViscosity(dt, orig, flags, viscosity)
    if has_density:
        density = fluid.advectScalar(dt, batch_dict["density"], U, flags, method="maccormackFluidNet",
                                     boundary_width=1, sample_outside_fluid=sampleOutsideFluid,
                                     maccormack_strength=maccormackStrength, geom=geom)
        if mconf.get("correctScalar", False):
            div = fluid.velocityDivergence(U, flags, geom=geom)
            fluid.correctScalar(dt, density, div, flags)
    else:
        density = torch.zeros_like(flags)
    U = fluid.advectVelocity(dt=dt, orig=orig, U=U, flags=flags, method="maccormackFluidNet", boundary_width=1,
                             maccormack_strength=maccormackStrength, geom=geom)
    setConstVals(batch_dict, p, U, flags, density)
    if has_density and buoyancyScale > 0:
        gvec, _ = _gravity(mconf, 1.0)
        gravity = (torch.tensor(gvec, dtype=torch.float32) * (-buoyancyScale)).tolist()
        U = fluid.addBuoyancy(U, flags, density, gravity, mconf["operatingDensity"], dt, geom=geom)
    if has_density and gravityScale > 0:                        # simulate.py:107-114 (inside the density branch)
        gvec, _ = _gravity(mconf, 1.0)
        gravity = (torch.tensor(gvec, dtype=torch.float32) * (-gravityScale)).tolist()
        U = fluid.addGravity(U, flags, gravity, dt, geom=geom)
    if output_div:
        return
    periodic = "periodic-x" in mconf and "periodic-y" in mconf

    def wall_bcs(U):
        if periodic:
            U_temp = U.clone()
        U = fluid.setWallBcs(U, flags, geom=geom)
        if periodic:
            if mconf["periodic-x"]:
                U[:, 1, :, :, 1] = U_temp[:, 1, :, :, U.size(4) - 1]
            if mconf["periodic-y"]:
                U[:, 0, :, 1] = U_temp[:, 0, :, U.size(3) - 1]
        return U

    if sim_method != "convnet":
        U = wall_bcs(U)
    elif stick:                                                  # simulate.py:129-130
        fluid.setWallBcsStick(U, flags, batch_dict["flags_stick"])
    setConstVals(batch_dict, p, U, flags, density)
    if sim_method == "convnet":
        data = torch.cat((p, U, flags, density), 1)
        p, U = net(data)
        if stick:                                                # simulate.py:165-166
            fluid.setWallBcsStick(U, flags, batch_dict["flags_stick"])
    else:
        div = fluid.velocityDivergence(U, flags, geom=geom)
        is3D = U.size(2) > 1
        p, residual = fluid.solveLinearSystemJacobi(flags=flags, div=div, is_3d=is3D, p_tol=mconf["pTol"],
                                                    max_iter=mconf["jacobiIter"], geom=geom)
        fluid.velocityUpdate(pressure=p, U=U, flags=flags, geom=geom)
        U = wall_bcs(U)
    setConstVals(batch_dict, p, U, flags, density)
    batch_dict["U"] = U
    batch_dict["density"] = density
    batch_dict["p"] = p
