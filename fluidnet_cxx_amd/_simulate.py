"""One fluid time step -- drop-in for the reference's `lib.simulate` (pytorch/lib/simulate.py:28-171).

`simulate(mconf, batch_dict, net, sim_method)` keeps the reference's calling convention (in place on
batch_dict, same mconf keys).  Two execution modes:
  fused=True  (default): one call into the native `simulate_step_` (the whole step enqueued by C++, no
              Python between kernels);
  fused=False: operator by operator through the `fluid.*` surface in the reference's order (what the
              parity tests use to compare stage by stage).
`workspace` (a uint8 tensor of ext.step_workspace_bytes) avoids a per-step allocation; with it, `static_flags=True`
promises that batch_dict['flags'] has not changed since the previous call on that workspace (3D Jacobi then reuses
its obstacle mask).  `static_flags` may also be the C ABI's bit set (FnxStepParams.static_flags): 1 = flags unchanged,
2 = the four BC arrays unchanged (the BC stages then go by a 1-byte class map kept in the workspace), 4 = that map was
already built by an earlier call with bit 2 -- e.g. 0 for the first step, 3 for the second, 7 from the third on.
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


def simulate(mconf, batch_dict, net, sim_method, output_div=False, fused=True, workspace=None, static_flags=False,
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
