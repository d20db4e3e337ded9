#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference, no GPU).  It
  1. copies the reference's 4 C++ sources of `fluidnet_cpp` to a scratch dir under /tmp, applies
     three mechanical API renames so they compile against torch 2.10 (no arithmetic change):
       max_values -> amax, at::kByte -> at::kBool, `1 - mask` -> mask.logical_not()
     and JIT-builds the extension there (nothing from the reference enters this repo);
  2. imports the reference's Python package `lib` unmodified, with three harness-side shims
     (torch.device -> cpu, Tensor.cuda -> identity, torch.uint8 -> torch.bool);
  3. runs every hot-path operator on seeded inputs and stores inputs + outputs as .npz.

The fixtures are data only (inputs and expected outputs).  Usage:
    python tools/make_golden.py [--only ops2d,ops3d,plume,sim,cnn,gen,stick,grid,dump,grad]
"""
import argparse
import importlib.util
import os
import shutil
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/oracle"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def build_reference_extension():
    src = os.path.join(SCRATCH, "src")
    bld = os.path.join(SCRATCH, "build")
    os.makedirs(src, exist_ok=True)
    os.makedirs(bld, exist_ok=True)
    if not os.path.exists(os.path.join(bld, "fluidnet_cpp.so")):
        cpp = os.path.join(REF, "pytorch/lib/fluid/cpp")
        for f in os.listdir(cpp):
            if f.endswith((".cpp", ".h")):
                shutil.copy(os.path.join(cpp, f), src)
        seds = [
            ("s/maxT.max_values(1, true)/maxT.amax(1, true)/", ["calc_line_trace.cpp"]),
            ("s/at::kByte/at::kBool/g", ["calc_line_trace.cpp", "fluids_init.cpp", "grid.cpp", "advect_type.cpp"]),
            ("s/maskSolid.equal(1-maskFluid)/maskSolid.equal(maskFluid.logical_not())/", ["fluids_init.cpp"]),
            ("s/T m3 = 1 - (m0.__or__(m1).__or__(m2));/T m3 = (m0.__or__(m1).__or__(m2)).logical_not();/",
             ["grid.cpp"]),
        ]
        for expr, files in seds:
            for f in files:
                subprocess.check_call(["sed", "-i", expr, os.path.join(src, f)])
    import torch.utils.cpp_extension as ce
    ce.load(name="fluidnet_cpp", build_directory=bld, with_cuda=False, verbose=False,
            sources=[os.path.join(src, f) for f in
                     ["grid.cpp", "advect_type.cpp", "calc_line_trace.cpp", "fluids_init.cpp"]])
    return bld


def import_reference():
    import torch
    import torch._dynamo  # noqa: F401  (pre-load lazies before torch.device is shimmed)
    sys.dont_write_bytecode = True
    bld = build_reference_extension()
    real_device = torch.device

    class _Meta(type):
        def __instancecheck__(cls, inst):
            return isinstance(inst, real_device)

    class _DevShim(metaclass=_Meta):
        def __new__(cls, *a, **k):
            return real_device("cpu")

    torch.device = _DevShim
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.uint8 = torch.bool
    sys.path[:0] = [bld, os.path.join(REF, "pytorch")]
    import fluidnet_cpp  # noqa: F401
    import lib
    return torch, lib, fluidnet_cpp


# ---------------------------------------------------------------------------------------------
def make_flags(rng, B, D, H, W, boxes=True, empties=False):
    f = np.full((B, 1, D, H, W), 1.0, np.float32)
    f[:, :, :, 0, :] = 2; f[:, :, :, -1, :] = 2; f[:, :, :, :, 0] = 2; f[:, :, :, :, -1] = 2
    if D > 1:
        f[:, :, 0] = 2; f[:, :, -1] = 2
    if boxes:
        zs = slice(None) if D == 1 else slice(D // 3, D // 3 + 3)
        f[:, :, zs, H // 3:H // 3 + 4, W // 4:W // 4 + 5] = 2          # box
        f[:, :, zs if D == 1 else slice(D // 2, D // 2 + 1), 2 * H // 3, 2 * W // 3] = 2   # single cell
        if D == 1:
            f[:, :, :, H // 2, W // 2:W // 2 + 7] = 2                   # bar
    if empties:
        f[:, :, :, 3, 3] = 4; f[:, :, :, 3, 4] = 4; f[:, :, :, H - 4, W - 5] = 4
    return f


def t(x, torch):
    return torch.from_numpy(np.ascontiguousarray(x))


def gen_ops(torch, lib, ext, name, B, D, H, W, sigma, dt, seed, empties=False, boxes=True, jac_iters=7):
    fluid = lib.fluid
    rng = np.random.default_rng(seed)
    is3d = D > 1
    nc = 3 if is3d else 2
    flags = make_flags(rng, B, D, H, W, boxes, empties)
    U = (rng.standard_normal((B, nc, D, H, W)) * sigma).astype(np.float32)
    rho = rng.random((B, 1, D, H, W)).astype(np.float32)
    p = rng.standard_normal((B, 1, D, H, W)).astype(np.float32)
    out = dict(flags=flags, U=U, rho=rho, p=p, dt=np.float32(dt), is3d=np.int32(is3d))
    tf, tU, trho, tp = t(flags, torch), t(U, torch), t(rho, torch), t(p, torch)

    for meth in ("maccormackFluidNet", "eulerFluidNet"):
        for so in (False, True):
            try:
                r = ext.advect_scalar(float(dt), trho, tU, tf, meth, 1, so, 0.6)
                out[f"advect_scalar_{meth}_{int(so)}"] = r.numpy().copy()
            except RuntimeError as e:       # reference aborts on its own rare corner cases (Q8)
                print(f"  [{name}] advect_scalar {meth} so={so} raised: {str(e).splitlines()[0]}")
        r = ext.advect_vel(float(dt), tU, tU, tf, meth, 1, 0.6)
        out[f"advect_vel_{meth}"] = r.numpy().copy()
    # advect a different field than the carrier
    orig = (rng.standard_normal((B, nc, D, H, W))).astype(np.float32)
    out["orig"] = orig
    out["advect_vel_orig"] = ext.advect_vel(float(dt), t(orig, torch), tU, tf, "maccormackFluidNet", 1, 0.75).numpy().copy()

    div = fluid.velocityDivergence(tU.clone(), tf)
    out["divergence"] = div.numpy().copy()
    # NOTE (reference defect, B>1 only): solve_linear_system indexes with idx_b (B,D,H,W) next to a
    # (B,1,D,H,W) zero index (fluids_init.cpp:870-901), which broadcasts to (B,B,D,H,W) and mixes samples;
    # its B>1 output is garbage.  The per-sample semantics are pinned by calling it sample by sample.
    def jac(tol, iters):
        ps, rs = [], []
        for b in range(B):
            pb, rb = ext.solve_linear_system(tf[b:b + 1].contiguous(), div[b:b + 1].contiguous(), is3d, tol, iters, False)
            ps.append(pb); rs.append(rb.item())
        return torch.cat(ps, 0), max(rs), rs
    pj, res, _ = jac(0.0, jac_iters)
    out["jacobi_p"] = pj.numpy().copy(); out["jacobi_res"] = np.float32(res); out["jacobi_iters"] = np.int32(jac_iters)
    pj1, res1, _ = jac(0.0, 1)
    out["jacobi1_p"] = pj1.numpy().copy(); out["jacobi1_res"] = np.float32(res1)
    if B == 1:
        # early exit on tolerance: pick a tolerance between the residual at sweep 3 and 4
        _, r3, _ = jac(0.0, 3)
        _, r4, _ = jac(0.0, 4)
        tol = 0.5 * (r3 + r4)
        pt, rt, _ = jac(tol, 50)
        out["jacobi_tol"] = np.float32(tol); out["jacobi_tol_p"] = pt.numpy().copy(); out["jacobi_tol_res"] = np.float32(rt)
    g = torch.tensor([0.3, 0.25, -0.2], dtype=torch.float32)
    out["gravity"] = g.numpy().copy(); out["rho_star"] = np.float32(0.05)
    if not is3d:
        Uu = tU.clone(); fluid.velocityUpdate(tp, Uu, tf); out["velocity_update"] = Uu.numpy().copy()
        Uw = tU.clone(); fluid.setWallBcs(Uw, tf); out["set_wall_bcs"] = Uw.numpy().copy()
    Ub = tU.clone(); fluid.addBuoyancy(Ub, tf, trho, g, 0.05, float(dt)); out["add_buoyancy"] = Ub.numpy().copy()
    Ug = tU.clone(); fluid.addGravity(Ug, tf, g, float(dt)); out["add_gravity"] = Ug.numpy().copy()
    if not is3d:
        Uv = tU.clone(); fluid.addViscosity(float(dt), Uv, tf, 0.07); out["add_viscosity"] = Uv.numpy().copy()
        out["viscosity"] = np.float32(0.07)
    out["occupancy"] = fluid.flagsToOccupancy(tf).numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"ops_{name}.npz"), **out)
    print(f"  wrote ops_{name}.npz")


def plume_setup(torch, lib, res, B=1):
    fluid = lib.fluid
    p = torch.zeros((1, 1, 1, res, res)); U = torch.zeros((1, 2, 1, res, res))
    flags = torch.zeros((1, 1, 1, res, res)); density = torch.zeros((1, 1, 1, res, res))
    fluid.emptyDomain(flags)
    bd = dict(p=p, U=U, flags=flags, density=density)
    fluid.createPlumeBCs(bd, 0.1, 2, 0.145)
    return bd


def plume_mconf(torch, jacobi_iter=28):
    import yaml
    mconf = torch.load(os.path.join(REF, "trained_models/ScaleNet_ShortTerm_LongTermLoss/convModel_mconf.pth"),
                       weights_only=False)
    with open(os.path.join(REF, "pytorch/plumeConfig.yaml")) as f:
        mconf.update(yaml.safe_load(f))
    mconf["jacobiIter"] = jacobi_iter
    mconf["pTol"] = 0.0
    return mconf


def load_net(torch, lib, mconf, seed=0):
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    path = os.path.join(REF, "trained_models/ScaleNet_ShortTerm_LongTermLoss/ScaleNet_ShortTerm_LongTermLoss_saved.py")
    spec = importlib.util.spec_from_file_location("model_saved", path)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    net = mod.FluidNet(mconf, dropout=False)
    w = make_scalenet_weights(seed)
    sd = net.state_dict()
    for k, v in w.items():
        assert sd[k].shape == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))
    net.eval()
    return net


def gen_plume(torch, lib, ext):
    """Developed 128x128 plume (60 Jacobi-28 steps): op-level vectors on a physical state + step states."""
    fluid = lib.fluid
    mconf = plume_mconf(torch)
    bd = plume_setup(torch, lib, 128)
    states = {}
    with torch.no_grad():
        for it in range(1, 61):
            lib.simulate(mconf, bd, None, "jacobi")
            if it in (1, 5, 20, 60):
                for k in ("U", "density", "p"):
                    states[f"{k}_{it}"] = bd[k].numpy().copy()
    out = dict(states)
    for k in ("flags", "UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
        out[k] = bd[k].numpy().copy()
    # op-level outputs on the developed state
    U, rho, flags = bd["U"].clone(), bd["density"].clone(), bd["flags"]
    out["advect_scalar"] = ext.advect_scalar(0.1, rho, U, flags, "maccormackFluidNet", 1, False, 0.6).numpy().copy()
    out["advect_vel"] = ext.advect_vel(0.1, U, U, flags, "maccormackFluidNet", 1, 0.6).numpy().copy()
    div = fluid.velocityDivergence(U, flags); out["divergence"] = div.numpy().copy()
    pj, res = ext.solve_linear_system(flags, div, False, 0.0, 28, False)
    out["jacobi28_p"] = pj.numpy().copy(); out["jacobi28_res"] = np.float32(res.item())
    np.savez_compressed(os.path.join(OUT, "plume128.npz"), **out)
    print("  wrote plume128.npz")


F2_CFG = {"viscosity": 0.02, "gravityScale": 0.5, "correctScalar": True, "periodic-x": True, "periodic-y": True}


def gen_sim_small(torch, lib, ext):
    """64x64 plume, Jacobi-28 and convnet (hash-seeded weights), states after 1, 3, 10 steps."""
    mconf = plume_mconf(torch)
    out = {}
    with torch.no_grad():
        for method in ("jacobi", "convnet"):
            bd = plume_setup(torch, lib, 64)
            net = load_net(torch, lib, mconf) if method == "convnet" else None
            for it in range(1, 11):
                lib.simulate(mconf, bd, net, method)
                if it in (1, 3, 10):
                    for k in ("U", "density", "p"):
                        out[f"{method}_{k}_{it}"] = bd[k].numpy().copy()
            if method == "jacobi":
                for k in ("flags", "UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
                    out[k] = bd[k].numpy().copy()
        # every optional stage of lib.simulate switched on (all off in the shipped configs): viscosity, correctScalar,
        # gravity, periodic patches
        m2 = dict(mconf); m2.update(F2_CFG)
        bd = plume_setup(torch, lib, 64)
        for it in range(1, 7):
            lib.simulate(m2, bd, None, "jacobi")
            if it in (1, 3, 6):
                for k in ("U", "density", "p"):
                    out[f"f2_{k}_{it}"] = bd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "sim64.npz"), **out)
    print("  wrote sim64.npz")


def gen_cnn(torch, lib, ext):
    """MultiScaleNet and FluidNet.forward on a 32x48 field with hash-seeded weights."""
    mconf = plume_mconf(torch)
    net = load_net(torch, lib, mconf)
    rng = np.random.default_rng(7)
    B, H, W = 2, 32, 48
    flags = make_flags(rng, B, 1, H, W)
    x = rng.standard_normal((B, 2, H, W)).astype(np.float32)
    x[:, 1] = (flags[:, 0, 0] == 2)
    out = dict(x=x, flags=flags)
    with torch.no_grad():
        out["multiscale"] = net.multiScale(t(x, torch)).numpy().copy()
        # intermediate tower outputs (for debugging the conv stack)
        import torch.nn.functional as F
        xt = t(x, torch)
        q = F.interpolate(xt, (H // 4, W // 4), mode="bilinear", align_corners=False)
        out["x_quarter"] = q.numpy().copy()
        c4 = net.multiScale.convN_4(q); out["c4"] = c4.numpy().copy()
        h = F.interpolate(xt, (H // 2, W // 2), mode="bilinear", align_corners=False)
        out["x_half"] = h.numpy().copy()
        out["c4_up"] = F.interpolate(c4, (H // 2, W // 2), mode="bilinear", align_corners=False).numpy().copy()
        U = (rng.standard_normal((B, 2, 1, H, W)) * 0.5).astype(np.float32)
        p = np.zeros((B, 1, 1, H, W), np.float32)
        rho = rng.random((B, 1, 1, H, W)).astype(np.float32)
        inp = np.concatenate([p, U, flags, rho], 1)
        out["fluidnet_in"] = inp
        pp, UU = net(t(inp, torch))
        out["fluidnet_p"] = pp.numpy().copy(); out["fluidnet_U"] = UU.numpy().copy()
        out["scale"] = net.scale(t(U, torch)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "cnn.npz"), **out)
    print("  wrote cnn.npz")


def gen_generators(torch, lib, ext):
    """emptyDomain / createPlumeBCs / createRayleighTaylorBCs outputs (flags bit-exact)."""
    fluid = lib.fluid
    out = {}
    for res in (16, 128):
        bd = plume_setup(torch, lib, res)
        for k in ("flags", "UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
            out[f"plume{res}_{k}"] = bd[k].numpy().copy()
    f3 = torch.zeros((2, 1, 6, 7, 8)); fluid.emptyDomain(f3); out["empty3d"] = f3.numpy().copy()
    bd = dict(p=torch.zeros(1, 1, 1, 40, 32), U=torch.zeros(1, 2, 1, 40, 32), flags=torch.zeros(1, 1, 1, 40, 32),
              density=torch.zeros(1, 1, 1, 40, 32))
    fluid.emptyDomain(bd["flags"])
    fluid.createRayleighTaylorBCs(bd, dict(perturbThickness=100, perturbAmplitude=0.01, height=0.5), -0.01, 0.01)
    out["rt_density"] = bd["density"].numpy().astype(np.float32)
    # createCylinder (geometry_utils.py:4-34) on a 2D and a 3D grid
    for tag, shape in (("cyl2d", (1, 1, 1, 40, 32)), ("cyl3d", (1, 1, 5, 24, 28))):
        cd = dict(flags=torch.zeros(shape))
        fluid.emptyDomain(cd["flags"])
        lib.fluid.geometry_utils.createCylinder(cd, 15.5, 20.0, 6.3) if hasattr(lib.fluid, "geometry_utils") else None
        out[tag + "_flags"] = cd["flags"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "generators.npz"), **out)
    print("  wrote generators.npz")


def gen_grid(torch, lib, ext):
    """getCentered (grid.py:7-32) on seeded MAC velocities, 2D (B=2) and 3D."""
    out = {}
    rng = np.random.default_rng(11)
    for tag, shape in (("2d", (2, 2, 1, 12, 17)), ("3d", (1, 3, 5, 6, 9))):
        U = rng.standard_normal(shape).astype(np.float32)
        out[f"U_{tag}"] = U
        out[f"centered_{tag}"] = lib.fluid.getCentered(torch.from_numpy(U.copy())).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "grid.npz"), **out)
    print("  wrote grid.npz")


def gen_dump(torch, lib, ext):
    """The arrays the reference driver hands to pyevtk.gridToVTK (plume.py:317-408) for a seeded 2D state with
    obstacles, on a window smaller than the domain.  The driver has no function for this: the statements are inline in
    its main loop, so they are executed here from the reference file itself (by line range) in a namespace that holds
    the state."""
    import textwrap
    import numpy.ma as ma
    rng = np.random.default_rng(21)
    H, W = 24, 38
    flags = make_flags(rng, 1, 1, H, W, boxes=True)
    bd = dict(U=rng.standard_normal((1, 2, 1, H, W)).astype(np.float32), p=rng.standard_normal((1, 1, 1, H, W)).astype(np.float32),
              density=rng.random((1, 1, 1, H, W)).astype(np.float32), flags=flags)
    out = {f"in_{k}": v for k, v in bd.items()}
    win = dict(minX=2, maxX=35, minY=1, maxY=20, maxX_win=33, maxY_win=19)
    out["window"] = np.array([win["minX"], win["maxX"], win["minY"], win["maxY"]], np.int32)
    lines = open(os.path.join(REF, "pytorch/plume.py")).read().splitlines()
    ns = dict(torch=torch, np=np, ma=ma, fluid=lib.fluid, batch_dict={k: torch.from_numpy(v.copy()) for k, v in bd.items()}, **win)
    exec(textwrap.dedent("\n".join(lines[316:329])), ns)      # plume.py:317-329  nx, ny, dx, dy, x, y, z
    exec(textwrap.dedent("\n".join(lines[330:408])), ns)      # plume.py:331-408  the cell arrays
    for k in ("x", "y", "z", "divergence", "rho", "p", "velx", "vely", "gradRhox", "gradRhoy", "gradPx", "gradPy"):
        out[f"vtk_{k}"] = np.asarray(ns[k]).copy()
    np.savez_compressed(os.path.join(OUT, "dump.npz"), **out)
    print("  wrote dump.npz")


def gen_grad(torch, lib, ext):
    """Gradients of the stencil operators the training graph differentiates through (model.py:190-227 and
    fluid_net_train.py:366: velocityUpdate -> setWallBcs -> velocityDivergence), from the reference's own autograd:
    loss = sum(w_div * div) + sum(w_U * U), d loss / d p and d loss / d U_in, 2D with obstacles and Empty cells."""
    fluid = lib.fluid
    out = {}
    rng = np.random.default_rng(31)
    for tag, (B, H, W, empties) in (("a", (2, 20, 33, False)), ("b", (1, 24, 40, True))):
        flags = make_flags(rng, B, 1, H, W, boxes=True, empties=empties)
        U0 = rng.standard_normal((B, 2, 1, H, W)).astype(np.float32)
        p0 = rng.standard_normal((B, 1, 1, H, W)).astype(np.float32)
        wd = rng.standard_normal((B, 1, 1, H, W)).astype(np.float32)
        wu = rng.standard_normal((B, 2, 1, H, W)).astype(np.float32)
        tf = t(flags, torch)
        tU0 = t(U0, torch).requires_grad_(True)
        tp = t(p0, torch).requires_grad_(True)
        U = tU0 * 1.0                                   # (the operators work in place: not on a leaf)
        fluid.velocityUpdate(pressure=tp, U=U, flags=tf)
        U = fluid.setWallBcs(U, tf)
        div = fluid.velocityDivergence(U.contiguous(), tf)
        loss = (div * t(wd, torch)).sum() + (U * t(wu, torch)).sum()
        loss.backward()
        out.update({f"{tag}_flags": flags, f"{tag}_U": U0, f"{tag}_p": p0, f"{tag}_wd": wd, f"{tag}_wu": wu,
                    f"{tag}_U_out": U.detach().numpy().copy(), f"{tag}_div": div.detach().numpy().copy(),
                    f"{tag}_grad_U": tU0.grad.numpy().copy(), f"{tag}_grad_p": tp.grad.numpy().copy()})
    np.savez_compressed(os.path.join(OUT, "grad.npz"), **out)
    print("  wrote grad.npz")


def stick_flags(flags):
    """flags_stick as cylinder.py:76 builds it: a copy of flags with the no-slip cells set to TypeStick (128).  Marked
    here: the obstacle box (thick), the single obstacle cell, the 1-cell bar (fluid on both sides) and a stretch of the
    bottom and left domain walls (corner included)."""
    B, _, D, H, W = flags.shape
    fs = flags.copy()
    fs[:, :, :, H // 3:H // 3 + 4, W // 4:W // 4 + 5] = 128
    fs[:, :, :, 2 * H // 3, 2 * W // 3] = 128
    fs[:, :, :, H // 2, W // 2:W // 2 + 7] = 128
    fs[:, :, :, 0, 0:W // 2] = 128
    fs[:, :, :, 0:H // 2, 0] = 128
    fs[flags != 2] = flags[flags != 2]           # stick cells are obstacles (set_wall_bcs_stick.py:50-51)
    return fs


def gen_stick(torch, lib, ext):
    """setWallBcsStick (set_wall_bcs_stick.py:5-157).  As shipped it raises NameError on the bare names TypeObstacle /
    TypeFluid / TypeStick (:62 ff.); the harness binds those three names in the module's namespace (no source edit) --
    the 2D body then runs as written, including its own quirks (the 'both neighbours' test of the horizontal component
    checks the lower neighbour twice, :131; the corner sums count cur and left twice, :146-152).  The 3D branch has
    further typos (:85-86) and no z handling in the no-slip part: 2D only."""
    fluid = lib.fluid
    mod = sys.modules[fluid.setWallBcsStick.__module__]
    ct = fluid.CellType
    mod.TypeObstacle, mod.TypeFluid, mod.TypeStick = ct.TypeObstacle, ct.TypeFluid, ct.TypeStick
    out = {}
    for name, B, H, W, sigma, seed, empties in (("a", 2, 20, 33, 2.0, 11, False), ("b", 1, 24, 40, 5.0, 12, True)):
        rng = np.random.default_rng(seed)
        flags = make_flags(rng, B, 1, H, W, True, empties)
        fs = stick_flags(flags)
        U = (rng.standard_normal((B, 2, 1, H, W)) * sigma).astype(np.float32)
        tU = t(U, torch).clone()
        fluid.setWallBcsStick(tU, t(flags, torch), t(fs, torch))
        out.update({f"{name}_flags": flags, f"{name}_flags_stick": fs, f"{name}_U": U, f"{name}_out": tU.numpy().copy()})
    # lib.simulate with 'flags_stick' in the batch (convnet method: simulate.py:129-130,165-166), 64x64 plume around a
    # no-slip cylinder, hash-seeded weights, 3 steps
    mconf = plume_mconf(torch)
    with torch.no_grad():
        bd = plume_setup(torch, lib, 64)
        fluid.createCylinder(bd, 32, 30, 6)
        fsk = bd["flags"].clone()
        fsk[(bd["flags"] == 2) & (torch.arange(64).view(1, 1, 1, 64, 1) > 5) & (torch.arange(64).view(1, 1, 1, 64, 1) < 58)
            & (torch.arange(64).view(1, 1, 1, 1, 64) > 5) & (torch.arange(64).view(1, 1, 1, 1, 64) < 58)] = 128
        bd["flags_stick"] = fsk
        out["sim_flags"] = bd["flags"].numpy().copy(); out["sim_flags_stick"] = fsk.numpy().copy()
        net = load_net(torch, lib, mconf)
        for it in range(1, 4):
            lib.simulate(mconf, bd, net, "convnet")
            for k in ("U", "density", "p"):
                out[f"sim_{k}_{it}"] = bd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "stick.npz"), **out)
    print("  wrote stick.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="ops2d,ops3d,plume,sim,cnn,gen,stick,grid,dump,grad")
    a = ap.parse_args()
    only = set(a.only.split(","))
    os.makedirs(OUT, exist_ok=True)
    torch, lib, ext = import_reference()
    torch.set_num_threads(8)
    if "ops2d" in only:
        gen_ops(torch, lib, ext, "2d_a", B=2, D=1, H=20, W=33, sigma=2.0, dt=0.1, seed=1)          # CFL ~0.2-0.8
        gen_ops(torch, lib, ext, "2d_b", B=1, D=1, H=24, W=40, sigma=12.0, dt=0.25, seed=2, empties=True)  # CFL ~3+
        gen_ops(torch, lib, ext, "2d_c", B=1, D=1, H=16, W=16, sigma=1.0, dt=0.1, seed=3, boxes=False)
        gen_ops(torch, lib, ext, "2d_d", B=3, D=1, H=48, W=64, sigma=6.0, dt=0.2, seed=4)          # CFL ~1-4
    if "ops3d" in only:
        gen_ops(torch, lib, ext, "3d_a", B=1, D=9, H=10, W=12, sigma=2.0, dt=0.1, seed=5, jac_iters=3)
        gen_ops(torch, lib, ext, "3d_b", B=2, D=8, H=12, W=10, sigma=8.0, dt=0.25, seed=6, jac_iters=3)
    if "plume" in only:
        gen_plume(torch, lib, ext)
    if "sim" in only:
        gen_sim_small(torch, lib, ext)
    if "cnn" in only:
        gen_cnn(torch, lib, ext)
    if "gen" in only:
        gen_generators(torch, lib, ext)
    if "stick" in only:
        gen_stick(torch, lib, ext)
    if "grid" in only:
        gen_grid(torch, lib, ext)
    if "dump" in only:
        gen_dump(torch, lib, ext)
    if "grad" in only:
        gen_grad(torch, lib, ext)


if __name__ == "__main__":
    main()
