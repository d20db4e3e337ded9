"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module).

Every function takes/returns numpy float32 arrays shaped (B,C,D,H,W) like the reference operators.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfluid_oracle.so")


class OraGrid(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("D", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("is3D", ctypes.c_int), ("zoff", ctypes.c_int), ("Dglob", ctypes.c_int)]


# z-slab view used by the multi-GPU decomposition tests: arrays hold planes [zoff, zoff+D) of a Dglob-deep domain
_SLAB = [0, 0]


def set_slab(zoff=0, dglob=0):
    _SLAB[0], _SLAB[1] = int(zoff), int(dglob)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("fluid_oracle.c", "cnn_oracle.c")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfluid_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        _lib.ora_scalenet_weight_floats.restype = ctypes.c_size_t
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _grid(flags_like, is3d=None):
    B, _, D, H, W = flags_like.shape
    if is3d is None:
        is3d = D > 1
    return OraGrid(B, D, H, W, int(is3d), _SLAB[0] if is3d else 0, _SLAB[1] if is3d else 0)


METHODS = {"eulerFluidNet": 0, "maccormackFluidNet": 1}


def advect_scalar(dt, src, U, flags, method="maccormackFluidNet", bnd=1, sample_outside=False, strength=0.75,
                  quirks=False):
    g = _grid(flags, U.shape[1] == 3)
    src, ps = _f(src); U, pu = _f(U); flags, pf = _f(flags)
    dst = np.empty_like(src)
    rc = lib().ora_advect_scalar(ctypes.byref(g), ctypes.c_float(dt), ps, pu, pf,
                                 dst.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), METHODS[method], int(bnd),
                                 int(sample_outside), ctypes.c_float(strength), int(quirks))
    assert rc == 0, rc
    return dst


def advect_vel(dt, orig, U, flags, method="maccormackFluidNet", bnd=1, strength=0.75, quirks=False):
    g = _grid(flags, U.shape[1] == 3)
    orig, po = _f(orig); U, pu = _f(U); flags, pf = _f(flags)
    dst = np.empty_like(U)
    rc = lib().ora_advect_vel(ctypes.byref(g), ctypes.c_float(dt), po, pu, pf,
                              dst.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), METHODS[method], int(bnd),
                              ctypes.c_float(strength), int(quirks))
    assert rc == 0, rc
    return dst


def velocity_divergence(U, flags):
    g = _grid(flags, U.shape[1] == 3)
    U, pu = _f(U); flags, pf = _f(flags)
    div = np.empty_like(flags)
    lib().ora_velocity_divergence(ctypes.byref(g), pu, pf, div.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return div


def jacobi(flags, div, is3d, p_tol, max_iter, quirks=False):
    g = _grid(flags, is3d)
    flags, pf = _f(flags); div, pd = _f(div)
    p = np.zeros_like(flags)
    res = ctypes.c_float(0); it = ctypes.c_int(0)
    rc = lib().ora_jacobi(ctypes.byref(g), pf, pd, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                          ctypes.byref(res), ctypes.c_float(p_tol), int(max_iter), int(quirks), ctypes.byref(it))
    assert rc == 0, rc
    return p, res.value, it.value


def jacobi_sweeps(flags, div, p, is3d, nsweeps, quirks=False):
    g = _grid(flags, is3d)
    flags, pf = _f(flags); div, pd = _f(div)
    p = np.array(p, dtype=np.float32, order="C", copy=True)
    lib().ora_jacobi_sweeps(ctypes.byref(g), pf, pd, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(nsweeps), int(quirks))
    return p


def velocity_update(p, U, flags):
    g = _grid(flags, U.shape[1] == 3)
    p, pp = _f(p); flags, pf = _f(flags)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    lib().ora_velocity_update(ctypes.byref(g), pp, U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf)
    return U


def add_buoyancy(U, flags, rho, gravity, rho_star, dt, quirks=False):
    g = _grid(flags, U.shape[1] == 3)
    flags, pf = _f(flags); rho, pr = _f(rho)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    gv = (ctypes.c_float * 3)(*[float(x) for x in gravity])
    lib().ora_add_buoyancy(ctypes.byref(g), U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf, pr, gv,
                           ctypes.c_float(rho_star), ctypes.c_float(dt), int(quirks))
    return U


def add_gravity(U, flags, gravity, dt):
    g = _grid(flags, U.shape[1] == 3)
    flags, pf = _f(flags)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    gv = (ctypes.c_float * 3)(*[float(x) for x in gravity])
    lib().ora_add_gravity(ctypes.byref(g), U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf, gv, ctypes.c_float(dt))
    return U


def add_viscosity(dt, U, flags, viscosity):
    g = _grid(flags, U.shape[1] == 3)
    flags, pf = _f(flags)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    rc = lib().ora_add_viscosity(ctypes.byref(g), ctypes.c_float(dt), U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf,
                                 ctypes.c_float(viscosity))
    assert rc == 0, "addViscosity is 2D only (reference viscosity.py:5)"
    return U


def set_wall_bcs(U, flags):
    g = _grid(flags, U.shape[1] == 3)
    flags, pf = _f(flags)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    lib().ora_set_wall_bcs(ctypes.byref(g), U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf)
    return U


def set_wall_bcs_stick(U, flags, flags_stick):
    """setWallBcsStick (2D): returns the new U."""
    g = _grid(flags, False)
    flags, pf = _f(flags)
    fs, ps = _f(flags_stick)
    Uin, pu = _f(U)
    out = np.empty_like(Uin)
    rc = lib().ora_set_wall_bcs_stick(ctypes.byref(g), pu, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pf, ps)
    if rc != 0:
        raise RuntimeError("setWallBcsStick: 2D only")
    return out


def set_const_vals(U, UBC, UBCInvMask, rho, rhoBC, rhoBCInvMask):
    g = _grid(rho, U.shape[1] == 3)
    U = np.array(U, dtype=np.float32, order="C", copy=True)
    rho = np.array(rho, dtype=np.float32, order="C", copy=True)
    a, pa = _f(UBC); b, pb = _f(UBCInvMask); c, pc = _f(rhoBC); d, pd = _f(rhoBCInvMask)
    lib().ora_set_const_vals(ctypes.byref(g), U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pa, pb,
                             rho.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), pc, pd)
    return U, rho


def flags_to_occupancy(flags):
    g = _grid(flags)
    flags, pf = _f(flags)
    occ = np.empty_like(flags)
    lib().ora_flags_to_occupancy(ctypes.byref(g), pf, occ.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return occ


def empty_domain(B, D, H, W, bnd=1):
    g = OraGrid(B, D, H, W, int(D > 1), _SLAB[0] if D > 1 else 0, _SLAB[1] if D > 1 else 0)
    flags = np.empty((B, 1, D, H, W), np.float32)
    lib().ora_empty_domain(ctypes.byref(g), flags.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(bnd))
    return flags


def create_cylinder(flags, cx, cy, radius):
    """geometry_utils.py:4-34; returns a new flags array"""
    f, pf = _f(flags.copy())
    g = _grid(f)
    lib().ora_create_cylinder(ctypes.byref(g), pf, ctypes.c_double(cx), ctypes.c_double(cy), ctypes.c_double(radius))
    return f


def create_box2d(flags, x0, x1, y0, y1):
    f, pf = _f(flags.copy())
    g = _grid(f)
    lib().ora_create_box2d(ctypes.byref(g), pf, ctypes.c_double(x0), ctypes.c_double(x1), ctypes.c_double(y0), ctypes.c_double(y1))
    return f


def get_centered(U):
    """grid.py:7-32"""
    u, pu = _f(U)
    B, nc, D, H, W = u.shape
    g = OraGrid(B, D, H, W, int(nc == 3), 0, 0)
    out = np.empty((B, 3, D, H, W), np.float32)
    lib().ora_get_centered(ctypes.byref(g), pu, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def velocity_divergence_backward(gdiv, flags, is3d=None):
    gd, pg = _f(gdiv); f, pf = _f(flags)
    g = _grid(f, is3d)
    out = np.empty((f.shape[0], 3 if g.is3D else 2) + f.shape[2:], np.float32)
    lib().ora_velocity_divergence_backward(ctypes.byref(g), pg, pf, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def velocity_update_backward(gout, flags):
    go, pg = _f(gout); f, pf = _f(flags)
    g = _grid(f, go.shape[1] == 3)
    gU = np.empty_like(go); gp = np.empty_like(f)
    lib().ora_velocity_update_backward(ctypes.byref(g), pg, pf, gU.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       gp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return gU, gp


def pack_weights(wdict, ndim=2):
    """Flatten a torch-named weight dict into the canonical blob (see cnn_oracle.c header)."""
    from fluidnet_cxx_amd.weights import scalenet_layers
    parts = []
    for L in scalenet_layers(2, ndim):
        parts.append(np.ascontiguousarray(wdict[L["name"] + ".weight"], np.float32).ravel())
        parts.append(np.ascontiguousarray(wdict[L["name"] + ".bias"], np.float32).ravel())
    blob = np.concatenate(parts)
    assert blob.size == lib().ora_scalenet_weight_floats(int(ndim == 3))
    return blob


def multiscale_forward(blob, x, is3d=None):
    """x: (B,2,H,W) or (B,2,D,H,W) -> (B,1,...); is3d defaults to "x has more than one z plane" """
    x5 = x if x.ndim == 5 else x[:, :, None]
    B, _, D, H, W = x5.shape
    if is3d is None:
        is3d = D > 1
    g = OraGrid(B, D, H, W, int(is3d), 0, 0)
    x5, px = _f(x5); blob, pb = _f(blob)
    p = np.empty((B, 1, D, H, W), np.float32)
    lib().ora_multiscale_forward(ctypes.byref(g), pb, px, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return p if x.ndim == 5 else p[:, :, 0]


def multiscale_forward_crop(blob, x, trim):
    """The forward pass on nested z-crops (the checker of fnx_multiscale_forward_crop; ora_multiscale_forward_crop):
    x (B,2,D,H,W), trim = [full-tower low, high, half-tower low, high] -> p (B,1,D - trim[0] - trim[1],H,W)"""
    B, _, D, H, W = x.shape
    g = OraGrid(B, D, H, W, 1, 0, 0)
    x, px = _f(x); blob, pb = _f(blob)
    tr = (ctypes.c_int * 4)(*[int(t) for t in trim])
    p = np.empty((B, 1, D - int(trim[0]) - int(trim[1]), H, W), np.float32)
    rc = lib().ora_multiscale_forward_crop(ctypes.byref(g), pb, px, tr, p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0, "multiscale_forward_crop: 3D only, D and the trims multiples of 4, half window containing the full one"
    return p


def scale_std(U, thr=1e-5):
    B, nc, D, H, W = U.shape
    g = OraGrid(B, D, H, W, int(nc == 3), 0, 0)
    U, pu = _f(U)
    s = np.empty((B,), np.float32)
    lib().ora_scale_std(ctypes.byref(g), pu, ctypes.c_float(thr), s.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return s


def fluidnet_forward(blob, inp, thr=1e-5, net=None):
    """FluidNet.forward (lib/model.py:118-227; ora_fluidnet_forward).  `net`: a callable x (B,2,D,H,W) -> p (B,1,D,H,W) that
    stands in for the MultiScaleNet (lib/multi_scale_net.py:118-127) -- the stages around the net are then the oracle's own
    operators applied in ora_fluidnet_forward's order with its arithmetic (fp32 divisions and products per element), so a test
    can pin everything a step does around a net whose output it takes from elsewhere."""
    B, cin, D, H, W = inp.shape
    nc = cin - 3
    if net is not None:
        inp = np.ascontiguousarray(inp, np.float32)
        U = inp[:, 1:1 + nc].copy()
        flags = inp[:, 1 + nc:2 + nc].copy()
        div = velocity_divergence(U, flags)
        s = scale_std(U, thr).reshape(B, 1, 1, 1, 1)
        U = (U / s).astype(np.float32)
        occ = np.where(flags == 1, np.float32(0), np.where(flags == 2, np.float32(1), flags)).astype(np.float32)
        x = np.concatenate([(div / s).astype(np.float32), occ], 1)
        p = np.ascontiguousarray(net(x), np.float32)
        U = velocity_update(p, U, flags)
        p = (p * s).astype(np.float32)
        U = (U * s).astype(np.float32)
        return p, set_wall_bcs(U, flags)
    g = OraGrid(B, D, H, W, int(nc == 3), 0, 0)
    inp, pi = _f(inp); blob, pb = _f(blob)
    p = np.empty((B, 1, D, H, W), np.float32); U = np.empty((B, nc, D, H, W), np.float32)
    lib().ora_fluidnet_forward(ctypes.byref(g), pb, pi, ctypes.c_float(thr),
                               p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                               U.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return p, U


def simulate_step(state, cfg, method="jacobi", blob=None, quirks=False, net=None):
    """One time step in the order of the reference's lib/simulate.py:28-171.

    state: dict with p, U, flags, density (+ optional UBC, UBCInvMask, densityBC, densityBCInvMask);
    cfg: dt, maccormackStrength, sampleOutsideFluid, buoyancyScale, gravityVec(x,y,z), operatingDensity,
         pTol, jacobiIter.  Returns a new dict (inputs untouched)."""
    U, flags, rho = state["U"], state["flags"], state["density"]
    is3d = U.shape[1] == 3
    dt = float(cfg["dt"])
    has_bc = "UBC" in state

    def const_vals(U, rho):
        if has_bc:
            return set_const_vals(U, state["UBC"], state["UBCInvMask"], rho, state["densityBC"],
                                  state["densityBCInvMask"])
        return U, rho

    orig = U
    if cfg.get("viscosity", 0) > 0:                              # simulate.py:66-69
        orig = add_viscosity(dt, U, flags, cfg["viscosity"])
    rho = advect_scalar(dt, rho, U, flags, "maccormackFluidNet", 1, cfg.get("sampleOutsideFluid", False),
                        cfg["maccormackStrength"], quirks)
    if cfg.get("correctScalar", False):                          # cpp/advection.py:9-12
        div = velocity_divergence(U, flags)
        half = np.float32(dt * 0.5)
        rho = np.where(flags == 1, rho + (half * rho) * div, rho).astype(np.float32)
    U = advect_vel(dt, orig, U, flags, "maccormackFluidNet", 1, cfg["maccormackStrength"], quirks)
    U, rho = const_vals(U, rho)
    bs = cfg.get("buoyancyScale", 0)
    if bs > 0:
        gv = cfg["gravityVec"]
        gvec = (np.array([gv["x"], gv["y"], gv["z"]], np.float32) * np.float32(-bs)).astype(np.float32)
        U = add_buoyancy(U, flags, rho, gvec, cfg.get("operatingDensity", 0.0), dt, quirks)
    gs = cfg.get("gravityScale", 0)
    if gs > 0:                                                   # simulate.py:107-114
        gv = cfg["gravityVec"]
        gvec = (np.array([gv["x"], gv["y"], gv["z"]], np.float32) * np.float32(-gs)).astype(np.float32)
        U = add_gravity(U, flags, gvec, dt)
    periodic = "periodic-x" in cfg and "periodic-y" in cfg

    def wall_bcs(U):                                             # simulate.py:120-128
        out = set_wall_bcs(U, flags)
        if periodic and cfg["periodic-x"]:
            out[:, 1, :, :, 1] = U[:, 1, :, :, -1]
        if periodic and cfg["periodic-y"]:
            out[:, 0, :, 1] = U[:, 0, :, -1]
        return out

    stick = state.get("flags_stick")                             # simulate.py:61-64
    if method == "jacobi":
        U = wall_bcs(U)
    elif stick is not None:                                      # simulate.py:129-130
        U = set_wall_bcs_stick(U, flags, stick)
    U, rho = const_vals(U, rho)
    if method == "jacobi":
        div = velocity_divergence(U, flags)
        p, _, _ = jacobi(flags, div, is3d, cfg.get("pTol", 0.0), cfg["jacobiIter"], quirks)
        U = velocity_update(p, U, flags)
        U = wall_bcs(U)
    else:
        inp = np.concatenate([state["p"], U, flags, rho], 1)
        p, U = fluidnet_forward(blob, inp, cfg.get("normalizeInputThreshold", 1e-5), net)
        if stick is not None:                                    # simulate.py:165-166
            U = set_wall_bcs_stick(U, flags, stick)
    U, rho = const_vals(U, rho)
    out = dict(state)
    out.update(p=p, U=U, density=rho)
    return out
