"""The CPU oracle against golden vectors captured from the reference's ATen implementation
(tools/make_golden.py).  Bit-exact for every stencil / advection op in 2D (all CFL) and for the reference's own
3D behaviour (quirks mode); fp32 tolerance for the CNN (conv arithmetic lives in PyTorch)."""
import numpy as np
import pytest

from util import PLUME_CFG, assert_bitexact, assert_close, plume_state

CASES_2D = ["ops_2d_a", "ops_2d_b", "ops_2d_c", "ops_2d_d"]
CASES_3D = ["ops_3d_a", "ops_3d_b"]


@pytest.mark.parametrize("case", CASES_2D + CASES_3D)
def test_ops_bitexact(oracle, golden, case):
    z = golden(case)
    O = oracle
    is3d, dt, q = bool(z["is3d"]), float(z["dt"]), bool(z["is3d"])
    flags, U, rho, p = z["flags"], z["U"], z["rho"], z["p"]
    for meth in ("maccormackFluidNet", "eulerFluidNet"):
        for so in (0, 1):
            assert_bitexact(O.advect_scalar(dt, rho, U, flags, meth, 1, bool(so), 0.6, q), z[f"advect_scalar_{meth}_{so}"],
                            f"advect_scalar {meth} so={so}")
        assert_bitexact(O.advect_vel(dt, U, U, flags, meth, 1, 0.6, q), z[f"advect_vel_{meth}"], f"advect_vel {meth}")
    assert_bitexact(O.advect_vel(dt, z["orig"], U, flags, "maccormackFluidNet", 1, 0.75, q), z["advect_vel_orig"], "advect_vel orig")
    assert_bitexact(O.velocity_divergence(U, flags), z["divergence"], "divergence")
    pj, res, _ = O.jacobi(flags, z["divergence"], is3d, 0.0, int(z["jacobi_iters"]), q)
    assert_bitexact(pj, z["jacobi_p"], "jacobi"); assert abs(res - float(z["jacobi_res"])) <= 1e-5 * float(z["jacobi_res"])
    pj, res, _ = O.jacobi(flags, z["divergence"], is3d, 0.0, 1, q)
    assert_bitexact(pj, z["jacobi1_p"], "jacobi 1 sweep")
    if "jacobi_tol" in z:
        pj, res, it = O.jacobi(flags, z["divergence"], is3d, float(z["jacobi_tol"]), 50, q)
        assert_bitexact(pj, z["jacobi_tol_p"], "jacobi p_tol exit")
        assert it < 50
    if not is3d:
        assert_bitexact(O.velocity_update(p, U, flags), z["velocity_update"], "velocity_update")
        assert_bitexact(O.set_wall_bcs(U, flags), z["set_wall_bcs"], "set_wall_bcs")
    assert_bitexact(O.add_buoyancy(U, flags, rho, z["gravity"], float(z["rho_star"]), dt, q), z["add_buoyancy"], "add_buoyancy")
    assert_bitexact(O.flags_to_occupancy(flags), z["occupancy"], "occupancy")
    assert_bitexact(O.add_gravity(U, flags, z["gravity"], dt), z["add_gravity"], "add_gravity")
    if not is3d:
        assert_bitexact(O.add_viscosity(dt, U, flags, float(z["viscosity"])), z["add_viscosity"], "add_viscosity")


def test_plume128_ops(oracle, golden):
    z = golden("plume128")
    U, rho, flags = z["U_60"], z["density_60"], z["flags"]
    assert_bitexact(oracle.advect_scalar(0.1, rho, U, flags, "maccormackFluidNet", 1, False, 0.6), z["advect_scalar"], "advect_scalar")
    assert_bitexact(oracle.advect_vel(0.1, U, U, flags, "maccormackFluidNet", 1, 0.6), z["advect_vel"], "advect_vel")
    assert_bitexact(oracle.velocity_divergence(U, flags), z["divergence"], "div")
    p, res, _ = oracle.jacobi(flags, z["divergence"], False, 0.0, 28)
    assert_bitexact(p, z["jacobi28_p"], "jacobi28")


def test_plume128_simulation_20_steps(oracle, golden):
    """Config C1: the whole lib.simulate loop, bit-exact after 1, 5 and 20 steps."""
    z = golden("plume128")
    st = plume_state(128)
    assert_bitexact(st["flags"], z["flags"], "flags")
    for k in ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
        assert_bitexact(st[k], z[k], k)
    for it in range(1, 21):
        st = oracle.simulate_step(st, PLUME_CFG, "jacobi")
        if it in (1, 5, 20):
            for k in ("U", "density", "p"):
                assert_bitexact(st[k], z[f"{k}_{it}"], f"{k} after {it} steps")


def test_generators(oracle, golden):
    z = golden("generators")
    assert_bitexact(oracle.empty_domain(2, 6, 7, 8), z["empty3d"], "emptyDomain 3D")
    for res in (16, 128):
        st = plume_state(res)
        assert_bitexact(st["flags"], z[f"plume{res}_flags"], "flags")
        for k in ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
            assert_bitexact(st[k], z[f"plume{res}_{k}"], k)


def test_geometry_generators(oracle, golden):
    """createCylinder against the reference's flags (2D and 3D); createBox2D (the reference's cannot run) against its
    documented meaning; getCentered against the reference's output."""
    from util import make_flags
    z = golden("generators")
    for tag, shape in (("cyl2d", (1, 1, 40, 32)), ("cyl3d", (1, 5, 24, 28))):
        got = oracle.create_cylinder(make_flags(*shape, boxes=False), 15.5, 20.0, 6.3)
        assert_bitexact(got, z[tag + "_flags"], tag)
    got = oracle.create_box2d(make_flags(1, 1, 20, 30, boxes=False), 5, 9, 3, 6)
    want = make_flags(1, 1, 20, 30, boxes=False); want[0, 0, 0, 3:6, 5:9] = 2
    assert_bitexact(got, want, "box2d")
    zg = golden("grid")
    for tag in ("2d", "3d"):
        assert_bitexact(oracle.get_centered(zg[f"U_{tag}"]), zg[f"centered_{tag}"], f"getCentered {tag}")


def test_cnn_tolerance(oracle, golden):
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    c = golden("cnn")
    blob = oracle.pack_weights(make_scalenet_weights(0))
    assert_close(oracle.multiscale_forward(blob, c["x"]), c["multiscale"], 2e-5, "MultiScaleNet")
    assert_close(oracle.scale_std(c["fluidnet_in"][:, 1:3]), c["scale"].ravel(), 1e-6, "_ScaleNet std")
    p, U = oracle.fluidnet_forward(blob, c["fluidnet_in"])
    assert_close(p, c["fluidnet_p"], 2e-5, "FluidNet p"); assert_close(U, c["fluidnet_U"], 2e-5, "FluidNet U")


def test_sim64_convnet(oracle, golden):
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    s = golden("sim64")
    blob = oracle.pack_weights(make_scalenet_weights(0))
    for method in ("jacobi", "convnet"):
        st = plume_state(64)
        for it in range(1, 4):
            st = oracle.simulate_step(st, PLUME_CFG, method, blob)
            if it in (1, 3):
                for k in ("U", "density", "p"):
                    if method == "jacobi":
                        assert_bitexact(st[k], s[f"{method}_{k}_{it}"], f"{method} {k} {it}")
                    else:
                        assert_close(st[k], s[f"{method}_{k}_{it}"], 1e-5, f"{method} {k} {it}")


def test_sim64_optional_stages(oracle, golden):
    """lib.simulate with viscosity, correctScalar, gravity and the periodic patches on (all off in shipped configs)."""
    from util import F2_CFG
    s = golden("sim64")
    st = plume_state(64)
    for it in range(1, 7):
        st = oracle.simulate_step(st, dict(PLUME_CFG, **F2_CFG), "jacobi")
        if it in (1, 3, 6):
            for k in ("U", "density", "p"):
                assert_bitexact(st[k], s[f"f2_{k}_{it}"], f"{k} after {it} steps")


def test_set_wall_bcs_stick(oracle, golden):
    """setWallBcsStick (2D): the reference's body with its three unbound names bound by the harness (tools/make_golden.py)."""
    z = golden("stick")
    for n in "ab":
        out = oracle.set_wall_bcs_stick(z[f"{n}_U"], z[f"{n}_flags"], z[f"{n}_flags_stick"])
        assert_bitexact(out, z[f"{n}_out"], f"set_wall_bcs_stick {n}")
        assert (out != z[f"{n}_U"]).sum() > 100          # (the case does exercise the operator)


def test_sim64_stick_convnet(oracle, golden):
    """lib.simulate with 'flags_stick' in the batch (convnet method, no-slip cylinder)."""
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    z = golden("stick")
    blob = oracle.pack_weights(make_scalenet_weights(0))
    st = plume_state(64)
    st["flags"] = z["sim_flags"]; st["flags_stick"] = z["sim_flags_stick"]
    for it in range(1, 3):
        st = oracle.simulate_step(st, PLUME_CFG, "convnet", blob)
        for k in ("U", "density", "p"):
            assert_close(st[k], z[f"sim_{k}_{it}"], 1e-5, f"stick convnet {k} {it}")


def test_known_answers(oracle):
    """Properties measured on the reference."""
    from util import make_flags
    flags = make_flags(1, 1, 20, 24, boxes=False)
    rng = np.random.default_rng(0)
    rho = rng.random((1, 1, 1, 20, 24)).astype(np.float32)
    U0 = np.zeros((1, 2, 1, 20, 24), np.float32)
    out = oracle.advect_scalar(0.1, rho, U0, flags)
    np.testing.assert_array_equal(out[..., 1:-1, 1:-1], rho[..., 1:-1, 1:-1])
    assert (out[..., 0, :] == 0).all() and (out[..., :, 0] == 0).all()
    assert (oracle.advect_vel(0.1, U0, U0, flags) == 0).all()
    p, res, _ = oracle.jacobi(flags, np.zeros_like(rho), False, 0.0, 5)
    assert (p == 0).all() and res == 0
    # uniform unit shift, euler, sample_outside: exact one-cell shift away from the walls
    U1 = np.zeros_like(U0); U1[:, 0] = 1.0
    sh = oracle.advect_scalar(1.0, rho, U1, flags, "eulerFluidNet", 1, True)
    np.testing.assert_array_equal(sh[..., 2:-2, 3:-2], rho[..., 2:-2, 2:-3])


@pytest.mark.parametrize("shape", [(2, 1, 24, 40), (1, 8, 16, 24)])
def test_fluidnet_forward_net_hook_same_bits(oracle, shape):
    """oracle.fluidnet_forward(..., net=f) -- the stages around the MultiScaleNet written with the oracle's operators so that a
    test can take the net's output from elsewhere (tests/test_fullsize_gpu.py does, at 1024^2 and 256^3) -- is bit for bit
    ora_fluidnet_forward when f is the oracle's own net."""
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    from util import random_state
    B, D, H, W = shape
    is3d = D > 1
    s = random_state(B, D, H, W, 0.7, seed=12)
    blob = oracle.pack_weights(make_scalenet_weights(0, ndim=3 if is3d else 2), 3 if is3d else 2)
    inp = np.concatenate([s["p"], s["U"], s["flags"], s["rho"]], 1)
    p0, U0 = oracle.fluidnet_forward(blob, inp, 1e-5)
    p1, U1 = oracle.fluidnet_forward(blob, inp, 1e-5, net=lambda x: oracle.multiscale_forward(blob, x, is3d))
    assert_bitexact(p1, p0, "p")
    assert_bitexact(U1, U0, "U")
