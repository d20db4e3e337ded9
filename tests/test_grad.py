"""Adjoints of the stencil operators the training graph differentiates through (f4: velocityUpdate -> setWallBcs ->
velocityDivergence, lib/model.py:190-227, fluid_net_train.py:366).  Goldens are gradients from the REFERENCE's own autograd
over its ATen chains (tests/golden/grad.npz, tools/make_golden.py:gen_grad).  CPU: the oracle's per-cell adjoints chained by
hand against them; GPU: torch autograd through the native operators against them, against the oracle in 3D, and the
adjoint identity <J x, w> == <x, J^T w>."""
import numpy as np
import pytest

from util import assert_bitexact, assert_close_rel, make_flags


def _chain_oracle(O, z, tag):
    """d loss / d U_in, d loss / d p of loss = sum(wd * div) + sum(wu * U) with the oracle's adjoints."""
    flags = z[f"{tag}_flags"]
    g_div = z[f"{tag}_wd"]
    g_U = z[f"{tag}_wu"] + O.velocity_divergence_backward(g_div, flags, False)      # U feeds the divergence and the loss
    g_U = O.set_wall_bcs(g_U, flags)                                                   # setWallBcs is its own adjoint
    return O.velocity_update_backward(g_U, flags)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_adjoints_vs_reference_autograd(oracle, golden, tag):
    z = golden("grad")
    flags = z[f"{tag}_flags"]
    # forward chain first (the goldens hold it too)
    U = oracle.velocity_update(z[f"{tag}_p"], z[f"{tag}_U"], flags)
    U = oracle.set_wall_bcs(U, flags)
    assert_bitexact(U, z[f"{tag}_U_out"], "U after velocityUpdate + setWallBcs")
    assert_bitexact(oracle.velocity_divergence(U, flags), z[f"{tag}_div"], "divergence")
    gU, gp = _chain_oracle(oracle, z, tag)
    # each adjoint on its own is exact (no products, at most two terms per entry); along the chain an entry of grad_U sums
    # three terms and one of grad_p up to four, in an order autograd chooses: a few ulp
    assert_close_rel(gU, z[f"{tag}_grad_U"], 1e-6, "grad U")
    assert_close_rel(gp, z[f"{tag}_grad_p"], 1e-6, "grad p")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_autograd_vs_reference_autograd(golden, tag):
    import torch
    from fluidnet_cxx_amd import fluid
    z = golden("grad")
    dev = torch.device("cuda:0")
    T = lambda k: torch.from_numpy(z[f"{tag}_{k}"].copy()).to(dev)
    flags = T("flags")
    U0 = T("U").requires_grad_(True)
    p = T("p").requires_grad_(True)
    U = U0 * 1.0
    assert fluid.velocityUpdate(pressure=p, U=U, flags=flags) is None          # in place, like the reference
    U = fluid.setWallBcs(U, flags)
    div = fluid.velocityDivergence(U.contiguous(), flags)
    loss = (div * T("wd")).sum() + (U * T("wu")).sum()
    loss.backward()
    assert_bitexact(U.detach().cpu().numpy(), z[f"{tag}_U_out"], "U")
    assert_bitexact(div.detach().cpu().numpy(), z[f"{tag}_div"], "div")
    assert_close_rel(U0.grad.cpu().numpy(), z[f"{tag}_grad_U"], 1e-6, "grad U")
    assert_close_rel(p.grad.cpu().numpy(), z[f"{tag}_grad_p"], 1e-6, "grad p")
    # no graph, no autograd Function: the plain launches
    with torch.no_grad():
        V = T("U")
        fluid.velocityUpdate(T("p"), V, flags)
        assert not V.requires_grad and fluid.velocityDivergence(V, flags).grad_fn is None


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 1, 33, 70), (1, 9, 20, 70)])
def test_adjoints_vs_oracle_and_dot_product(oracle, shape):
    """2D and 3D (the reference's 3D velocityUpdate raises, so 3D has no autograd golden): the native adjoints equal the
    oracle's bit for bit, and they ARE the adjoints of the native forward operators: <J x, w> == <x, J^T w> in fp64."""
    import torch
    from fluidnet_cxx_amd import fluid
    from fluidnet_cxx_amd._ext import ext
    B, D, H, W = shape
    is3d = D > 1
    nc = 3 if is3d else 2
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    flags_np = make_flags(B, D, H, W, boxes=True)
    if not is3d:
        flags_np[:, :, :, 5, 7:12] = 4                                         # Empty cells (2D masks m_fe / m_ef)
    flags = torch.from_numpy(flags_np).to(dev)
    gd = rng.standard_normal((B, 1, D, H, W)).astype(np.float32)
    gu = rng.standard_normal((B, nc, D, H, W)).astype(np.float32)
    a = ext.velocity_divergence_backward(torch.from_numpy(gd).to(dev), flags, is3d, None).cpu().numpy()
    assert_bitexact(a, oracle.velocity_divergence_backward(gd, flags_np, is3d), "divergence adjoint")
    gU, gp = ext.velocity_update_backward(torch.from_numpy(gu).to(dev), flags, None)
    oU, op = oracle.velocity_update_backward(gu, flags_np)
    assert_bitexact(gU.cpu().numpy(), oU, "velocityUpdate adjoint (U)")
    assert_bitexact(gp.cpu().numpy(), op, "velocityUpdate adjoint (p)")
    # dot-product test: J = velocityDivergence (linear in U);  K = velocityUpdate (linear in (p, U))
    x = torch.from_numpy(rng.standard_normal((B, nc, D, H, W)).astype(np.float32)).to(dev)
    Jx = fluid.velocityDivergence(x, flags).double()
    lhs = float((Jx * torch.from_numpy(gd).to(dev).double()).sum())
    rhs = float((x.double() * torch.from_numpy(a).to(dev).double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)
    pr = torch.from_numpy(rng.standard_normal((B, 1, D, H, W)).astype(np.float32)).to(dev)
    Ux = x.clone()
    fluid.velocityUpdate(pr, Ux, flags)
    lhs = float((Ux.double() * torch.from_numpy(gu).to(dev).double()).sum())
    rhs = float((x.double() * gU.double()).sum() + (pr.double() * gp.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (lhs, rhs)


def _aten_forward_3d(torch, p, U, flags):
    """velocityUpdate -> velocityDivergence in 3D as plain (differentiable) tensor expressions: the default 3D semantics the
    oracle and the kernels implement (fluid-fluid faces only, the reference's own 3D velocityUpdate raises:
    solver_cpp/src/projection/update_vel.cpp:58-117 is the intent), written independently of both."""
    f, P = flags[:, 0], p[:, 0]
    D, H, W = f.shape[1:]
    inner = (slice(None), slice(1, D - 1), slice(1, H - 1), slice(1, W - 1))
    comps = []
    for a, (dk, dj, di) in enumerate([(0, 0, 1), (0, 1, 0), (1, 0, 0)]):
        lower = (slice(None), slice(1 - dk, D - 1 - dk), slice(1 - dj, H - 1 - dj), slice(1 - di, W - 1 - di))
        m = ((f[inner] == 1) & (f[lower] == 1)).to(U.dtype)
        comps.append(m * (U[:, a][inner] - (P[inner] - P[lower])))
    V = U.clone()
    V[:, :, 1:D - 1, 1:H - 1, 1:W - 1] = torch.stack(comps, 1)
    div = torch.zeros_like(p)
    d = (V[:, 0, 1:-1, 1:-1, 1:-1] - V[:, 0, 1:-1, 1:-1, 2:]) + (V[:, 1, 1:-1, 1:-1, 1:-1] - V[:, 1, 1:-1, 2:, 1:-1]) \
        + (V[:, 2, 1:-1, 1:-1, 1:-1] - V[:, 2, 2:, 1:-1, 1:-1])
    div[:, 0, 1:-1, 1:-1, 1:-1] = d * (f[inner] != 2).to(U.dtype)
    return V, div


def _case_3d():
    rng = np.random.default_rng(21)
    B, D, H, W = 2, 9, 12, 18
    flags = make_flags(B, D, H, W, boxes=True)
    return dict(flags=flags, p=rng.standard_normal((B, 1, D, H, W)).astype(np.float32),
                U=rng.standard_normal((B, 3, D, H, W)).astype(np.float32),
                wd=rng.standard_normal((B, 1, D, H, W)).astype(np.float32),
                wu=rng.standard_normal((B, 3, D, H, W)).astype(np.float32))


def test_oracle_adjoints_3d_vs_torch_autograd(oracle):
    """3D: the reference has no working 3D velocityUpdate to differentiate, so the gradients are pinned to torch autograd over
    an independent tensor formulation of the default 3D forward: forward values and both gradients."""
    import torch
    c = _case_3d()
    p = torch.from_numpy(c["p"]).requires_grad_(True)
    U = torch.from_numpy(c["U"]).requires_grad_(True)
    V, div = _aten_forward_3d(torch, p, U, torch.from_numpy(c["flags"]))
    assert_close_rel(oracle.velocity_update(c["p"], c["U"], c["flags"]), V.detach().numpy(), 1e-6, "3D velocityUpdate")
    assert_close_rel(oracle.velocity_divergence(V.detach().numpy(), c["flags"]), div.detach().numpy(), 1e-6, "3D divergence")
    ((div * torch.from_numpy(c["wd"])).sum() + (V * torch.from_numpy(c["wu"])).sum()).backward()
    gV = c["wu"] + oracle.velocity_divergence_backward(c["wd"], c["flags"], True)
    gU, gp = oracle.velocity_update_backward(gV, c["flags"])
    assert_close_rel(gU, U.grad.numpy(), 1e-6, "3D grad U")
    assert_close_rel(gp, p.grad.numpy(), 1e-6, "3D grad p")


@pytest.mark.gpu
def test_autograd_3d_vs_torch_autograd():
    """The same through torch autograd over the NATIVE 3D operators (in place, mark_dirty) on the GPU."""
    import torch
    from fluidnet_cxx_amd import fluid
    c = _case_3d()
    dev = torch.device("cuda:0")
    p0 = torch.from_numpy(c["p"]).requires_grad_(True)
    U0 = torch.from_numpy(c["U"]).requires_grad_(True)
    V0, div0 = _aten_forward_3d(torch, p0, U0, torch.from_numpy(c["flags"]))
    ((div0 * torch.from_numpy(c["wd"])).sum() + (V0 * torch.from_numpy(c["wu"])).sum()).backward()
    flags = torch.from_numpy(c["flags"]).to(dev)
    p = torch.from_numpy(c["p"]).to(dev).requires_grad_(True)
    U = torch.from_numpy(c["U"]).to(dev).requires_grad_(True)
    V = U.clone()
    fluid.velocityUpdate(p, V, flags)
    div = fluid.velocityDivergence(V, flags)
    assert_close_rel(V.detach().cpu().numpy(), V0.detach().numpy(), 1e-6, "3D velocityUpdate (native)")
    assert_close_rel(div.detach().cpu().numpy(), div0.detach().numpy(), 1e-6, "3D divergence (native)")
    ((div * torch.from_numpy(c["wd"]).to(dev)).sum() + (V * torch.from_numpy(c["wu"]).to(dev)).sum()).backward()
    assert_close_rel(U.grad.cpu().numpy(), U0.grad.numpy(), 1e-6, "3D grad U (native)")
    assert_close_rel(p.grad.cpu().numpy(), p0.grad.numpy(), 1e-6, "3D grad p (native)")
    # the adjoints cover whole fields: a compute window (z-slab driver) has no differentiable form
    from fluidnet_cxx_amd._ext import ext
    with pytest.raises(AssertionError, match="compute window"):
        fluid.velocityDivergence(U, flags, geom=ext.Geom(k_begin=2, k_end=5))
