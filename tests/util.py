import numpy as np

PLUME_CFG = dict(dt=0.1, maccormackStrength=0.6, sampleOutsideFluid=False, buoyancyScale=0.25, gravityScale=0,
                 viscosity=0, correctScalar=False, gravityVec=dict(x=0.0, y=-1.0, z=0.0), operatingDensity=0.0,
                 pTol=0.0, jacobiIter=28, normalizeInputThreshold=1e-5)


F2_CFG = {"viscosity": 0.02, "gravityScale": 0.5, "correctScalar": True, "periodic-x": True, "periodic-y": True}


def make_flags(B, D, H, W, boxes=True, empties=False, seed=0):
    """Border wall + interior obstacles (box, single cell, bar) like tools/make_golden.py."""
    f = np.full((B, 1, D, H, W), 1.0, np.float32)
    f[:, :, :, 0, :] = 2; f[:, :, :, -1, :] = 2; f[:, :, :, :, 0] = 2; f[:, :, :, :, -1] = 2
    if D > 1:
        f[:, :, 0] = 2; f[:, :, -1] = 2
    if boxes and H >= 12 and W >= 12:
        zs = slice(None) if D == 1 else slice(D // 3, D // 3 + 3)
        f[:, :, zs, H // 3:H // 3 + 4, W // 4:W // 4 + 5] = 2
        f[:, :, zs if D == 1 else slice(D // 2, D // 2 + 1), 2 * H // 3, 2 * W // 3] = 2
        if D == 1:
            f[:, :, :, H // 2, W // 2:W // 2 + 7] = 2
    if empties:
        f[:, :, :, 3, 3] = 4; f[:, :, :, 3, 4] = 4; f[:, :, :, H - 4, W - 5] = 4
    return f


def random_state(B, D, H, W, sigma, seed, boxes=True, empties=False):
    rng = np.random.default_rng(seed)
    nc = 3 if D > 1 else 2
    return dict(flags=make_flags(B, D, H, W, boxes, empties),
                U=(rng.standard_normal((B, nc, D, H, W)) * sigma).astype(np.float32),
                rho=rng.random((B, 1, D, H, W)).astype(np.float32),
                p=rng.standard_normal((B, 1, D, H, W)).astype(np.float32))


def plume_state(res, D=1):
    """Initial state + BC masks of the plume configs (reference plume.py:131-163, createPlumeBCs)."""
    import math
    is3d = D > 1
    nc = 3 if is3d else 2
    flags = make_flags(1, D, res, res, boxes=False)
    st = dict(p=np.zeros((1, 1, D, res, res), np.float32), U=np.zeros((1, nc, D, res, res), np.float32),
              density=np.zeros((1, 1, D, res, res), np.float32), flags=flags)
    cx, rad = res // 2, math.floor(res * 0.145)
    x = np.arange(res) - cx
    r2 = (x * x)[None, None, :]
    if is3d:
        z = np.arange(D) - D // 2
        r2 = r2 + (z * z)[:, None, None]
    inside = np.broadcast_to(r2 <= rad * rad, (D, 4, res))
    UBC = np.zeros_like(st["U"]); UBC[0, 1, :, 0:4] = inside * 2.0
    UBCInvMask = np.ones_like(st["U"]); UBCInvMask[:, :, :, 0:4] = 0
    dBC = np.zeros_like(st["density"]); dBC[0, 0, :, 0:4] = inside * np.float32(0.1)
    dMask = np.ones_like(st["density"]); dMask[0, 0, :, 0:4] = ~inside
    st.update(UBC=UBC, UBCInvMask=UBCInvMask, densityBC=dBC.astype(np.float32), densityBCInvMask=dMask)
    return st


def assert_bitexact(a, b, what=""):
    """Same BITS: fp32 arrays are compared through their int32 views, so +0 / -0 differ and equal NaNs compare equal."""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype == np.float32 and b.dtype == np.float32:
        bad = np.ascontiguousarray(a).view(np.int32) != np.ascontiguousarray(b).view(np.int32)
    else:
        bad = a != b
    if bad.any():
        idx = np.argwhere(bad)[:5]
        with np.errstate(invalid="ignore"):
            d = np.abs(a.astype(np.float64) - b)
        raise AssertionError(f"{what}: {int(bad.sum())}/{a.size} cells differ in bits, max |d| = "
                             f"{np.nanmax(d):.3e}, first at {idx.tolist()}")


def assert_close(a, b, rtol, what=""):
    """|a-b| <= rtol * max(1, |b|max) everywhere (the per-op tolerance form of SURVEY.md 0.10)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    d = np.abs(a - b).max()
    assert d <= rtol * scale, f"{what}: max |d| = {d:.3e} > {rtol:.1e} * {scale:.3e}"


def assert_close_rel(a, b, rtol, what=""):
    """|a-b| <= rtol * |b|max everywhere: relative to the reference's own magnitude, no floor at 1 (CNN outputs are
    O(0.1) on random weights, where the max(1, .) form would be 10x looser than it reads)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.abs(b).max())
    d = np.abs(a - b).max()
    assert d <= rtol * scale, f"{what}: max |d| = {d:.3e} > {rtol:.1e} * |ref|max {scale:.3e} (relative {d / max(scale, 1e-300):.2e})"
