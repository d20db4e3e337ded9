"""The single-launch step of small 2D grids (csrc/fnx_small.hip; FnxStepParams.static_flags bit 0 set, bit 3 clear): the whole
Jacobi-method step -- both advections, the stages, the solve, the post-projection pass -- as ONE launch with grid barriers
between its phases.  It runs the cell functions of the separate launches, so every comparison here is bit for bit:
the reference's own 128^2 plume goldens, the CPU oracle on ragged grids / batches / obstacle fields, the multi-launch path,
and a HIP-graph replay."""
import numpy as np
import pytest
import torch

from util import PLUME_CFG, assert_bitexact, make_flags, plume_state

pytestmark = pytest.mark.gpu

NO_SINGLE = 8          # FnxStepParams.static_flags bit 3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def to_dev(st, dev):
    return {k: T(v, dev) for k, v in st.items()}


def ws_for(bd):
    from fluidnet_cxx_amd._ext import ext
    f = bd["flags"]
    # (poisoned: the single-launch path must not depend on what a fresh workspace holds)
    return torch.full((ext.step_workspace_bytes(f.size(0), f.size(2), f.size(3), f.size(4), False),), 0xA5, dtype=torch.uint8,
                      device=f.device)


def single_launch_count(fn):
    """number of single-launch steps among the steps fn() runs (HIP-event profile class FNX_PROF_STEP2D = 7)"""
    from fluidnet_cxx_amd._ext import ext
    ext.profile_enable(True)
    fn()
    torch.cuda.synchronize()
    n = ext.profile_read(7)[1]
    ext.profile_enable(False)
    return n


def test_plume128_single_launch_vs_reference(dev, golden):
    """Config C1 (the reference's 128^2 plume, Jacobi-28) with a kept workspace: step 1 takes the five launches, steps 2-20 the
    single launch -- the reference's own fields after 1, 5, 20 steps, bit for bit."""
    from fluidnet_cxx_amd import simulate
    z = golden("plume128")
    bd = to_dev(plume_state(128), dev)
    ws = ws_for(bd)

    def run():
        for it in range(1, 21):
            simulate(PLUME_CFG, bd, None, "jacobi", workspace=ws, static_flags=(0, 3, 7)[min(it - 1, 2)])
            if it in (1, 5, 20):
                for k in ("U", "density", "p"):
                    assert_bitexact(N(bd[k]), z[f"{k}_{it}"], f"{k} after {it} steps")
    assert single_launch_count(run) == 19


CASES = [
    # B, H, W, jacobiIter, extra mconf, BCs
    (1, 128, 128, 28, {}, True),
    (1, 96, 130, 5, {}, True),                      # ragged: partial 32-blocks and 64-cell segments, one round
    (1, 70, 100, 16, dict(sampleOutsideFluid=True), True),
    (1, 64, 160, 17, dict(gravityScale=0.5), True),    # two rounds (9 + 8)
    (2, 64, 96, 40, {}, True),                      # batch of two, three rounds (14 + 14 + 12)
    (3, 40, 40, 1, {}, False),                      # one sweep, no BC arrays
    (1, 33, 65, 33, dict(buoyancyScale=0.0), False),   # 2 x 3 blocks of which 1 x 1 cell columns, no buoyancy
]


@pytest.mark.parametrize("B,H,W,iters,extra,bcs", CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_j{c[3]}" for c in CASES])
def test_single_launch_vs_oracle_and_multi_launch(dev, oracle, B, H, W, iters, extra, bcs):
    """Obstacles, an inflow BC, a rough developed flow: four steps (the last three single-launch) against the oracle's step and
    against the same steps with the single launch switched off."""
    from fluidnet_cxx_amd import simulate
    rng = np.random.default_rng(H * 1000 + W)
    res = max(H, W)
    base = plume_state(res)
    sts = []
    for b in range(B):
        st = {k: np.ascontiguousarray(v[:, :, :, :H, :W]) for k, v in base.items()}
        st["flags"] = make_flags(1, 1, H, W, boxes=min(H, W) >= 40, seed=b)
        st["U"] = (st["U"] + rng.standard_normal(st["U"].shape).astype(np.float32) * np.float32(1.0 + 0.5 * b)).astype(np.float32)
        st["density"] = rng.random(st["density"].shape).astype(np.float32)
        if not bcs:
            for k in ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
                st.pop(k)
        sts.append(st)
    mconf = dict(PLUME_CFG, jacobiIter=iters, **extra)
    cat = {k: np.concatenate([s[k] for s in sts], 0) for k in sts[0]}
    bd_s, bd_m = to_dev(cat, dev), to_dev(cat, dev)
    ws_s, ws_m = ws_for(bd_s), ws_for(bd_m)

    def run():
        for it in range(4):
            sf = (0, 3, 7)[min(it, 2)]
            simulate(mconf, bd_s, None, "jacobi", workspace=ws_s, static_flags=sf)
            simulate(mconf, bd_m, None, "jacobi", workspace=ws_m, static_flags=sf | NO_SINGLE)
            for b in range(B):
                sts[b] = oracle.simulate_step(sts[b], mconf, "jacobi")
            for k in ("U", "density", "p"):
                assert_bitexact(N(bd_s[k]), N(bd_m[k]), f"{k} after step {it + 1}: single launch vs five launches")
                for b in range(B):
                    assert_bitexact(N(bd_s[k][b:b + 1]), sts[b][k], f"{k} of sample {b} after step {it + 1}: single launch vs oracle")
    assert single_launch_count(run) == 3


def test_single_launch_graph_replay(dev):
    """The single launch captured in a HIP graph and replayed (what bench.py times): the arrival counter carries over from replay
    to replay; same bits as eager steps."""
    from fluidnet_cxx_amd import simulate
    st = plume_state(128)
    st["flags"] = make_flags(1, 1, 128, 128, boxes=True)
    bd_g, bd_e = to_dev(st, dev), to_dev(st, dev)
    ws_g, ws_e = ws_for(bd_g), ws_for(bd_e)
    for it in range(3):
        simulate(PLUME_CFG, bd_g, None, "jacobi", workspace=ws_g, static_flags=(0, 3, 7)[it])
        simulate(PLUME_CFG, bd_e, None, "jacobi", workspace=ws_e, static_flags=(0, 3, 7)[it])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        simulate(PLUME_CFG, bd_g, None, "jacobi", workspace=ws_g, static_flags=7)
    # (the capture records the step, it does not run it)
    for _ in range(25):
        g.replay()
        simulate(PLUME_CFG, bd_e, None, "jacobi", workspace=ws_e, static_flags=7)
    torch.cuda.synchronize()
    for k in ("U", "density", "p"):
        assert_bitexact(N(bd_g[k]), N(bd_e[k]), f"{k}: 25 graph steps vs 25 eager steps")
    assert float(bd_g["U"].abs().max()) > 0


def test_grids_beyond_the_block_limit_take_the_launches(dev):
    """more than 32 blocks of 32 x 32 cells: the promise bits change nothing (five launches)"""
    from fluidnet_cxx_amd import simulate
    bd = to_dev(plume_state(192), dev)
    ws = ws_for(bd)

    def run():
        for it in range(3):
            simulate(PLUME_CFG, bd, None, "jacobi", workspace=ws, static_flags=(0, 3, 7)[it])
    assert single_launch_count(run) == 0
