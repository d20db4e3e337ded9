"""GPU-box check: long simulations, HIP vs the CPU oracle, compared bit for bit every few steps.
2D 160x200 with obstacles (300 steps, Jacobi-28) and 3D 40x48x56 with obstacles (120 steps, Jacobi-40)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from util import PLUME_CFG, make_flags
from oracle import oracle as O
from fluidnet_cxx_amd import simulate
dev = torch.device("cuda:0")


def state(D, H, W, seed):
    rng = np.random.default_rng(seed)
    nc = 3 if D > 1 else 2
    flags = make_flags(1, D, H, W, boxes=True)
    st = dict(flags=flags, p=np.zeros((1, 1, D, H, W), np.float32), U=np.zeros((1, nc, D, H, W), np.float32),
              density=np.zeros((1, 1, D, H, W), np.float32))
    UBC = np.zeros_like(st["U"]); M = np.ones_like(st["U"])
    UBC[0, 1, :, 0:4, W // 3:2 * W // 3] = 2.0; M[:, :, :, 0:4] = 0
    dBC = np.zeros_like(st["density"]); dM = np.ones_like(st["density"])
    dBC[0, 0, :, 0:4, W // 3:2 * W // 3] = 0.1; dM[0, 0, :, 0:4, W // 3:2 * W // 3] = 0
    st["U"] += (rng.standard_normal(st["U"].shape) * 0.05).astype(np.float32)
    st.update(UBC=UBC, UBCInvMask=M, densityBC=dBC, densityBCInvMask=dM)
    return st


for name, (D, H, W), steps, iters, every in (("2D 160x200", (1, 160, 200), 300, 28, 50), ("3D 40x48x56", (40, 48, 56), 120, 40, 20)):
    cfg = dict(PLUME_CFG, jacobiIter=iters, gravityVec=dict(x=0.0, y=-1.0, z=0.2 if D > 1 else 0.0))
    st = state(D, H, W, 5)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in st.items()}
    ws = None
    worst = 0
    for it in range(1, steps + 1):
        simulate(cfg, bd, None, "jacobi")
        st = O.simulate_step(st, cfg, "jacobi")
        if it % every == 0:
            for k in ("U", "density", "p"):
                a = bd[k].cpu().numpy()
                bad = int((a != st[k]).sum())
                worst = max(worst, bad)
                assert bad == 0, f"{name}: {k} differs on {bad} cells after {it} steps (max {np.abs(a - st[k]).max():.3e})"
            print(f"{name}: bit-identical after {it} steps (|U|max {np.abs(st['U']).max():.3f}, rho max {st['density'].max():.3f})", flush=True)
print("long parity ok")
