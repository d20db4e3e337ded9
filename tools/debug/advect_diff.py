"""GPU debug: cells where fnx_advect_step differs from advectScalar + advectVelocity, with their surroundings."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from util import random_state
from fluidnet_cxx_amd import fluid as fl
from fluidnet_cxx_amd._ext import ext
B, D, H, W, sigma = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
s = random_state(B, D, H, W, sigma, seed=21, empties=True)
dev = torch.device("cuda")
tf, tU, trho = (torch.from_numpy(s[k]).to(dev) for k in ("flags", "U", "rho"))
for so in (False, True):
    r, u = ext.advect_step(0.13, trho, tU, tf, so, 0.7)
    wr = fl.advectScalar(0.13, trho, tU, tf, "maccormackFluidNet", 1, so, 0.7)
    wu = fl.advectVelocity(0.13, tU, tU, tf, "maccormackFluidNet", 1, 0.7)
    for name, a, b in (("rho", r, wr), ("U", u, wu)):
        bad = (a.view(torch.int32) != b.view(torch.int32)).cpu().numpy()
        idx = np.argwhere(bad)
        print(f"so={so} {name}: {len(idx)} cells differ")
        for (bb, c, k, j, i) in idx[:40]:
            fn = s["flags"][bb, 0, max(k-1,0):k+2, max(j-1,0):j+2, max(i-1,0):i+2]
            print(f"   b={bb} c={c} k={k} j={j} i={i}  got {a[bb,c,k,j,i].item():+.6e} want {b[bb,c,k,j,i].item():+.6e}  nonfluid in 3x3x3: {int((fn != 1).sum())}  "
                  f"|U|dt={np.abs(s['U'][bb,:,k,j,i]).max()*0.13:.3f}  tile x {i//64}.{i%64} y {j//8}.{j%8}")
