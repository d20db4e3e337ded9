"""GPU-box probe: where the single-launch step of a small 2D grid (csrc/fnx_small.hip) spends its time.  Needs the library built
with FNX_EXTRA_HIPCC_FLAGS=-DFNX_SMALL_STAMPS (workgroup 0 then leaves the 100-MHz clock at every phase boundary in the workspace
words behind the arrival counter).  usage: small_step_phases.py [res] [jacobiIter]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from fluidnet_cxx_amd import simulate
from fluidnet_cxx_amd._ext import ext
from util import PLUME_CFG, plume_state
dev = torch.device('cuda:0')
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 28
m = dict(PLUME_CFG, jacobiIter=iters)
bd = {k: torch.from_numpy(v).to(dev) for k, v in plume_state(res).items()}
ws = torch.zeros(ext.step_workspace_bytes(1, 1, res, res, False), dtype=torch.uint8, device=dev)
al = lambda x: (x + 255) & ~255
n = res * res
off = al(4 * n) + al(8 * n) + al(4 * n) + al(n) + al(8 * n)          # rho2, U2, div, class map, viscous velocity -> the counter's slot
acc = None
for it in range(60):
    simulate(m, bd, None, "jacobi", workspace=ws, static_flags=(0, 3, 7)[min(it, 2)])
    torch.cuda.synchronize()
    if it >= 20:
        st = ws[off:off + 256].cpu().numpy().view(np.uint64)[1:].astype(np.int64)
        d = st - st[0]
        acc = d if acc is None else acc + d
acc = acc / 40.0 * 0.01          # us
nr = (iters + 15) // 16
names = {1: "A advect fwd", 2: "barrier", 3: "B advect bwd", 4: "barrier", 5: "C stage+div", 6: "barrier"}
for r in range(nr):
    names[7 + 2 * r] = f"D jacobi round {r}"; names[8 + 2 * r] = "barrier"
names[30] = "E post-projection"
prev = 0.0
for k in sorted(names):
    print(f"{names[k]:24s} {acc[k - 1] - prev:7.2f} us   (at {acc[k - 1]:7.2f})")
    prev = acc[k - 1]
