"""GPU experiment: time of one 3D Jacobi solve (us per 2-sweep pass) for a grid.
   python tools/jacobi3d_time.py D H W [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from fluidnet_cxx_amd import fluid
D, H, W = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda")
flags = torch.zeros(1, 1, D, H, W, device=dev); fluid.emptyDomain(flags)
div = torch.randn(1, 1, D, H, W, device=dev)
for _ in range(3):
    fluid.solveLinearSystemJacobi(flags, div, True, 0.0, iters)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fluid.solveLinearSystemJacobi(flags, div, True, 0.0, iters)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print(f"{D}x{H}x{W} Jacobi-{iters}: {us:9.1f} us per solve, {us / (iters / 2):7.2f} us per 2-sweep pass (incl. mask build + launch gaps)")
