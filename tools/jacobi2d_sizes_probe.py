"""GPU-box probe: 2D Jacobi solve time (28 and 100 sweeps) by grid size, HIP events around fnx_jacobi_sweeps, graph-free.
python tools/jacobi2d_sizes_probe.py"""
import sys, torch
sys.path.insert(0, ".")
from fluidnet_cxx_amd import fluid as fl
from fluidnet_cxx_amd._ext import ext
dev = torch.device("cuda:0")
for res in (128, 192, 256, 384, 512, 768, 1024, 1536, 2048):
    flags = torch.zeros(1, 1, 1, res, res, device=dev); fl.emptyDomain(flags)
    div = torch.randn(1, 1, 1, res, res, device=dev)
    p = torch.zeros_like(div)
    ws = torch.empty(ext.jacobi_workspace_bytes(1, 1, res, res, False), dtype=torch.uint8, device=dev)
    for n in (28, 100):
        for _ in range(5):
            ext.jacobi_sweeps_(flags, div, p, False, n, ws, False, None, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ext.jacobi_sweeps_(flags, div, p, False, n, ws, False, None, True)
        e1.record(); torch.cuda.synchronize()
        print(f"{res:5d}^2 x {n:3d} sweeps: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per solve")
