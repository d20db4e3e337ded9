"""GPU experiment: do an interior-sized and an edge-sized plane-range launch of the 3D solver overlap when issued on two
streams?  (FNX_JACOBI_BLOCKS limits the wave slots a launch sizes itself for.)  python tools/jacobi_overlap_probe.py"""
import sys, torch
sys.path.insert(0, ".")
from fluidnet_cxx_amd import fluid
from fluidnet_cxx_amd._ext import ext
dev = torch.device("cuda")
D, H, W = 76, 512, 512
flags = torch.zeros(1, 1, D, H, W, device=dev); fluid.emptyDomain(flags)
div = torch.randn(1, 1, D, H, W, device=dev)
p = torch.randn(1, 1, D, H, W, device=dev); q = torch.zeros_like(p); q2 = torch.zeros_like(p)
ws = torch.empty(ext.jacobi_workspace_bytes(1, D, H, W, True), dtype=torch.uint8, device=dev)
ext.jacobi_pass_(flags, div, p, q, 2, 0, 0, ws, False)          # builds the mask
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def interior(): ext.jacobi_pass_(flags, div, p, q, 2, 16, 60, ws, True)
def edges(): ext.jacobi_pass_(flags, div, p, q2, 2, 2, 16, ws, True, 60)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def serial(): interior(); edges()
def overlapped():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): interior()
    with torch.cuda.stream(s2): edges()
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"interior 44 planes {timeit(interior):.1f} us, edges 14+14 planes {timeit(edges):.1f} us, serial {timeit(serial):.1f} us, two streams {timeit(overlapped):.1f} us")
