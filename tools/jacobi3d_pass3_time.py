"""GPU-box probe (round 5): the THREE-sweep 3D Jacobi pass (jacobi3d_march3_kernel) against the two-sweep pass, per sweep, for whole
passes and for the short plane ranges of the z-slab driver's edge chains.  python tools/jacobi3d_pass3_time.py [D H W]"""
import sys
import torch
sys.path.insert(0, ".")
from fluidnet_cxx_amd import fluid
from fluidnet_cxx_amd._ext import ext
D, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
dev = torch.device("cuda")
flags = torch.zeros(1, 1, D, H, W, device=dev); fluid.emptyDomain(flags)
div = torch.randn(1, 1, D, H, W, device=dev)
a = torch.zeros_like(div); b = torch.zeros_like(div)
ws = torch.empty(ext.jacobi_workspace_bytes(1, D, H, W, True), dtype=torch.uint8, device=dev)
ext.jacobi_pass_(flags, div, None, a, 2, 0, 0, ws, False, layout=2)


def timed(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def chain(n, kb, ke, kb2=-1):
    def f():
        ext.jacobi_pass_(flags, div, a, b, n, kb, ke, ws, True, kb2, layout=3)
        ext.jacobi_pass_(flags, div, b, a, n, kb, ke, ws, True, kb2, layout=3)
    return f


print(f"{D}x{H}x{W}, row-quad layout in and out, back-to-back launches (us per launch | per sweep):")
for what, kb, ke, kb2 in (("whole domain", 0, 0, -1), ("two ranges of 6 planes (an edge part)", 6, 12, D - 12), ("two ranges of 8 planes", 4, 12, D - 12),
                          ("two ranges of 12 planes", 3, 15, D - 16), ("one range of 16 planes", 8, 24, -1)):
    t2 = timed(chain(2, kb, ke, kb2)) / 2
    t3 = timed(chain(3, kb, ke, kb2)) / 2
    print(f"  {what:40s} two-sweep {t2:7.2f} | {t2 / 2:6.2f}    three-sweep {t3:7.2f} | {t3 / 3:6.2f}")
