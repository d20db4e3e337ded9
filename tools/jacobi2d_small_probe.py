import sys, os, torch, time
sys.path.insert(0, '.')
from fluidnet_cxx_amd import fluid as fl
dev = torch.device('cuda:0')
for res in (128, 192, 256, 384, 512, 768):
    flags = torch.zeros(1, 1, 1, res, res, device=dev); fl.emptyDomain(flags)
    div = torch.randn(1, 1, 1, res, res, device=dev)
    for _ in range(20): fl.solveLinearSystemJacobi(flags, div, False, 0.0, 28)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fl.solveLinearSystemJacobi(flags, div, False, 0.0, 28)
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(res, "OY", os.environ.get("FNX_JACOBI_OY", "auto"), f"{(time.perf_counter() - t) / 200 * 1e6:.1f} us per 28-sweep solve")
