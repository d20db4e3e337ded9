"""GPU experiment: Jacobi solve time vs (K, OY) tile parameters. Run on the GPU box."""
import os, subprocess, sys, json
code = r'''
import torch, time, sys
sys.path.insert(0, ".")
from fluidnet_cxx_amd import fluid
res, iters = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda")
flags = torch.zeros(1,1,1,res,res, device=dev); fluid.emptyDomain(flags)
div = torch.randn(1,1,1,res,res, device=dev)
for _ in range(5): fluid.solveLinearSystemJacobi(flags, div, False, 0.0, iters)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fluid.solveLinearSystemJacobi(flags, div, False, 0.0, iters)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1)/50*1e3)
'''
for res, iters in ((128, 28), (1024, 28), (2048, 100)):
    for K in (2, 4, 6, 8):
        for OY in (8, 16, 32):
            env = dict(os.environ, FNX_JACOBI_K=str(K), FNX_JACOBI_OY=str(OY))
            out = subprocess.run([sys.executable, "-c", code, str(res), str(iters)], env=env, capture_output=True, text=True)
            try:
                print(f"res={res} iters={iters} K={K} OY={OY}: {float(out.stdout.strip().splitlines()[-1]):8.1f} us/solve", flush=True)
            except Exception:
                print(res, K, OY, "FAILED", out.stderr[-300:], flush=True)
