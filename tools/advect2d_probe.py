"""GPU-box probe: the 2D advection of a developed plume state through both kernel families (plan 'tiles' / 'cells'), for
rocprofv3 --kernel-trace --stats (tools/gpu_kernel_stats.sh).   python tools/advect2d_probe.py [res=1024]"""
import sys, torch
sys.path.insert(0, ".")
import bench
from fluidnet_cxx_amd import simulate
from fluidnet_cxx_amd._ext import ext
dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = dict(bench.WORKLOADS["plume2d_1024_jacobi"], res=res); m = bench.mconf_for(w)
bd = bench.plume_state_torch(res, 1, dev)
for i in range(60):
    simulate(m, bd, None, "jacobi")
for plan in ("tiles", "cells"):
    for _ in range(20):
        r, u = ext.advect_step(float(m["dt"]), bd["density"], bd["U"], bd["flags"], False, float(m["maccormackStrength"]), plan=plan)
torch.cuda.synchronize()
print("done", float(u.abs().max()))
