import sys, torch
sys.path.insert(0, ".")
import bench
from fluidnet_cxx_amd import simulate
from fluidnet_cxx_amd._ext import ext
dev = torch.device("cuda:0")
w = bench.WORKLOADS["plume3d_slab_jacobi"]; m = bench.mconf_for(w); m["jacobiIter"] = 4
bd = bench.plume_state_torch(512, 64, dev)
ws = torch.empty(ext.step_workspace_bytes(1, 64, 512, 512, True), dtype=torch.uint8, device=dev)
m2 = dict(m); m2["jacobiIter"] = 100
for i in range(40):
    simulate(m2 if i < 30 else m, bd, None, "jacobi", workspace=ws, static_flags=(0, 3, 7)[min(i, 2)])
torch.cuda.synchronize()
print("done", float(bd["U"].abs().max()))
