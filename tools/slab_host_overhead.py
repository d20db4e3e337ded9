"""GPU-box probe: host time to ENQUEUE one slab step vs GPU time to execute it (world=1)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator
dev = torch.device("cuda:0")
w = bench.WORKLOADS["plume3d_slab_jacobi"]; m = bench.mconf_for(w)
layout = SlabLayout(64, 1, 0, 6)
bd = bench.plume_state_torch(512, layout.D_local, dev, 0, 64)
sim = SlabSimulator(layout, m, sweeps_per_exchange=4)
for _ in range(3): sim.step(bd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): sim.step(bd)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {(t1-t0)/10*1e3:.3f} ms/step, total {(t2-t0)/10*1e3:.3f} ms/step")
