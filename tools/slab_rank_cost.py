"""GPU-box probe: compute-side cost of a MIDDLE rank of the z-slab decomposition, on one GPU.

Three slabs of the 512 x 512 x 192 plume (the per-GPU shape of bench.py --gpus 3..8) are advanced in lock-step in one
process; ghost exchanges are served by device copies, so what is timed is the kernel work of the ranks (ghost planes,
edge-first Jacobi passes, plane-range launches) without any interconnect.  Reported per rank and compared with the
single-slab step of `bench.py` (no ghosts)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
dev = torch.device("cuda:0")
w = bench.WORKLOADS["plume3d_slab_jacobi"]; m = bench.mconf_for(w)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wsw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
schedule = sys.argv[3] if len(sys.argv) > 3 else "deep_first"
D = 64 * world
halo = int(sys.argv[4]) if len(sys.argv) > 4 else 6
layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
states = [bench.plume_state_torch(512, l.D_local, dev, l.z_offset, D) for l in layouts]
sims = [SlabSimulator(l, m, sweeps_per_exchange=wsw, schedule=schedule, static_flags=True) for l in layouts]
for _ in range(3): lockstep_step(sims, states)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 8
t0 = time.perf_counter(); e0.record()
for _ in range(n): lockstep_step(sims, states)
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
gpu = e0.elapsed_time(e1) / n
from fluidnet_cxx_amd._ext import ext
ext.profile_enable(True)
for _ in range(2): lockstep_step(sims, states)
torch.cuda.synchronize()
tags = {k: ext.profile_read(v) for k, v in bench.PROF.items()}
ext.profile_enable(False)
print("  per rank and step: " + ", ".join(f"{k} {ms / 2 / world:.3f} ms ({n // 2 // world} launches)" for k, (ms, n) in tags.items() if n))
print(f"world={world} w={wsw} {schedule}: GPU {gpu:.3f} ms per lock-step of {world} slabs = {gpu / world:.3f} ms per rank (host enqueue {(t1 - t0) / n * 1e3:.3f} ms)")
