"""GPU-box probe: two ranks sharing ONE GPU, slab exchange over the nccl (RCCL) backend.
Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/slab_nccl_probe.py"""
import os, sys
import numpy as np, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_slab import CFG, global_state, local_state, check_owned, reference_steps
from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
D, H, W = 32, 20, 70
gs = global_state(D, H, W, seed=3)
layout = SlabLayout(D, world, rank, 6)
sim = SlabSimulator(layout, CFG, sweeps_per_exchange=4)
st = local_state(gs, layout, dev)
for _ in range(2):
    sim.step(st)
torch.cuda.synchronize()
ref = reference_steps(gs, 2)
check_owned(st, ref, layout, "nccl 2 ranks on one GPU")
print(f"rank {rank}: slab exchange over nccl OK, owned planes bit-identical to the single-domain oracle")
dist.destroy_process_group()
