import sys, os
sys.path.insert(0, '/root/repo')
import torch, bench
from fluidnet_cxx_amd import simulate
dev = torch.device('cuda:0')
w = bench.WORKLOADS['plume3d_256_jacobi']; m = bench.mconf_for(w); bd = bench.build_state(w, dev)
for _ in range(30): simulate(m, bd, None, 'jacobi')
torch.cuda.synchronize()
for _ in range(10): simulate(m, bd, None, 'jacobi', fused=False)
torch.cuda.synchronize()
