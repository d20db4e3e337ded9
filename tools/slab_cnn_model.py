"""GPU-box probe: the CNN-projection step of a MIDDLE rank of the z-slab decomposition through the C++ driver (fnx_slab_step with
prm.method = 1), on one GPU, against an assumed interconnect (the link-model communicator) -- next to the ghost-free single slab.
512 x 512 x 64 planes per rank, halo 52 (the 49 ghost planes of the normalised velocity travel once per step).
usage: slab_cnn_model.py [precision=fp32]      env MODEL_LINKS="0:0,20:75" MODEL_STEPS=3"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd._ext import ext
from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
from fluidnet_cxx_amd.weights import make_scalenet_weights

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(os.environ.get("MODEL_STEPS", 3))
cfgs = ((0, 0), (20, 75), (25, 55))
if os.environ.get("MODEL_LINKS"):
    cfgs = tuple(tuple(int(v) for v in c.split(":")) for c in os.environ["MODEL_LINKS"].split(","))
w = dict(bench.WORKLOADS["plume3d_slab_jacobi"]); m = bench.mconf_for(w)
m.update(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True, normalizeInputChan="UDiv",
         normalizeInputThreshold=1e-5, is3D=True, precisionMode=mode)
net = FluidNet.from_weights(m, make_scalenet_weights(0, ndim=3), dev)


def timed(step):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


l1 = SlabLayout(64, 1, 0, 52)
st = bench.plume_state_torch(512, l1.D_local, dev, 0, 64)
sim = NativeSlabSimulator(l1, m, comm=None, static_flags=True, cfl_check_every=0, method="convnet", net=net)
base = timed(lambda: sim.step(st))
print(f"{mode}: ghost-free slab (1 rank), CNN projection: {base:.2f} ms/step", flush=True)
del sim, st
layout = SlabLayout(64 * 3, 3, 1, 52)
for lat, gbps in cfgs:
    st = bench.plume_state_torch(512, layout.D_local, dev, layout.z_offset, layout.D_global)
    sim = NativeSlabSimulator(layout, m, comm=ext.slab_comm_link_model(float(lat), float(gbps)), static_flags=True, cfl_check_every=0,
                              method="convnet", net=net)
    t = timed(lambda: sim.step(st))
    print(f"{mode}: middle rank, link {gbps:4d} GB/s + {lat:2d} us -> {t:.2f} ms/step   (ghost-free / middle = {base / t * 100:.1f} %)", flush=True)
    del sim, st
