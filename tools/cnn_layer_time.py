"""GPU-box probe: one 3x3(x3) MFMA layer of the MultiScaleNet alone, per precision mode: time per launch.
usage: cnn_layer_time.py [2d|3d] [cin] [cout] [reps]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd._ext import ext
from fluidnet_cxx_amd.weights import make_scalenet_weights
dev = torch.device('cuda:0')
case = sys.argv[1] if len(sys.argv) > 1 else "3d"
is3d = case == "3d"
shape = (1, 2, 256, 256, 256) if is3d else (1, 2, 1024, 1024)
x = torch.randn(shape, device=dev)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else (2 if is3d else 10)
for mode in ("fp32", "bf16x6"):
    mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
                 normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=is3d, precisionMode=mode)
    net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=3 if is3d else 2), dev)
    net.multiScale(x); torch.cuda.synchronize()
    for _ in range(reps): net.multiScale(x)
    torch.cuda.synchronize()
    print(mode, "done", flush=True)
