"""GPU-box probe: one 3D MultiScaleNet forward per precision mode at D x H x W (default 256^3), three times each; run under
rocprofv3 --kernel-trace and read with tools/show_conv_trace.py.   python tools/cnn3d_layers_probe.py [mode] [D H W]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from cnn_forward_helper import make_input
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd.weights import make_scalenet_weights
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
D, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 256, 256)
mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
             normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=True, precisionMode=mode)
net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=3), "cuda:0")
t = torch.from_numpy(make_input(D, H, W, 5)).to("cuda:0")
for _ in range(3):
    p = net.multiScale(t)
torch.cuda.synchronize()
print(mode, float(p.abs().max()))
