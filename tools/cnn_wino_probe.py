"""GPU box: time the MFMA conv launches of one MultiScaleNet forward (HIP events, FNX_PROF_CONV_MFMA) and print a hash
of the output, so that Winograd kernel variants (FNX_CONV_WINO=2/3, other switches) can be compared for bits and speed:
    FNX_CONV_WINO=3 python tools/cnn_wino_probe.py [res] [depth]"""
import hashlib, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd._ext import ext
from fluidnet_cxx_amd.weights import make_scalenet_weights
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1
is3d = D > 1
dev = torch.device('cuda:0')
w = make_scalenet_weights(0, ndim=3 if is3d else 2)
mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
             normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=is3d)
net = FluidNet.from_weights(mconf, w, dev)
x = torch.from_numpy(np.random.default_rng(3).standard_normal((1, 2, D, res, res)).astype(np.float32)).to(dev)
out = net.multiScale(x); torch.cuda.synchronize()
n = 3 if is3d else 10
ext.profile_enable(True)
for _ in range(n):
    out = net.multiScale(x)
torch.cuda.synchronize()
ms, nl = ext.profile_read(1)
work = ext.profile_read_work(1)
ext.profile_enable(False)
h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"WINO={os.environ.get('FNX_CONV_WINO', 'default')} {D}x{res}x{res}: conv_mfma {ms / n:.3f} ms/forward over {nl // n} launches, "
      f"mfma_util {work / (ms * 1e-3) / 1e12 / 157.3:.3f}, out sha {h}, finite {bool(torch.isfinite(out).all())}")
