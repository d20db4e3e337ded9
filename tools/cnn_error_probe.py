import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from oracle import oracle as O
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd.weights import make_scalenet_weights
dev=torch.device('cuda:0')
w=make_scalenet_weights(0)
mconf=dict(model="ScaleNet", inputChannels=dict(div=True,pDiv=False,UDiv=False), normalizeInput=True, normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=False)
net=FluidNet.from_weights(mconf,w,dev)
x=np.random.default_rng(3).standard_normal((1,2,1,515,509)).astype(np.float32)
out=net.multiScale(torch.from_numpy(x).to(dev)).cpu().numpy()
ref=O.multiscale_forward(O.pack_weights(w,2),x)
print("FNX_CONV_WINO=%s: max|d| %.3e, max|ref| %.3e, rel %.3e" % (os.environ.get("FNX_CONV_WINO","2"), np.abs(out-ref).max(), np.abs(ref).max(), np.abs(out-ref).max()/max(1,np.abs(ref).max())))
