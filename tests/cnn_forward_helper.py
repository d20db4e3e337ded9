"""Helper of test_parity_gpu.py::test_cnn_benchmark_size: one MultiScaleNet forward on a seeded input, with the default
kernel selection ("fp32": Winograd where the launch fills the chip) or precision_mode "fp32_direct" (direct implicit-GEMM
kernels only).   python tests/cnn_forward_helper.py D H W seed out.npy [precision_mode]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def make_input(D, H, W, seed):
    """x (1,2,D,H,W): channel 0 ~ N(0,1) (div / std), channel 1 the occupancy of a walled domain with obstacle boxes."""
    from util import make_flags
    rng = np.random.default_rng(seed)
    x = np.empty((1, 2, D, H, W), np.float32)
    x[0, 0] = rng.standard_normal((D, H, W), dtype=np.float32)
    x[0, 1] = (make_flags(1, D, H, W, boxes=True)[0, 0] == 2).astype(np.float32)
    return x


def forward(x, dev="cuda:0", precision_mode="fp32"):
    import torch
    from fluidnet_cxx_amd import FluidNet
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    is3d = x.shape[2] > 1
    mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
                 normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=is3d, precisionMode=precision_mode)
    net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=3 if is3d else 2), dev)
    t = torch.from_numpy(x).to(dev)
    if not is3d:
        t = t[:, :, 0].contiguous()
    p = net.multiScale(t)
    torch.cuda.synchronize()
    return p.cpu().numpy().reshape(1, 1, *x.shape[2:])


if __name__ == "__main__":
    D, H, W, seed = (int(v) for v in sys.argv[1:5])
    np.save(sys.argv[5], forward(make_input(D, H, W, seed), precision_mode=sys.argv[6] if len(sys.argv) > 6 else "fp32"))
