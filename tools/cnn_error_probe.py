"""GPU-box probe: error of each precision mode of the MultiScaleNet forward against the CPU oracle (fp32, direct sums), on random inputs of
four shapes: 515 x 509 and 384 x 352 (2D; the second with W % 4 == 0: the 16-byte halo DMA), 16 x 126 x 130 and 8 x 128 x 96 (3D).
'fp32' is the default (Winograd F(4x4) for the 64/128-channel layers since round 6), 'fp32_f2' F(2x2) everywhere (rounds 2-5).
usage: python tools/cnn_error_probe.py [2d|3d]"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import oracle as O
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd.weights import make_scalenet_weights
dev = torch.device('cuda:0')
SHAPES = {"2d": [(1, 2, 1, 515, 509), (1, 2, 1, 384, 352)], "3d": [(1, 2, 16, 126, 130), (1, 2, 8, 128, 96)]}
for case, shape in [(c, sh) for c in (sys.argv[1:] or ["2d", "3d"]) for sh in SHAPES[c]]:
    is3d = case == "3d"
    nd = 3 if is3d else 2
    w = make_scalenet_weights(0, ndim=nd)
    case = f"{case} {'x'.join(str(v) for v in shape[2:] if v > 1)}"
    x = np.random.default_rng(3).standard_normal(shape).astype(np.float32)
    ref = O.multiscale_forward(O.pack_weights(w, nd), x)
    outs = {}
    for mode in ("fp32", "fp32_f2", "fp32_direct", "bf16x6", "bf16x3"):
        mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
                     normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=is3d, precisionMode=mode)
        net = FluidNet.from_weights(mconf, w, dev)
        t = torch.from_numpy(x).to(dev)
        out = net.multiScale(t if is3d else t[:, :, 0].contiguous()).cpu().numpy().reshape(ref.shape)
        outs[mode] = out
        d = np.abs(out.astype(np.float64) - ref).max()
        print(f"{case} {mode:12s}: max|d| vs oracle {d:.3e}, |ref|max {np.abs(ref).max():.3e}, relative {d / np.abs(ref).max():.3e}", flush=True)
    print(f"{case} bf16x6 vs fp32: max|d| {np.abs(outs['bf16x6'].astype(np.float64) - outs['fp32']).max():.3e}; identical: {np.array_equal(outs['bf16x6'], outs['fp32'])}")
