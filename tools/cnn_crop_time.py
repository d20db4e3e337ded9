"""GPU-box probe: the MultiScaleNet on a z-slab rank's window (owned + 2 x 48 planes of 512 x 512), every tower on the whole
window against the nested crops of SlabSimulator._convnet_projection, and on the owned planes alone (what a single domain pays
for them).  usage: cnn_crop_time.py [owned planes] [H] [W] [mode]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd.slab import SlabSimulator as S
from fluidnet_cxx_amd.weights import make_scalenet_weights
dev = torch.device('cuda:0')
owned = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
G, MF, MH = S.NET_MARGIN, S.NET_MARGIN_FULL, S.NET_MARGIN_HALF
mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True, normalizeInputChan="UDiv",
             normalizeInputThreshold=1e-5, is3D=True, precisionMode=mode)
net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=3), dev)


def timed(x, trim, reps=3):
    net.multiScale(x, trim); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        net.multiScale(x, trim)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


xo = torch.randn(1, 2, owned, H, W, device=dev)
xw = torch.randn(1, 2, owned + 2 * G, H, W, device=dev)
t_own = timed(xo, None)
t_win = timed(xw, None)
t_nest = timed(xw, [G - MF, G - MF, G - MH, G - MH])
print(f"{mode} {owned} owned planes of {H} x {W}: owned planes alone {t_own:.2f} ms; interior rank, every tower on owned +- {G}: {t_win:.2f} ms "
      f"({t_win / t_own:.2f} x); nested crops (+- {MF} / {MH} / {G}): {t_nest:.2f} ms ({t_nest / t_own:.2f} x)")
