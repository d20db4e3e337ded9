"""GPU debug: 3D MultiScaleNet, oracle on a corner crop vs the full-field GPU result, for growing domains."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cnn_forward_helper import forward, make_input
from oracle import oracle as O
from fluidnet_cxx_amd.weights import make_scalenet_weights
O.build()
blob = O.pack_weights(make_scalenet_weights(0, ndim=3), 3)
for (D, H, W) in ((64, 72, 80), (112, 112, 112), (160, 128, 128), (256, 256, 256)):
    x = make_input(D, H, W, seed=5)
    got = forward(x)
    cz, cy, cx = min(D, 64), min(H, 72), min(W, 80)
    t0 = time.time()
    po = O.multiscale_forward(blob, np.ascontiguousarray(x[:, :, :cz, :cy, :cx]))
    M = 48
    vz, vy, vx = (cz if cz == D else cz - M), (cy if cy == H else cy - M), (cx if cx == W else cx - M)
    a, b = got[:, :, :vz, :vy, :vx], po[:, :, :vz, :vy, :vx]
    d = np.abs(a.astype(np.float64) - b)
    print(f"{D}x{H}x{W}: crop {cz}x{cy}x{cx} valid {vz}x{vy}x{vx}  max|d| {d.max():.3e}  |ref|max {np.abs(b).max():.3e}  (oracle {time.time()-t0:.0f}s)", flush=True)
    if d.max() > 1e-4:
        idx = np.argwhere(d > 1e-4)
        print("   bad cells:", len(idx), "min idx", idx.min(0).tolist(), "max idx", idx.max(0).tolist())
