import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from fluidnet_cxx_amd import FluidNet
from fluidnet_cxx_amd._ext import ext
from fluidnet_cxx_amd.weights import make_scalenet_weights
dev = torch.device('cuda:0')
PROF = dict(conv_mfma=1, conv_direct=4, conv_mfma16=5, conv_bf16=6)
MODES = [m for m in sys.argv[1:] if m in ("fp32", "bf16x6", "bf16x3", "fp32_direct", "fp32_f4", "fp32_f2")] or ["fp32", "bf16x6", "bf16x3"]
for case in [c for c in sys.argv[1:] if c in ("2d", "3d")] or ["2d", "3d"]:
    is3d = case == "3d"
    shape = (1, 2, 256, 256, 256) if is3d else (1, 2, 1024, 1024)
    x = torch.randn(shape, device=dev)
    for mode in MODES:
        mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
                     normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=is3d, precisionMode=mode)
        net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=3 if is3d else 2), dev)
        for _ in range(2): net.multiScale(x)
        torch.cuda.synchronize()
        n = 3 if is3d else 20
        t0 = time.perf_counter()
        for _ in range(n): net.multiScale(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ext.profile_enable(True)
        npf = 2 if is3d else 10
        for _ in range(npf): net.multiScale(x)
        torch.cuda.synchronize()
        tm = {k: ext.profile_read(v) for k, v in PROF.items()}
        wk = {k: ext.profile_read_work(v) for k, v in PROF.items()}
        ext.profile_enable(False)
        txt = ", ".join(f"{k} {t / npf:.3f} ms/{c // npf}" for k, (t, c) in tm.items() if c)
        util = wk["conv_bf16"] / (tm["conv_bf16"][0] * 1e-3) / 2.5e15 if tm["conv_bf16"][1] else 0
        print(f"{case} {mode}: forward {ms:.3f} ms; {txt}; bf16 MFMA util {util:.3f}", flush=True)
        del net
        torch.cuda.empty_cache()
