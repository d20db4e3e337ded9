#!/usr/bin/env python3
"""Timeline of one workgroup of conv3_wino4_kernel (a build with -DW4_TIMELINE=<workgroup>: ABFLAGS=-DW4_TIMELINE=1000 tools/ab_libs.sh
build tl, the library copied over fluidnet_cxx_amd/libfluidnet_hip.so): runs the 1024^2 net and prints, for its last F(4x4) launch, per wave, the
s_memtime deltas between the stamps (100 start, 101 prologue done, per stage 10 row 0 | 1 row 1 | 2 vmcnt | 3 barrier | 4 DMA issued |
12 row 2 | 5 row 3 | 6 barrier | 14 row 4 | 15 row 5, 102 loop done, 103 stores issued).  s_memtime ticks at 100 MHz x ... see the
printed total against the kernel's duration."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import fluidnet_cxx_amd  # noqa: E402
from fluidnet_cxx_amd import FluidNet  # noqa: E402
from fluidnet_cxx_amd.weights import make_scalenet_weights  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(fluidnet_cxx_amd.__file__), "libfluidnet_hip.so"))
dev = torch.device("cuda:0")
x = torch.randn((1, 2, 1024, 1024), device=dev)
mconf = dict(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
             normalizeInputChan="UDiv", normalizeInputThreshold=1e-5, is3D=False, precisionMode="fp32")
net = FluidNet.from_weights(mconf, make_scalenet_weights(0, ndim=2), dev)
for _ in range(3):
    net.multiScale(x)            # (the last conv3_wino4_kernel launch of a forward is the full-resolution 128 -> 64 layer: 32 stages)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (8 * 256))()
rc = lib.fnx_debug_w4_timeline(out)
a = np.array(out, dtype=np.uint64).reshape(8, 256)
for wv in range(8):
    ks = (a[wv] >> np.uint64(48)).astype(int)
    ts = (a[wv] & np.uint64((1 << 48) - 1)).astype(np.int64)
    n = int((ts != 0).sum())
    if n == 0:
        continue
    print(f"wave {wv}: {n} stamps, total {ts[n - 1] - ts[0]} ticks")
    line = []
    for i in range(1, n):
        line.append(f"{ks[i]}:{ts[i] - ts[i - 1]}")
    print("  " + " ".join(line[:12]))
    mid = [i for i in range(1, n) if ks[i] == 10]
    if len(mid) > 6:
        i0 = mid[len(mid) // 2]
        print("  mid-kernel stage: " + " ".join(line[i0 - 1:i0 + 9]))
        i0 = mid[len(mid) // 2 + 1]
        print("  next stage:       " + " ".join(line[i0 - 1:i0 + 9]))
    print("  tail: " + " ".join(line[-3:]))
