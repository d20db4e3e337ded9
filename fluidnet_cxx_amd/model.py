"""FluidNet / MultiScaleNet forward on MI355X (reference pytorch/lib/model.py:42-227,
pytorch/lib/multi_scale_net.py:100-127, ScaleNet configuration of convModel_mconf.pth).

Inference only.  Weights come as a torch-style state dict (names `multiScale.convN_4.encode.0.weight` ...)
or the flat blob described in include/fluidnet_hip.h; they are repacked once on the device.
"""
import numpy as np
import torch

from ._ext import ext
from .weights import scalenet_layers


def blob_from_state_dict(sd, ndim=2):
    parts = []
    for L in scalenet_layers(2, ndim):
        for suffix in (".weight", ".bias"):
            v = sd[L["name"] + suffix]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            parts.append(np.ascontiguousarray(v, np.float32).ravel())
    return np.concatenate(parts)


class MultiScaleNet:
    """x (B,2,H,W) or (B,2,D,H,W) -> (B,1,...)   (multi_scale_net.py:118-127)"""

    def __init__(self, state_dict, device="cuda", is3D=False):
        self.is3D = bool(is3D)
        blob = torch.from_numpy(blob_from_state_dict(state_dict, 3 if is3D else 2)).to(device)
        self.packed = ext.scalenet_pack(blob, self.is3D)

    def __call__(self, x):
        return ext.multiscale_forward(self.packed, x.contiguous())


class FluidNet:
    """input_ (B,5|6,D,H,W) = [p, U, flags, density] -> (p, U)   (model.py:76-227)

    Supports the shipped configuration: model='ScaleNet', inputChannels={'div'}, normalizeInput on 'UDiv'."""

    def __init__(self, mconf, state_dict, device="cuda", dropout=False):
        assert mconf.get("model", "ScaleNet") == "ScaleNet", "only the ScaleNet variant is accelerated"
        ic = mconf.get("inputChannels", {"div": True, "pDiv": False, "UDiv": False})
        assert ic.get("div", False) and not ic.get("pDiv", False) and not ic.get("UDiv", False), \
            "inputChannels must be {div} (convModel_mconf.pth)"
        assert mconf.get("normalizeInput", True) and mconf.get("normalizeInputChan", "UDiv") == "UDiv"
        self.mconf = mconf
        self.is3D = bool(mconf.get("is3D", False))
        self.threshold = float(mconf.get("normalizeInputThreshold", 1e-5))
        self.multiScale = MultiScaleNet(state_dict, device, self.is3D)
        self.packed = self.multiScale.packed

    def eval(self):
        return self

    def __call__(self, input_):
        p, U = ext.fluidnet_forward(self.packed, input_.contiguous(), self.threshold)
        return p, U

    forward = __call__
