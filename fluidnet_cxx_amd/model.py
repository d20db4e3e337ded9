"""FluidNet / MultiScaleNet forward on MI355X (reference pytorch/lib/model.py:42-227,
pytorch/lib/multi_scale_net.py:100-127, ScaleNet configuration of convModel_mconf.pth).

Inference only.  `FluidNet` is constructed and loaded the way the reference drivers do it (plume.py:119-123):

    net = FluidNet(mconf, dropout=False)
    net = net.cuda()
    net.load_state_dict(state['state_dict'])
    net.eval()
    p, U = net(torch.cat((p, U, flags, density), 1))

Weights are torch-style state-dict entries (`multiScale.convN_4.encode.0.weight` ...).  They are repacked for the MFMA
kernels lazily, on the first forward on a given device (and again after `load_state_dict` / `.to()`).
"""
from collections import OrderedDict

import numpy as np
import torch

from ._ext import ext
from .weights import make_scalenet_weights, scalenet_layers

# Parameters of the reference FluidNet that its ScaleNet forward never reads (model.py:59-72: conv1, convBank, conv2,
# conv3, convOut are only used by the 'FluidNet' variant).  A checkpoint of the reference carries them; load_state_dict
# keeps them verbatim so that state_dict() round-trips.
_UNUSED_PREFIXES = ("conv1.", "convBank.", "conv2.", "conv3.", "convOut.", "scale.")


def _layer_keys(ndim):
    return [L["name"] + sfx for L in scalenet_layers(2, ndim) for sfx in (".weight", ".bias")]


def blob_from_state_dict(sd, ndim=2):
    parts = []
    for k in _layer_keys(ndim):
        v = sd[k]
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        parts.append(np.ascontiguousarray(v, np.float32).ravel())
    return np.concatenate(parts)


class MultiScaleNet:
    """x (B,2,H,W) or (B,2,D,H,W) -> (B,1,...)   (multi_scale_net.py:118-127)"""

    def __init__(self, state_dict, device="cuda", is3D=False, precision_mode="fp32"):
        self.is3D = bool(is3D)
        # "fp32": exact-fp32 MFMA arithmetic, 3x3 layers in the Winograd domain where the launch fills the chip;
        # "fp32_direct": every convolution a direct sum over its taps (include/fluidnet_hip.h: FNX_PRECISION_*)
        # "bf16x6" (opt-in): the 64/128-output-channel Winograd layers as six bf16 MFMA products per fp32 product
        # "bf16x3" (opt-in): the same layers with the three products without a low piece (tolerance 1e-4 |ref|max)
        # "fp32_f4": the 64/128-output-channel 3x3(x3) layers in the Winograd F(4x4,3x3) domain, exact-fp32 MFMAs -- what "fp32" runs since
        # round 6; "fp32_f2": F(2x2) for every Winograd layer, the default of rounds 2-5
        self.precision_mode = precision_mode
        blob = torch.from_numpy(blob_from_state_dict(state_dict, 3 if is3D else 2)).to(device)
        self.packed = ext.scalenet_pack(blob, self.is3D)

    def __call__(self, x, trim=None):
        """trim (3D, the z-slab driver): nested z-crops -- [full-tower low, high, half-tower low, high] full-resolution planes
        (include/fluidnet_hip.h: fnx_multiscale_forward_crop); the result then holds planes [trim[0], D - trim[1])"""
        return ext.multiscale_forward(self.packed, x.contiguous(), self.precision_mode, list(trim) if trim is not None else [])


class FluidNet:
    """input_ (B,5|6,D,H,W) = [p, U, flags, density] -> (p, U)   (model.py:76-227)

    `FluidNet(mconf, dropout=True)` like the reference (model.py:45); `dropout` only exists for signature
    compatibility -- every Dropout layer is the identity in eval mode, which is the only mode here (simulate.py:140).
    Supports the shipped configuration: model='ScaleNet', inputChannels={'div'}, normalizeInput on 'UDiv'.
    A freshly constructed net holds deterministic random-init weights of the reference architecture
    (weights.make_scalenet_weights(0)); `load_state_dict` replaces them."""

    def __init__(self, mconf, dropout=True):
        assert mconf.get("model", "ScaleNet") == "ScaleNet", "only the ScaleNet variant is accelerated"
        ic = mconf.get("inputChannels", {"div": True, "pDiv": False, "UDiv": False})
        assert ic.get("div", False) and not ic.get("pDiv", False) and not ic.get("UDiv", False), \
            "inputChannels must be {div} (convModel_mconf.pth)"
        assert mconf.get("normalizeInput", True) and mconf.get("normalizeInputChan", "UDiv") == "UDiv"
        self.mconf = mconf
        self.dropout = dropout
        self.inDims = mconf.get("inputDim", 2)
        self.is3D = bool(mconf.get("is3D", False))
        self.threshold = float(mconf.get("normalizeInputThreshold", 1e-5))
        # not a reference key: "fp32" (default), "fp32_direct" (no Winograd), "bf16x6" or "bf16x3" (opt-in), see MultiScaleNet
        self.precision_mode = str(mconf.get("precisionMode", "fp32"))
        self.training = False
        self._ndim = 3 if self.is3D else 2
        self._params = OrderedDict((k, torch.from_numpy(v)) for k, v in make_scalenet_weights(0, ndim=self._ndim).items())
        self._extra = OrderedDict()          # the reference's unused parameters, kept for state_dict()
        self._device = None                  # set by cuda()/to(); else the device of the first input
        self._ms = None                      # MultiScaleNet packed for `_ms_device`
        self._ms_device = None

    # ---- construction helpers -----------------------------------------------------------------------------------------
    @classmethod
    def from_weights(cls, mconf, weights, device="cuda", dropout=False):
        """Shortcut: a net holding `weights` (name -> array/tensor), packed on `device`."""
        net = cls(mconf, dropout)
        net.load_state_dict(weights)
        return net.to(device)

    # ---- the nn.Module surface the drivers use ------------------------------------------------------------------------
    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def to(self, device=None, *_, **__):
        if device is not None:
            self._device = torch.device(device)
            self._ensure_packed(self._device)
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        assert not mode, "fluidnet_cxx_amd.FluidNet is inference-only (no backward pass)"
        return self

    def state_dict(self):
        sd = OrderedDict()
        for k, v in self._extra.items():
            sd[k] = v
        for k, v in self._params.items():
            sd[k] = v.to(self._device) if self._device is not None else v
        return sd

    def load_state_dict(self, state_dict, strict=True):
        """Like nn.Module.load_state_dict: every multiScale.* parameter must be present with the right shape; the
        reference FluidNet's parameters that the ScaleNet forward never reads are accepted and kept."""
        want = _layer_keys(self._ndim)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._params and not k.startswith(_UNUSED_PREFIXES)]
        if missing or (strict and unexpected):
            msg = "Error(s) in loading state_dict for FluidNet:"
            if missing:
                msg += "\n\tMissing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + "."
            if strict and unexpected:
                msg += "\n\tUnexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + "."
            raise RuntimeError(msg)
        new = OrderedDict()
        for k in want:
            v = state_dict[k]
            v = v.detach().to("cpu", torch.float32) if torch.is_tensor(v) else torch.from_numpy(np.asarray(v, np.float32))
            if tuple(v.shape) != tuple(self._params[k].shape):
                raise RuntimeError(f"Error(s) in loading state_dict for FluidNet:\n\tsize mismatch for {k}: copying a param "
                                   f"with shape {tuple(v.shape)} from checkpoint, the shape in current model is "
                                   f"{tuple(self._params[k].shape)}.")
            new[k] = v.contiguous().clone()
        self._params = new
        self._extra = OrderedDict((k, v) for k, v in state_dict.items() if k.startswith(_UNUSED_PREFIXES))
        self._ms = None                      # repacked on the next forward / to()
        if self._device is not None:
            self._ensure_packed(self._device)

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _ensure_packed(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._ms is None or self._ms_device != device:
            self._ms = MultiScaleNet(self._params, device, self.is3D, self.precision_mode)
            self._ms_device = device
        return self._ms

    @property
    def multiScale(self):
        return self._ensure_packed(self._device if self._device is not None else "cuda")

    @property
    def packed(self):
        return self.multiScale.packed

    def packed_for(self, device):
        """The packed weight blob on `device` (what fnx_simulate_step takes as FnxState.net)."""
        return self._ensure_packed(device).packed

    def __call__(self, input_):
        ms = self._ensure_packed(input_.device)
        p, U = ext.fluidnet_forward(ms.packed, input_.contiguous(), self.threshold, self.precision_mode)
        return p, U

    forward = __call__
