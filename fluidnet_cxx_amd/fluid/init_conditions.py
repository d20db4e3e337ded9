"""Input generators (reference lib/fluid/util.py:5-47, lib/fluid/init_conditions.py:4-127).

Setup-time code, not on the per-step hot path: emptyDomain is a native kernel, the BC masks are built with
a handful of device-side torch ops.  Flag grids are bit-exact with the reference's generators.
"""
import math

import torch

from .._ext import ext


def emptyDomain(flags, boundary_width=1):
    """util.py:5-47: border of width `boundary_width` -> obstacle (2), interior -> fluid (1). In place."""
    assert boundary_width > 0, "Boundary width must be greater than zero!"
    assert flags.dim() == 5, "Flags tensor should be 5D"
    assert flags.size(1) == 1, "Flags should have only one channels (scalar field)"
    ext.empty_domain_(flags, int(boundary_width))


def createPlumeBCs(batch_dict, density_val, u_scale, rad):
    """init_conditions.py:4-83.  Inlet on rows y = 0..3: inside |x - W//2| <= floor(W*rad): U = (0, u_scale[,0]),
    density = density_val; UBCInvMask is 0 on ALL of rows 0..3; densityBCInvMask 0 only inside the inlet.
    3D (the reference has only a TODO, :58): the inlet is the disc (x-cx)^2 + (z-cz)^2 <= r^2 on the y = 0..3 slabs."""
    assert len(batch_dict) == 4, "Batch must contain 4 tensors (p, UDiv, flags, density)"
    U = batch_dict["U"]
    density = batch_dict["density"]
    assert U.dim() == 5 and U.size(0) == 1, "Only single batches allowed (inference)"
    xdim, ydim, zdim = U.size(4), U.size(3), U.size(2)
    is3D = U.size(1) == 3
    dev = U.device
    centerX = xdim // 2
    plumeRad = math.floor(xdim * rad)
    UBC = torch.zeros_like(U)
    UBCInvMask = torch.ones_like(U)
    densityBC = torch.zeros_like(density)
    densityBCInvMask = torch.ones_like(density)
    x = torch.arange(xdim, device=dev).view(1, 1, xdim) - centerX
    r2 = x.pow(2)
    if is3D:
        z = torch.arange(zdim, device=dev).view(zdim, 1, 1) - zdim // 2
        r2 = r2 + z.pow(2)
    inside = (r2 <= plumeRad * plumeRad).expand(zdim, 4, xdim)          # (D, 4, W)
    UBC[0, 1, :, 0:4] = inside.to(U.dtype) * float(u_scale)
    UBCInvMask[:, :, :, 0:4] = 0
    densityBC[0, 0, :, 0:4] = inside.to(U.dtype) * float(density_val)
    densityBCInvMask[0, 0, :, 0:4] = (~inside).to(U.dtype)
    batch_dict["UBC"] = UBC
    batch_dict["UBCInvMask"] = UBCInvMask
    batch_dict["densityBC"] = densityBC
    batch_dict["densityBCInvMask"] = densityBCInvMask


def createRayleighTaylorBCs(batch_dict, mconf, rho1, rho2):
    """init_conditions.py:88-127: tanh density interface with a cosine perturbation."""
    assert len(batch_dict) == 4, "Batch must contain 4 tensors (p, UDiv, flags, density)"
    U = batch_dict["U"]
    resX, resY = U.size(4), U.size(3)
    dev = U.device
    X = torch.arange(0, resX, device=dev).view(1, resX).expand(resY, resX)
    Y = torch.arange(0, resY, device=dev).view(resY, 1).expand(resY, resX)
    thick, ampl, h = mconf["perturbThickness"], mconf["perturbAmplitude"], mconf["height"]
    density = 0.5 * (rho2 + rho1 + (rho2 - rho1) * torch.tanh(thick * (Y / resY - (h + ampl * torch.cos(2 * math.pi * (X / resX))))))
    batch_dict["density"] = density.to(U.dtype).view(1, 1, 1, resY, resX).contiguous()
