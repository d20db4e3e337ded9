"""Obstacle geometry written into batch_dict['flags'] in place -- the surface of the reference's
`lib/fluid/geometry_utils.py` (createCylinder :4-34, createBox2D :36-63), built on the tensor's own device."""
import torch

from . import CellType


def _xy(flags):
    assert flags.dim() == 5, "Input flags must have 5 dimensions"
    assert flags.size(0) == 1, "Only batches of size 1 allowed (inference)"
    H, W = flags.size(3), flags.size(4)
    y = torch.arange(H, device=flags.device).view(1, 1, 1, H, 1)
    x = torch.arange(W, device=flags.device).view(1, 1, 1, 1, W)
    return x, y


def createCylinder(batch_dict, centerX, centerY, radius):
    """Marks every cell with (x-centerX)^2 + (y-centerY)^2 <= radius^2 as an obstacle, on all z planes
    (geometry_utils.py:26-33: integer cell indices against the float centre)."""
    assert "flags" in batch_dict, "Error: flags key is not in batch dict"
    flags = batch_dict["flags"]
    x, y = _xy(flags)
    inside = (x - centerX) ** 2 + (y - centerY) ** 2 <= radius * radius
    flags.masked_fill_(inside.expand_as(flags), float(CellType.TypeObstacle))
    batch_dict["flags"] = flags


def createBox2D(batch_dict, x0, x1, y0, y1):
    """Marks the cells x0 <= x < x1, y0 <= y < y1 as obstacles.  The reference's version cannot run (it tests
    `Y >= y1 and Y < y1` and then names an undefined mask, geometry_utils.py:59-62); this is the box its docstring
    describes."""
    assert "flags" in batch_dict, "Error: flags key is not in batch dict"
    flags = batch_dict["flags"]
    x, y = _xy(flags)
    inside = (x >= x0) & (x < x1) & (y >= y0) & (y < y1)
    flags.masked_fill_(inside.expand_as(flags), float(CellType.TypeObstacle))
    batch_dict["flags"] = flags
