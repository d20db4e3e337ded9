"""Obstacle geometry written into batch_dict['flags'] in place -- the surface of the reference's
`lib/fluid/geometry_utils.py` (createCylinder :4-34, createBox2D :36-63) as native kernels (fnx_create_cylinder,
fnx_create_box2d): no host round trip, flags bit-exact with the reference's tensor expression."""
from .._ext import ext


def _flags(batch_dict):
    assert "flags" in batch_dict, "Error: flags key is not in batch dict"
    flags = batch_dict["flags"]
    assert flags.dim() == 5, "Input flags must have 5 dimensions"
    assert flags.size(0) == 1, "Only batches of size 1 allowed (inference)"
    return flags


def createCylinder(batch_dict, centerX, centerY, radius):
    """Marks every cell with (x-centerX)^2 + (y-centerY)^2 <= radius^2 as an obstacle, on all z planes
    (geometry_utils.py:26-33: integer cell indices against the float centre)."""
    flags = _flags(batch_dict)
    ext.create_cylinder_(flags, float(centerX), float(centerY), float(radius))
    batch_dict["flags"] = flags


def createBox2D(batch_dict, x0, x1, y0, y1):
    """Marks the cells x0 <= x < x1, y0 <= y < y1 as obstacles.  The reference's version cannot run (it tests
    `Y >= y1 and Y < y1` and then names an undefined mask, geometry_utils.py:59-62); this is the box its docstring
    describes."""
    flags = _flags(batch_dict)
    ext.create_box2d_(flags, float(x0), float(x1), float(y0), float(y1))
    batch_dict["flags"] = flags
