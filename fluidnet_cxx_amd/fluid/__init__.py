"""`lib.fluid`-compatible operator surface on MI355X (reference pytorch/lib/fluid/__init__.py:1-14).

Same names, keyword arguments and in-place / return conventions as the reference so that
`simulate.py`-shaped callers run unchanged; every operator is one call into the native
extension `fluidnet_cpp` (hand-written HIP behind the C ABI of include/fluidnet_hip.h).
"""
from enum import IntEnum

from .._ext import ext as _ext

# Mantaflow / FluidNet cell-type codes as the flags grid stores them (fp32 at the boundary).  Values are checked
# against the C ABI's FNX_TYPE_* constants in tests/test_abi.py.
CellType = IntEnum("CellType", dict(TypeNone=0, TypeFluid=1, TypeObstacle=2, TypeEmpty=4, TypeInflow=8, TypeOutflow=16,
                                    TypeOpen=32, TypeStick=128, TypeReserved=256))
Geom = _ext.Geom     # per-call 3D geometry options (ref_quirks, z-slab view, compute window); no reference counterpart
from .ops import (advectScalar, advectVelocity, correctScalar, solveLinearSystemJacobi, velocityDivergence,
                  velocityUpdate, addBuoyancy, addGravity, addViscosity, setWallBcs, setWallBcsStick, flagsToOccupancy, setConstVals,
                  getDx, getCentered)
from .init_conditions import emptyDomain, createPlumeBCs, createRayleighTaylorBCs
from .geometry_utils import createCylinder, createBox2D

__all__ = ["CellType", "Geom", "advectScalar", "advectVelocity", "correctScalar", "solveLinearSystemJacobi",
           "velocityDivergence", "velocityUpdate", "addBuoyancy", "addGravity", "addViscosity", "setWallBcs", "setWallBcsStick", "flagsToOccupancy", "setConstVals",
           "getDx", "getCentered", "emptyDomain", "createPlumeBCs", "createRayleighTaylorBCs", "createCylinder", "createBox2D"]
