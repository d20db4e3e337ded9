"""`lib.fluid`-compatible operator surface on MI355X (reference pytorch/lib/fluid/__init__.py:1-14).

Same names, keyword arguments and in-place / return conventions as the reference so that
`simulate.py`-shaped callers run unchanged; every operator is one call into the native
extension `fluidnet_cpp` (hand-written HIP behind the C ABI of include/fluidnet_hip.h).
"""
from .cell_type import CellType
from .ops import (advectScalar, advectVelocity, correctScalar, solveLinearSystemJacobi, velocityDivergence,
                  velocityUpdate, addBuoyancy, setWallBcs, flagsToOccupancy, setConstVals, getDx)
from .init_conditions import emptyDomain, createPlumeBCs, createRayleighTaylorBCs

__all__ = ["CellType", "advectScalar", "advectVelocity", "correctScalar", "solveLinearSystemJacobi",
           "velocityDivergence", "velocityUpdate", "addBuoyancy", "setWallBcs", "flagsToOccupancy", "setConstVals",
           "getDx", "emptyDomain", "createPlumeBCs", "createRayleighTaylorBCs"]
