from enum import IntEnum


class CellType(IntEnum):
    """Mantaflow / FluidNet cell types (reference lib/fluid/cell_type.py:5-14, cpp/cell_type.h:7-18)."""
    TypeNone = 0
    TypeFluid = 1
    TypeObstacle = 2
    TypeEmpty = 4
    TypeInflow = 8
    TypeOutflow = 16
    TypeOpen = 32
    TypeStick = 128
    TypeReserved = 256
