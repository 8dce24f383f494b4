"""Interior slicers (``pyapes/mesh/tools.py:7-32``)."""
from __future__ import annotations

from ..geometry.basis import DIR_TO_NUM, SIDE_TO_NUM


def boundary_slicer(dim: int, bcs: list) -> list[slice]:
    """``[1:-1]`` per axis; a periodic face opens its side of the slice."""
    lim: list[list[int | None]] = [[1, -1] for _ in range(dim)]
    for bc in bcs:
        if bc.bc_type == "periodic":
            lim[DIR_TO_NUM[bc.bc_face[0]]][SIDE_TO_NUM[bc.bc_face[1]]] = None
    return [slice(*l) for l in lim]


def inner_slicer(dim: int, pad: int | None = 1) -> list[slice]:
    return [slice(pad, -pad if isinstance(pad, int) else None) for _ in range(dim)]
