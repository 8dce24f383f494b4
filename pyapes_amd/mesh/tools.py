"""Interior slicers (``pyapes/mesh/tools.py:7-32``)."""
from __future__ import annotations

from ..geometry.basis import DIR_TO_NUM, SIDE_TO_NUM


def boundary_slicer(dim: int, bcs: list) -> list[slice]:
    """The interior set ``[1:-1]`` per axis, with the lower / upper end opened on the side of every
    periodic face.  The face letter is looked up in the xyz table whatever the mesh is -- as in the
    reference, so a periodic face of an rz mesh raises here (``KeyError`` for ``r``, ``IndexError`` for
    ``z`` -> axis 2 of a 2-D mesh)."""
    opened = {(DIR_TO_NUM[bc.bc_face[0]], SIDE_TO_NUM[bc.bc_face[1]]) for bc in bcs if bc.bc_type == "periodic"}
    if any(axis >= dim for axis, _ in opened):
        raise IndexError("list index out of range")
    return [slice(None if (a, 0) in opened else 1, None if (a, 1) in opened else -1) for a in range(dim)]


def inner_slicer(dim: int, pad: int | None = 1) -> list[slice]:
    """``[pad:-pad]`` on every axis (``pad=None``: everything)."""
    stop = -pad if isinstance(pad, int) else None
    return [slice(pad, stop)] * dim
