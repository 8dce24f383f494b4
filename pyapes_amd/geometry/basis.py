"""Face / axis naming tables and the ``Geometry`` base (``pyapes/geometry/basis.py``)."""
from __future__ import annotations

from typing import Any

DIR = ["x", "y", "z"]
DIR_TO_NUM: dict[str, int] = {"x": 0, "y": 1, "z": 2}
NUM_TO_DIR: dict[int, str] = {v: k for k, v in DIR_TO_NUM.items()}
DIR_TO_NUM_RZ: dict[str, int] = {"r": 0, "z": 1}
NUM_TO_DIR_RZ: dict[int, str] = {v: k for k, v in DIR_TO_NUM_RZ.items()}
SIDE_TO_NUM: dict[str, int] = {"l": 0, "u": 1}
FDIR = ["xl", "xu", "yl", "yu", "zl", "zu"]
"""Face ids: axis letter + l(ower) / u(pper) (basis.py:16)."""
FDIR_RZ = ["rl", "ru", "zl", "zu"]
"""Face ids of the axisymmetric (r, z) mesh (basis.py:18)."""


def n2d_coord(coord: str) -> dict[int, str]:
    if coord == "xyz":
        return NUM_TO_DIR
    if coord == "rz":
        return NUM_TO_DIR_RZ
    raise RuntimeError("DiffFlux: unknown coordinate system.")


def d2n_coord(coord: str) -> dict[str, int]:
    """axis letter -> mesh axis for a coordinate system"""
    return DIR_TO_NUM_RZ if coord == "rz" else DIR_TO_NUM


def face_list(coord: str) -> list[str]:
    return FDIR_RZ if coord == "rz" else FDIR


class GeoTypeIdentifier(list):
    """``int in GeoTypeIdentifier([1, 2.0])`` is true if any element is an int (basis.py:35-42)."""

    def __contains__(self, typ: type) -> bool:  # type: ignore[override]
        return any(isinstance(v, typ) for v in self)


class Geometry:
    """Interface of a domain geometry."""

    @property
    def dim(self) -> int:
        raise NotImplementedError

    @property
    def type(self) -> str:
        raise NotImplementedError

    @property
    def size(self) -> float:
        raise NotImplementedError

    @property
    def lower(self) -> list[float]:
        raise NotImplementedError

    @property
    def upper(self) -> list[float]:
        raise NotImplementedError

    @property
    def config(self) -> dict[int, dict[str, Any]]:
        raise NotImplementedError

    def __eq__(self, other: Any) -> bool:
        return self.lower == other.lower and self.size == other.size

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(lower={self.lower}, upper={self.upper}, "
                f"size={self.size:.1e})")


class GeoBounder(type):
    """Metaclass giving ``Box[0:1, 0:2]`` (basis.py:98-133)."""

    def __getitem__(cls, item: tuple[slice, ...] | slice):
        if not isinstance(item, (tuple, slice)):
            raise IndexError("GeoBounder: bounds must be a tuple of slices")
        if isinstance(item, slice):
            item = (item,)
        lower, upper = [], []
        for s in item:
            assert isinstance(s, slice)
            assert type(s.start) in (float, int) and type(s.stop) in (float, int)
            assert s.step is None, "GeoBounder: step must be None"
            lower.append(float(s.start))
            upper.append(float(s.stop))
        return cls(lower, upper)


def box_faces(lower: list[float], upper: list[float], coord: str = "xyz") -> list[dict[str, Any]]:
    """Face table of an axis-aligned box: for every face its anchor corner ``x_p``, extent
    ``e_x`` and id.  Same content as ``bound_edge_and_corner`` (basis.py:136-201), including
    its face ORDER (2-D lists the second axis' faces first), derived instead of tabulated."""
    dim = len(lower)
    assert 0 < dim < 4, "Dimensions must be 1, 2 and 3!"
    assert coord in ("xyz", "rz"), "Coordinate must be either xyz or rz!"
    letters = ["r", "z"] if coord == "rz" else DIR
    axes = {1: [0], 2: [1, 0], 3: [0, 1, 2]}[dim]
    faces = []
    for a in axes:
        for side in ("l", "u"):
            xp = list(lower)
            if side == "u":
                xp[a] = upper[a]
            ex = [upper[b] - xp[b] for b in range(dim)]
            ex[a] = 0.0
            faces.append({"e_x": ex, "x_p": xp, "face": letters[a] + side})
    return faces
