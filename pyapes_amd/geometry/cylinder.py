"""``Cylinder`` geometry: the axisymmetric (r, z) domain (``pyapes/geometry/cylinder.py:10-99``)."""
from __future__ import annotations

from math import pi

from .basis import GeoBounder, Geometry, box_faces


class Cylinder(Geometry, metaclass=GeoBounder):
    """``Cylinder([0, 0], [1, 2])`` or ``Cylinder[0:1, 0:2]``: always two-dimensional, the leading
    axis is the radius r >= 0, the second the axis z.  A mesh on it has ``coord_sys == "rz"`` and
    the faces ``rl, ru, zl, zu``."""

    def __init__(self, lower, upper):
        assert len(lower) == 2 and len(upper) == 2, \
            "Cylinder: a length of inputs has to be 2 since it is axisymmetric (r-z)!)"
        assert lower[0] >= 0, "Cylinder: lower bound of radius has to be larger (or equal) to 0!"
        self._lower = [float(v) for v in lower]
        self._upper = [float(v) for v in upper]
        self._dim = 2
        self._config = dict(enumerate(box_faces(self._lower, self._upper, "rz")))
        self.face = [c["face"] for c in self._config.values()]

    @property
    def dim(self) -> int:
        return self._dim

    @property
    def type(self) -> str:
        return "cylinder"

    @property
    def size(self) -> float:
        """pi r^2 z with r, z the extents (cylinder.py:62-74)"""
        return pi * (self._upper[0] - self._lower[0]) ** 2 * (self._upper[1] - self._lower[1])

    @property
    def X(self) -> float:
        return self._lower[0]

    @property
    def Y(self) -> float:
        return self._lower[1]

    @property
    def config(self):
        return self._config

    @property
    def lower(self) -> list[float]:
        return self._lower

    @property
    def upper(self) -> list[float]:
        return self._upper
