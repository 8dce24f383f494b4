"""``Cylinder`` geometry: the axisymmetric (r, z) domain (``pyapes/geometry/cylinder.py:10-99``)."""
from __future__ import annotations

from math import pi

from .basis import GeoBounder
from .domain import RectDomain


class Cylinder(RectDomain, metaclass=GeoBounder):
    """``Cylinder([0, 0], [1, 2])`` or ``Cylinder[0:1, 0:2]``: always two axes, radius r >= 0 first, then
    z.  A mesh on it has ``coord_sys == "rz"`` and the faces ``rl, ru, zl, zu``; ``size`` is
    pi r^2 z of the extents (cylinder.py:62-74)."""

    _coord, _type = "rz", "cylinder"

    def __init__(self, lower, upper):
        assert len(lower) == 2 and len(upper) == 2, \
            "Cylinder: a length of inputs has to be 2 since it is axisymmetric (r-z)!)"
        assert lower[0] >= 0, "Cylinder: lower bound of radius has to be larger (or equal) to 0!"
        super().__init__(lower, upper)

    @property
    def size(self) -> float:
        dr, dz = self.extents()
        return pi * dr ** 2 * dz
