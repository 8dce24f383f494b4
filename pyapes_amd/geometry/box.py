"""``Box`` geometry (``pyapes/geometry/box.py:12-92``)."""
from __future__ import annotations

from math import prod

from .basis import GeoBounder
from .domain import RectDomain


class Box(RectDomain, metaclass=GeoBounder):
    """``Box([0, 0], [1, 2])`` or ``Box[0:1, 0:2]``, 1 to 3 axes; ``size`` is the volume."""

    _coord, _type = "xyz", "box"

    def __init__(self, lower, upper):
        assert len(lower) == len(upper), "Box: length of inputs has to be matched!"
        super().__init__(lower, upper)

    @property
    def size(self) -> float:
        return prod(self.extents())
