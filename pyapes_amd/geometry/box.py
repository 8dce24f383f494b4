"""``Box`` geometry (``pyapes/geometry/box.py:12-92``)."""
from __future__ import annotations

from .basis import GeoBounder, Geometry, box_faces


class Box(Geometry, metaclass=GeoBounder):
    """``Box([0, 0], [1, 2])`` or ``Box[0:1, 0:2]``; bounds are stored as floats."""

    def __init__(self, lower, upper):
        assert len(lower) == len(upper), "Box: length of inputs has to be matched!"
        self._lower = [float(v) for v in lower]
        self._upper = [float(v) for v in upper]
        self._dim = len(self._lower)
        self._config = dict(enumerate(box_faces(self._lower, self._upper)))
        self.face = [c["face"] for c in self._config.values()]

    @property
    def dim(self) -> int:
        return self._dim

    @property
    def type(self) -> str:
        return "box"

    @property
    def size(self) -> float:
        s = 1.0
        for lo, up in zip(self._lower, self._upper):
            s *= float(up - lo)
        return s

    @property
    def X(self) -> float:
        return self._lower[0]

    @property
    def Y(self) -> float:
        return self._lower[1]

    @property
    def Z(self) -> float:
        return self._lower[2]

    @property
    def config(self):
        return self._config

    @property
    def lower(self) -> list[float]:
        return self._lower

    @property
    def upper(self) -> list[float]:
        return self._upper
