"""Axis-aligned rectangular domains: what ``Box`` and ``Cylinder`` share."""
from __future__ import annotations

from .basis import Geometry, box_faces


class RectDomain(Geometry):
    """Lower / upper corner (stored as floats), the face table in the reference's order and the
    ``X, Y, Z`` accessors of the lower corner.  Subclasses fix the coordinate system, the domain
    ``type`` string the mesh dispatches on, and ``size``."""

    _coord = "xyz"
    _type = "rect"

    def __init__(self, lower, upper):
        self._lower = [float(v) for v in lower]
        self._upper = [float(v) for v in upper]
        self._config = dict(enumerate(box_faces(self._lower, self._upper, self._coord)))
        self.face = [entry["face"] for entry in self._config.values()]

    dim = property(lambda self: len(self._lower))
    type = property(lambda self: self._type)
    lower = property(lambda self: self._lower)
    upper = property(lambda self: self._upper)
    config = property(lambda self: self._config)
    X = property(lambda self: self._lower[0])
    Y = property(lambda self: self._lower[1])
    Z = property(lambda self: self._lower[2])

    def extents(self) -> list[float]:
        return [float(u - l) for l, u in zip(self._lower, self._upper)]
