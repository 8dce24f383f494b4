from .basis import GeoTypeIdentifier
from .box import Box
from .cylinder import Cylinder

__all__ = ["Box", "Cylinder", "GeoTypeIdentifier"]
