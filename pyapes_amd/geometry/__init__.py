from .basis import GeoTypeIdentifier
from .box import Box

__all__ = ["Box", "GeoTypeIdentifier"]
