from .fields import Field

__all__ = ["Field"]
