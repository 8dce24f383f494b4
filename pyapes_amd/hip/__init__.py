from .lib import PaError, PaReport, PaTerm, load_library
from .context import HipContext, context_for

__all__ = ["PaError", "PaReport", "PaTerm", "load_library", "HipContext", "context_for"]
