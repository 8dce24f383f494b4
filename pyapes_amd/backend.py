"""Device / dtype gate (the roles of ``pyapes/backend.py:7-94``).

``Mesh(..., device, dtype)`` builds these.  On PyTorch-ROCm the GPU device string is
still ``"cuda"``.  A mesh on ``"cpu"`` can be constructed (containers, BC tables and
the equation DSL are host logic) but has no compute path: operators and solvers
raise, they never fall back to torch arithmetic.
"""
from __future__ import annotations

import torch

TORCH_DEVICE = ["cpu", "cuda", "mps"]
DTYPE_SINGLE = ["single", "s", 32]
DTYPE_DOUBLE = ["double", "d", 64]

# precision -> (float, complex, int) torch dtypes
_DTYPES = {
    32: (torch.float32, torch.complex64, torch.int32),
    64: (torch.float64, torch.complex128, torch.int64),
}


class DType:
    """``DType("single" | "double" | "s" | "d" | 32 | 64)`` carries the torch dtypes of a mesh as
    ``.float / .complex / .int / .bool``.  As in the reference (backend.py:28-41), making one also
    makes its float type torch's process-wide default dtype."""

    __slots__ = ("precision", "float", "complex", "int", "bool")

    def __init__(self, precision: str | int = "double"):
        bits = 32 if precision in DTYPE_SINGLE else (64 if precision in DTYPE_DOUBLE else None)
        if bits is None:
            raise ValueError("Invalid precision type!")
        self.precision = precision
        self.float, self.complex, self.int = _DTYPES[bits]
        self.bool = torch.bool
        torch.set_default_dtype(self.float)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, DType) and other.float == self.float

    def __hash__(self) -> int:
        return hash(self.float)

    def __repr__(self) -> str:
        return f"(torch.dtype){self.precision}"


class TorchDevice:
    """``TorchDevice("cuda").device`` is the ``torch.device`` (backend.py:71-94)."""

    __slots__ = ("device_type", "device")

    def __init__(self, device_type: str = "cpu"):
        assert device_type in TORCH_DEVICE
        self.device_type = device_type
        self.device = torch.device(device_type.lower())

    def __repr__(self) -> str:
        return f"Device on {self.device}"


def require_gpu(t: torch.Tensor, what: str) -> None:
    """Every compute entry point calls this: no GPU tensor, no result."""
    if not t.is_cuda:
        raise RuntimeError(
            f"pyapes_amd: {what} needs a mesh on device='cuda' (MI355X); this backend has no CPU "
            "compute path. Use the upstream pyapes torch path for CPU runs.")
