"""Device / dtype gate (mirrors ``pyapes/backend.py:7-94``).

``Mesh(..., device, dtype)`` builds these.  On PyTorch-ROCm the GPU device string is
still ``"cuda"``.  A mesh on ``"cpu"`` can be constructed (containers, BC tables and
the equation DSL are host logic) but has no compute path: operators and solvers
raise, they never fall back to torch arithmetic.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

TORCH_DEVICE = ["cpu", "cuda", "mps"]
DTYPE_SINGLE = ["single", "s", 32]
DTYPE_DOUBLE = ["double", "d", 64]


@dataclass
class DType:
    """``DType("single"|"double")`` -> ``.float/.int/.complex/.bool`` torch dtypes.

    Like the reference (backend.py:28-41) constructing it also makes that
    precision torch's process-wide default dtype.
    """

    precision: str | int = "double"

    def __post_init__(self):
        if self.precision in DTYPE_SINGLE:
            torch.set_default_dtype(torch.float32)
            self._float, self._complex, self._int = torch.float32, torch.complex64, torch.int32
        elif self.precision in DTYPE_DOUBLE:
            torch.set_default_dtype(torch.float64)
            self._float, self._complex, self._int = torch.float64, torch.complex128, torch.int64
        else:
            raise ValueError("Invalid precision type!")
        self._bool = torch.bool

    @property
    def float(self) -> torch.dtype:
        return self._float

    @property
    def int(self) -> torch.dtype:
        return self._int

    @property
    def complex(self) -> torch.dtype:
        return self._complex

    @property
    def bool(self) -> torch.dtype:
        return self._bool

    def __repr__(self) -> str:
        return f"(torch.dtype){self.precision}"


class TorchDevice:
    """``TorchDevice("cuda").device`` -> ``torch.device`` (backend.py:71-94)."""

    def __init__(self, device_type: str = "cpu"):
        assert device_type in TORCH_DEVICE
        self.device_type = device_type
        self._device = torch.device(device_type.lower())

    @property
    def device(self) -> torch.device:
        return self._device

    def __repr__(self) -> str:
        return f"Device on {self.device}"


def require_gpu(t: torch.Tensor, what: str) -> None:
    """Every compute entry point calls this: no GPU tensor, no result."""
    if not t.is_cuda:
        raise RuntimeError(
            f"pyapes_amd: {what} needs a mesh on device='cuda' (MI355X); this backend has no CPU "
            "compute path. Use the upstream pyapes torch path for CPU runs.")
