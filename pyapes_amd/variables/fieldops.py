"""In-place arithmetic sugar of ``Field`` (the operator overloads of ``pyapes/variables/fields.py:256-337``):
``f + g``, ``f - g``, ``f * g``, ``f / g`` and ``f <<= g`` modify ``f``'s tensor and return ``f``.
Kept apart from the container so that the container reads as what it is; plain torch tensor
bookkeeping, nothing of the stencil path lives here."""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor


def _is_field(obj: Any) -> bool:
    return isinstance(obj, FieldArithmetic)


class FieldArithmetic:
    """Mixin for ``Field``; expects ``self._VAR`` (``(dim, *nx)`` tensor) and ``self.dim``."""

    _VAR: Tensor
    dim: int

    def _per_component(self, values: list, what: str) -> None:
        assert len(values) == self.dim, what
        for i, v in enumerate(values):
            self._VAR[i] += v

    def __add__(self, other: Any):
        if _is_field(other):
            self._VAR += other()
        elif isinstance(other, float):
            self._VAR += other
        elif isinstance(other, list):
            self._per_component(other, "Field: input vector should match with Field dimension!")
        elif isinstance(other, Tensor):
            if other.size(0) == self.dim:       # literal: a same-sized tensor REPLACES the data (fields.py:275-277)
                self._VAR = other
            else:
                for i in range(other.size(0)):
                    self._VAR[i] += other[i]
        else:
            raise TypeError("Field: you can only add Field, float, Tensor, list[int], or list[float]!")
        return self

    def __sub__(self, other: Any):
        if not _is_field(other):
            raise TypeError("Field: you can only subtract Field!")
        self._VAR -= other()
        return self

    def __mul__(self, other: Any):
        if _is_field(other):
            self._VAR *= other()
        elif isinstance(other, (float, int)):
            self._VAR *= other
        else:
            raise TypeError("Field: you can only multiply Field, int, or float!")
        return self

    def __truediv__(self, other: Any):
        if not _is_field(other):
            raise TypeError("Field: you can only divide by Field!")
        positive = other().gt(0.0)              # only where the divisor is positive (fields.py:312-316)
        self._VAR[positive] /= other()[positive]
        return self

    def __ilshift__(self, other: Any):
        if _is_field(other):
            self._VAR = other()
        elif isinstance(other, Tensor):
            self.set_var_tensor(other)          # type: ignore[attr-defined]
        elif isinstance(other, (float, int)):
            self._VAR = torch.zeros_like(self._VAR) + other
        elif isinstance(other, list):
            assert self.dim == len(other), "Field: dimension mismatch!"
            self._VAR = torch.zeros_like(self._VAR)
            self._per_component(other, "Field: dimension mismatch!")
        else:
            raise TypeError("Field: you can only assign Field, Tensor, float, int, or list!")
        return self
