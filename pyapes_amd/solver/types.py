"""Config / op schemas (``pyapes/solver/types.py:11-70``)."""
from __future__ import annotations

from typing import Any, Callable, TypedDict

from torch import Tensor


class DivConfigType(TypedDict, total=False):
    limiter: str   # "none" | "upwind"
    edge: bool
    compat: bool   # new: limiter "upwind" reproduces the reference's literal output


class LaplacianConfigType(TypedDict):
    edge: bool


class GradConfigType(TypedDict):
    edge: bool


class DdtConfigType(TypedDict):
    scheme: str


class DiscretizerConfigType(TypedDict, total=False):
    div: DivConfigType
    laplacian: LaplacianConfigType
    grad: GradConfigType
    ddt: DdtConfigType


class OPStype(TypedDict):
    """One registered operator of an equation (same keys as the reference)."""

    name: str
    Aop: Callable[..., Tensor]
    target: Any
    param: tuple
    sign: float | int
    other: dict[str, float] | None
    A_coeffs: Any
    adjust_rhs: Callable[..., Tensor]
