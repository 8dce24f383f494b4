"""Equation DSL (mirrors ``pyapes/solver/fdm.py``).

``solver.set_eq(fdm.laplacian(1.0, var) == rhs)``; ``A + B``, ``A - B`` and ``-A``
register further operators / flip signs exactly as the reference does.  Each operator
call records an ``OPStype`` dict; nothing is discretised into arrays (matrix-free).
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from ..variables import Field
from .fdc import FDC
from .types import DiscretizerConfigType, OPStype


class Operators:
    """Base of the DSL operators (fdm.py:27-105)."""

    def __init__(self):
        self._ops: dict[int, OPStype] = {}
        self._rhs: Tensor | None = None
        self._config: DiscretizerConfigType | None = None

    @property
    def ops(self) -> dict[int, OPStype]:
        return self._ops

    @ops.setter
    def ops(self, other: dict) -> None:
        self._ops = other

    @property
    def rhs(self) -> Tensor | None:
        return self._rhs

    @rhs.setter
    def rhs(self, other: Tensor | None) -> None:
        self._rhs = other

    @property
    def var(self) -> Field:
        raise NotImplementedError

    def update_config(self, config: DiscretizerConfigType) -> None:
        self._config = config

    @property
    def config(self) -> DiscretizerConfigType | None:
        return self._config

    def __eq__(self, other: Any) -> "Operators":  # type: ignore[override]
        if isinstance(other, Tensor):
            self._rhs = other
        elif isinstance(other, Field):
            self._rhs = other()
        else:
            self._rhs = torch.zeros_like(self.var()) + other
        assert self._rhs.shape == self.var().shape, \
            f"FDM Operators: RHS shape {self._rhs.shape} does not match {self.var().shape}!"
        return self

    __hash__ = None  # type: ignore[assignment]

    def __add__(self, other: "Operators") -> "Operators":
        self._ops[list(self._ops.keys())[-1] + 1] = other.ops[0]
        return self

    def __sub__(self, other: "Operators") -> "Operators":
        other.ops[0]["sign"] = -1
        self._ops[list(self._ops.keys())[-1] + 1] = other.ops[0]
        return self

    def __neg__(self) -> "Operators":
        self._ops[0]["sign"] = -1
        return self


class Laplacian(Operators):
    """``laplacian(coeff, var)`` or ``laplacian(var)`` (fdm.py:108-169)."""

    def __call__(self, *inputs: Any) -> "Laplacian":
        if len(inputs) == 2:
            assert isinstance(inputs[0], (int, float, Tensor)), \
                "FDM Laplacian: if additional parameter is provided, it must be a float or Tensor!"
            coeffs = float(inputs[0]) if isinstance(inputs[0], int) else inputs[0]
            var = inputs[1]
        elif len(inputs) == 1:
            coeffs, var = None, inputs[0]
        else:
            raise TypeError("FDM: invalid input type!")
        fdc = FDC({"laplacian": {"edge": False}})
        self._var = var
        self._ops[0] = {"name": "Laplacian", "Aop": self.Aop, "target": var, "param": (coeffs,),
                        "sign": 1.0, "other": None, "A_coeffs": fdc.laplacian.build_A_coeffs(var),
                        "adjust_rhs": fdc.laplacian.adjust_rhs}
        return self

    @property
    def var(self) -> Field:
        return self._var

    @staticmethod
    def Aop(param: float | Tensor | None, var: Field, A_coeffs: Any) -> Tensor:
        out = FDC({"laplacian": {"edge": False}}).laplacian.apply(A_coeffs, var)
        return out if param is None else out * param


class Grad(Operators):
    """``grad(var)`` or ``grad(coeff, var)`` (fdm.py:172-230)."""

    def __call__(self, *inputs: Any) -> "Grad":
        if len(inputs) == 2:
            assert isinstance(inputs[0], (float, Tensor)), \
                "FDM Grad: if additional parameter is provided, it must be a float or Tensor!"
            coeffs, var = inputs
        elif len(inputs) == 1:
            assert isinstance(inputs[0], Field), "FDM Grad: invalid input type! Input must be a Field."
            coeffs, var = None, inputs[0]
        else:
            raise TypeError("FDM: invalid input type!")
        fdc = FDC({"grad": {"edge": False}})
        self._var = var
        self._ops[0] = {"name": "Grad", "Aop": self.Aop, "target": var, "param": (coeffs,),
                        "sign": 1.0, "other": None, "A_coeffs": fdc.grad.build_A_coeffs(var),
                        "adjust_rhs": fdc.grad.adjust_rhs}
        return self

    @property
    def var(self) -> Field:
        return self._var

    @staticmethod
    def Aop(param: float | Tensor | None, var: Field, A_coeffs: Any) -> Tensor:
        out = FDC({"grad": {"edge": False}}).grad.apply(A_coeffs, var)
        return out if param is None else out * param


class Div(Operators):
    """``div(var_j, var_i)`` / ``div(var_i)``; needs ``FDM(config)`` with a ``div`` entry
    (fdm.py:233-312)."""

    def __call__(self, *inputs: Any) -> "Div":
        if len(inputs) == 2:
            assert isinstance(inputs[0], (float, Tensor, Field)), \
                "FDM Grad: if additional parameter is provided, it must be a float or Tensor or Field!"
            var_j, var_i = inputs
        elif len(inputs) == 1:
            var_j, var_i = 1.0, inputs[0]
        else:
            raise TypeError("FDM: invalid input type!")
        assert isinstance(var_i, Field), "FDM Div: var_i must be a Field!"
        assert self.config is not None, "FDM Div: config must be provided!"
        self._var_j, self._var_i = var_j, var_i
        fdc = FDC(self.config)
        self._ops[0] = {"name": "Div", "Aop": self.Aop, "target": var_i, "param": (var_j, self.config),
                        "sign": 1.0, "other": None,
                        "A_coeffs": fdc.div.build_A_coeffs(var_j, var_i, self.config),
                        "adjust_rhs": fdc.div.adjust_rhs}
        return self

    @property
    def var(self) -> Field:
        return self._var_i

    @staticmethod
    def Aop(var_j: Any, config: DiscretizerConfigType, var_i: Field, A_coeffs: Any) -> Tensor:
        fdc = FDC(config)
        if not isinstance(var_j, (Tensor, float)):
            A_coeffs = fdc.div.build_A_coeffs(var_j, var_i, config)  # Field advection: refreshed per call
        return fdc.div.apply(A_coeffs, var_i)


class FDM:
    """``FDM(config).laplacian / .grad / .div`` (fdm.py:356-407).  Operator objects are per
    instance.  ``ddt`` does not exist in the reference either; explicit
    time marching is ``pyapes_amd.solver.march.euler_step``."""

    def __init__(self, config: DiscretizerConfigType | None = None) -> None:
        self.laplacian = Laplacian()
        self.grad = Grad()
        self.div = Div()
        if config is not None:
            self.config = config
            self.div.update_config(config)
