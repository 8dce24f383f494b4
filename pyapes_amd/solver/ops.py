"""``Solver`` facade and the operator sum ``_Aop`` (mirrors ``pyapes/solver/ops.py``)."""
from __future__ import annotations

import torch
from torch import Tensor

from ..variables import Field
from .fdm import Operators
from .linalg import ReportType, solve
from .tools import SolverConfig
from .types import OPStype


class Solver:
    """``Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 1000, "report": True}})``."""

    def __init__(self, config: SolverConfig | None = None):
        self.config = config
        self.var: Field | None = None
        self.rhs: Tensor | None = None
        self.eqs: dict[int, OPStype] = {}

    def set_eq(self, eq: Operators) -> None:
        """Take var / ops / rhs from the DSL object and add every operator's BC adjustment
        to the rhs IN PLACE -- the caller's tensor is modified, as in the reference
        (ops.py:61-77)."""
        self.var = eq.var
        self.eqs = eq.ops
        self.rhs = eq.rhs
        if self.rhs is not None:
            for e in self.eqs:
                op = self.eqs[e]
                if op["name"] == "Div":
                    param = op["param"]
                    assert len(param) == 2
                    self.rhs += op["adjust_rhs"](param[0], self.var, param[1])
                else:
                    self.rhs += op["adjust_rhs"](self.var)
        eq.ops = {}
        eq.rhs = None

    def Aop(self, var: Field) -> Tensor:
        assert self.rhs is not None, "Solver: rhs is missing. Did't you forget to set equation?"
        return _Aop(var, self.eqs)

    def solve(self) -> ReportType:
        assert self.var is not None and self.rhs is not None, \
            "Solver: target variable or rhs is missing. Did't you forget to set equation?"
        assert self.config is not None, "Solver: config is missing!"
        self.report = solve(self.var, self.rhs, _Aop, self.eqs, self.config["fdm"], self.var.mesh)
        return self.report

    def __repr__(self) -> str:
        desc = ""
        for op in self.eqs:
            desc += f"{op} - {self.eqs[op]['name']}, target: {self.eqs[op]['target'].name}, param: {self.eqs[op]['param']}\n"
        return desc + f"{len(self.eqs) + 1} - RHS\n"


def _Aop(target: Field, eqs: dict[int, OPStype]) -> Tensor:
    """``sum_k sign_k * Aop_k(target)`` evaluated by ONE kernel over the registered terms
    (ops.py:122-154); wrap-around values on non-interior nodes included, like the reference."""
    from .linalg import terms_of
    from ..hip.context import context_for
    from ..backend import require_gpu
    require_gpu(target(), "Solver.Aop")
    if target.dim != 1:
        raise NotImplementedError("pyapes_amd: solver equations are scalar (the reference's CG/BiCGSTAB "
                                  "only work for var.dim == 1, SURVEY Q7)")
    ctx = context_for(target.mesh)
    terms, bcs = terms_of(eqs)
    ctx.bind_bcs(target(), bcs, 0)
    ctx.set_terms(terms)
    out = torch.empty_like(target())
    ctx.aop(target()[0], interior_only=False, out=out[0])
    return out
