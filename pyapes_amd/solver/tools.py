"""Solver configs and the matrix-free stand-in for the reference's coefficient tables
(``pyapes/solver/tools.py``)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, TypedDict


class FDMSolverConfig(TypedDict, total=False):
    method: str    # "cg" | "bicgstab" | "jacobi" (new)
    tol: float
    max_it: int
    report: bool
    omega: float   # jacobi relaxation (new, default 1.0)
    save_old: bool  # keep Field.VARo = the iterate before the last iteration, as var.save_old() in the
                    # reference's loops does (linalg.py:110, 210); costs one extra pass per iteration


class SolverConfig(TypedDict):
    fdm: FDMSolverConfig


@dataclass
class StencilSpec:
    """What ``build_A_coeffs`` returns here.

    The reference materialises 5 x mesh.dim coefficient arrays (15 GiB at 512^3 fp64,
    ``tools.py:29-108`` + ``fdc.py:376-423``).  The HIP kernels evaluate the same
    coefficients from predicates on the node index, so all that has to be remembered
    is which operator it is and the BC list (types + order) it was built against.
    """

    op: str                       # "Laplacian" | "Grad" | "Div"
    bcs: list = field(default_factory=list)
    limiter: str = "none"
    compat: bool = False
    var_j: Any = None

    def __len__(self) -> int:  # the reference asserts len(A_coeffs) == 5 (fdc.py:184-186)
        return 5
