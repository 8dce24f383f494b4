"""Analytic Poisson inputs for tests/benchmarks: the formulas of
``pyapes/testing/poisson.py:20-87`` (1-D farside node66, 2-D node71, 3-D Shi et al. 2012)."""
from __future__ import annotations

from math import cos, exp, pi

import torch
from torch import Tensor

from ..geometry.basis import FDIR
from ..mesh import Mesh
from ..variables import Field


def poisson_rhs_nd(mesh: Mesh, var: Field) -> Tensor:
    rhs = torch.zeros_like(var())
    if mesh.dim == 1:
        rhs[0] = 1.0 - 2.0 * mesh.X ** 2
    elif mesh.dim == 2:
        rhs[0] = 6.0 * mesh.X * mesh.Y * (1.0 - mesh.Y) - 2.0 * (mesh.X ** 3)
    else:
        rhs[0] = torch.sin(pi * mesh.X) * torch.sin(pi * mesh.Y) * torch.sin(pi * mesh.Z)
    return rhs


def poisson_exact_nd(mesh: Mesh) -> Tensor:
    if mesh.dim == 1:
        return 7.0 / 9.0 - 2.0 / 9.0 * mesh.X + mesh.X ** 2 / 2.0 - mesh.X ** 4 / 6.0
    if mesh.dim == 2:
        return mesh.Y * (1.0 - mesh.Y) * (mesh.X ** 3)
    return -1.0 / (3 * pi ** 2) * torch.sin(pi * mesh.X) * torch.sin(pi * mesh.Y) * torch.sin(pi * mesh.Z)


def poisson_1d_bc(grid, mask, *_) -> Tensor:
    x = grid[0][mask]
    return 7.0 / 9.0 - 2.0 / 9.0 * x + x ** 2 / 2.0 - x ** 4 / 6.0


def poisson_2d_bc(grid, mask, *_) -> Tensor:
    return grid[1][mask] * (1.0 - grid[1][mask]) * (grid[0][mask] ** 3)


def poisson_bcs(dim: int = 3, debug: bool = False) -> list[dict]:
    val = poisson_1d_bc if dim == 1 else (poisson_2d_bc if dim == 2 else 0.0)
    return [{"bc_face": FDIR[i], "bc_type": "dirichlet", "bc_val": 4.44 if debug else val}
            for i in range(dim * 2)]


# -- axisymmetric Poisson problem of the reference's tests/test_solver.py:309-358 --------------------
# u = exp(-z) cos(r) on Cylinder[0:1, 0:1]; rl neumann 0, the other faces dirichlet (exact values)
def _rz_ru(grid, mask, *_) -> Tensor:
    return torch.exp(-grid[1][mask]) * cos(1)


def _rz_zl(grid, mask, *_) -> Tensor:
    return torch.cos(grid[0][mask])


def _rz_zu(grid, mask, *_) -> Tensor:
    return torch.cos(grid[0][mask]) * exp(-1)


def poisson_rz_bcs() -> list[dict]:
    return [{"bc_face": "rl", "bc_type": "neumann", "bc_val": 0.0, "bc_val_opt": None},
            {"bc_face": "ru", "bc_type": "dirichlet", "bc_val": _rz_ru, "bc_val_opt": None},
            {"bc_face": "zl", "bc_type": "dirichlet", "bc_val": _rz_zl, "bc_val_opt": None},
            {"bc_face": "zu", "bc_type": "dirichlet", "bc_val": _rz_zu, "bc_val_opt": None}]


def poisson_rz_rhs(mesh: Mesh, var: Field) -> Tensor:
    rhs = torch.zeros_like(var())
    rhs[0] = -torch.sin(mesh.X) / (mesh.X * torch.exp(mesh.Z))
    axis = mesh.X.eq(0.0)
    rhs[0][axis] = -1.0 / torch.exp(mesh.Z[axis])   # the r -> 0 limit
    return rhs


def poisson_rz_exact(mesh: Mesh) -> Tensor:
    return torch.exp(-mesh.Z) * torch.cos(mesh.X)
