"""Host-stepped solver loops for BC callables that READ THE ITERATE (the slow path of SURVEY 8b).

The reference evaluates a callable ``bc_val(grid, mask, var, bc_val_opt)`` inside EVERY BC fill, face after face,
with the field as it is at that moment -- the faces earlier in the list already filled (``bcs.py:200-213``,
``223-253``, called from ``linalg.py:122-125`` / ``243-261``).  The device loops (``pa_cg`` ...) evaluate callables
once per solve, which is the same thing only for callables of ``(grid, mask)``.  For one that reads ``var`` -- a
Robin-type condition, a value tied to the neighbouring interior node -- the loop has to come back to Python once
per face and iteration; this module is that loop: ``linalg.cg`` / ``linalg.bicgstab`` statement for statement
(``linalg.py:74-159``, ``162-279``), with

  * the operator application ``A d`` on the interior set: ``pa_aop(interior_only=1)`` (the tiled / generic HIP
    stencil kernels of the fast path),
  * the BC fill: one ``pa_apply_bc`` per face in list order, its callable evaluated immediately before with the
    current field -- exactly ``BC.apply``,
  * AXPYs, dot products and the stop test as the reference writes them (torch on the GPU).

It is slow by construction (a dozen launches and two host round trips per iteration) and only chosen when
``BC.depends_on_var`` says a callable needs it.  Jacobi has no counterpart in the reference and stays refused.
"""
from __future__ import annotations

import warnings
from typing import Any

import torch
from torch import Tensor

from ..mesh.tools import boundary_slicer


def _nan_to_num(t: Tensor) -> Tensor:
    return torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0)   # linalg.py:302-305


def _tolerance(a: Tensor, b: Tensor) -> float:
    """linalg.py:321-338 (scalar fields: one component)"""
    tol = torch.linalg.norm(a - b)
    v = float(tol)
    if v != v or v in (float("inf"), float("-inf")):
        raise RuntimeError(f"Invalid tolerance detected! tol: {v}")
    return v


class _Ops:
    def __init__(self, var: Any, terms: list[dict], ctx: Any):
        self.var, self.ctx, self.bcs = var, ctx, list(var.bcs)
        ctx.set_terms(terms)
        self.S = (0, *boundary_slicer(var.mesh.dim, var.bcs))
        self._bound = False

    def fill(self, x: Tensor) -> None:
        """``_apply_bc_otf`` (linalg.py:282-299): face after face, the callable sees the partly filled field"""
        for bc in self.bcs:
            self.ctx.apply_bcs(x, [bc], comps=[0])
        self._bound = False

    def A(self, d: Tensor) -> Tensor:
        """(A d) on the interior set, 0 elsewhere -- the stencil rows only need the faces' TYPES"""
        if not self._bound:
            self.ctx.bind_bcs(d, self.bcs, 0, types_only=True)
            self._bound = True
        return self.ctx.aop(d[0], interior_only=True).unsqueeze(0)


def cg(var: Any, rhs: Tensor, terms: list[dict], ctx: Any, tolerance: float, max_it: int) -> dict:
    """linalg.py:74-159"""
    op = _Ops(var, terms, ctx)
    S = op.S
    x = var()
    tol, itr = 1.0, 0
    op.fill(x)
    r = torch.zeros_like(x)
    r[S] = rhs[S] - op.A(x)[S]
    d = r.clone()
    while tol > tolerance:
        x_old = x.clone()
        Ad = op.A(d)
        alpha = _nan_to_num(torch.sum(r * r) / torch.sum(d * Ad))
        x = x + alpha * d
        op.fill(x)
        beta_denom = torch.sum(r * r)
        r -= alpha * Ad
        tol = _tolerance(x, x_old)
        beta = torch.sum(r * r) / beta_denom
        d = r + beta * d
        itr += 1
        if itr > max_it:
            break
    var.set_var_tensor(x)
    var.VARo = x_old if itr else x.clone()
    return {"itr": itr, "tol": tol}


def bicgstab(var: Any, rhs: Tensor, terms: list[dict], ctx: Any, tolerance: float, max_it: int) -> dict:
    """linalg.py:162-279"""
    op = _Ops(var, terms, ctx)
    S = op.S
    x = var()
    itr = 0
    op.fill(x)
    r0 = torch.zeros_like(x)
    r0[S] = rhs[S] - op.A(x)[S]
    r = r0.clone()
    v = torch.zeros_like(x)
    p = torch.zeros_like(x)
    rho: Any = 1.0
    alpha: Any = 1.0
    omega: Any = 1.0
    rho_next = torch.sum(r0 * r0)
    tol = float(torch.sqrt(rho_next))
    x_old = x.clone()
    finished = False
    while not finished:
        x_old = x.clone()
        beta = rho_next / rho * alpha / omega
        rho = rho_next
        p = r + beta * (p - omega * v)
        v = op.A(p)
        itr += 1
        alpha = _nan_to_num(rho / torch.sum(r0 * v))
        s = r - alpha * v
        tol = _tolerance(r, alpha * v)
        if tol <= tolerance:
            x = x + alpha * p
            op.fill(x)
            finished = True
            continue
        t = op.A(s)
        omega = _nan_to_num(torch.sum(t * s) / torch.sum(t * t))
        rho_next = -omega * torch.sum(r0 * t)
        x = x + alpha * p + s * omega
        op.fill(x)
        r = s - omega * t
        tol = _tolerance(s, omega * t)
        if tol <= tolerance:
            finished = True
        if itr >= max_it:
            break
    var.set_var_tensor(x)
    var.VARo = x_old
    return {"itr": itr, "tol": tol}


def run(method: str, var: Any, rhs: Tensor, terms: list[dict], ctx: Any, tolerance: float, max_it: int) -> dict:
    if method == "jacobi":
        raise NotImplementedError(
            "pyapes_amd: Jacobi with a BC callable that reads the field: no reference behaviour to follow "
            "(the reference has no Jacobi); use cg / bicgstab, or a callable of (grid, mask) only.")
    warnings.warn("pyapes_amd: a BC callable reads the field it is given; the solve runs on the host-stepped slow "
                  "path (one return to Python per face and iteration, like the reference)", RuntimeWarning, stacklevel=3)
    rhs = rhs if rhs.dim() == var.mesh.dim + 1 else rhs.unsqueeze(0)
    return (cg if method == "cg" else bicgstab)(var, rhs, terms, ctx, tolerance, max_it)
