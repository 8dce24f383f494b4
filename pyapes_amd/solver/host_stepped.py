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
  * AXPYs, dot products and the stop-test norm as the reference writes them, as HIP kernels too (``pa_vec_axpy``: the
    product rounded before the sum; ``pa_vec_dot``: products in the field's dtype, summed in double); the scalars
    between them are host numbers of the field's dtype.  torch only allocates and copies here.

It is slow by construction (a dozen launches and two host round trips per iteration) and only chosen when
``BC.depends_on_var`` says a callable needs it.  Jacobi has no counterpart in the reference and stays refused.
"""
from __future__ import annotations

import math
import warnings
from typing import Any

import numpy as np
import torch
from torch import Tensor


def _nan_to_num(v: Any) -> Any:
    """linalg.py:302-305 on a scalar of the field's dtype"""
    return type(v)(0.0) if (math.isnan(float(v)) or math.isinf(float(v))) else v


class _Ops:
    """The device calls of one solve.  Scalars (alpha, beta, rho ...) live on the host in the FIELD's dtype (numpy scalar
    types round like the reference's 0-dim tensors do); every array operation is a C-ABI call."""

    def __init__(self, var: Any, terms: list[dict], ctx: Any):
        self.var, self.ctx, self.bcs = var, ctx, list(var.bcs)
        ctx.set_terms(terms)
        self.T = np.float64 if var().dtype == torch.float64 else np.float32
        self._bound = False

    def fill(self, x: Tensor) -> None:
        """``_apply_bc_otf`` (linalg.py:282-299): face after face, the callable sees the partly filled field"""
        for bc in self.bcs:
            self.ctx.apply_bcs(x, [bc], comps=[0])
        self._bound = False

    def _bind(self, like: Tensor) -> None:
        if not self._bound:     # the stencil rows and the interior set only need the faces' TYPES
            self.ctx.bind_bcs(like, self.bcs, 0, types_only=True)
            self._bound = True

    def A(self, d: Tensor, out: Tensor | None = None) -> Tensor:
        """(A d) on the interior set, 0 elsewhere"""
        self._bind(d)
        return self.ctx.aop(d[0], interior_only=True, out=None if out is None else out[0]).unsqueeze(0)

    def residual(self, rhs: Tensor, x: Tensor) -> Tensor:
        """r = rhs - A x on the interior set, 0 elsewhere (linalg.py:99-101)"""
        r = self.A(x)
        self.ctx.vec_axpy(r[0], rhs[0], -1.0, r[0])
        self.ctx.vec_mask_interior(r[0])
        return r

    def axpy(self, out: Tensor, y: Tensor, a: Any, x: Tensor) -> Tensor:
        self.ctx.vec_axpy(out[0], y[0], float(a), x[0])
        return out

    def dot(self, a: Tensor, b: Tensor) -> Any:
        return self.T(self.ctx.vec_dot(a[0], b[0]))

    def norm_diff(self, a: Tensor, b: Tensor) -> float:
        """linalg.py:321-338 (scalar fields: one component): ||a - b||_2, raising on a non-finite value"""
        v = float(np.sqrt(self.T(self.ctx.vec_dot(a[0], b[0], diff=True))))
        if v != v or v in (float("inf"), float("-inf")):
            raise RuntimeError(f"Invalid tolerance detected! tol: {v}")
        return v

    def norm(self, a: Tensor) -> float:
        v = float(np.sqrt(self.dot(a, a)))
        if v != v or v in (float("inf"), float("-inf")):
            raise RuntimeError(f"Invalid tolerance detected! tol: {v}")
        return v


def cg(var: Any, rhs: Tensor, terms: list[dict], ctx: Any, tolerance: float, max_it: int) -> dict:
    """linalg.py:74-159"""
    op = _Ops(var, terms, ctx)
    x = var()
    tol, itr = 1.0, 0
    op.fill(x)
    r = op.residual(rhs, x)
    d = r.clone()
    Ad = torch.empty_like(x)
    x_old = x.clone()
    rr = op.dot(r, r)
    with np.errstate(all="ignore"):
        while tol > tolerance:
            x_old.copy_(x)
            op.A(d, out=Ad)
            alpha = _nan_to_num(rr / op.dot(d, Ad))
            op.axpy(x, x, alpha, d)                  # x = x + alpha * d
            op.fill(x)
            beta_denom = rr
            op.axpy(r, r, -alpha, Ad)                # r -= alpha * Ad
            tol = op.norm_diff(x, x_old)
            rr = op.dot(r, r)
            beta = rr / beta_denom
            op.axpy(d, r, beta, d)                   # d = r + beta * d
            itr += 1
            if itr > max_it:
                break
    var.set_var_tensor(x)
    var.VARo = x_old if itr else x.clone()
    return {"itr": itr, "tol": tol}


def bicgstab(var: Any, rhs: Tensor, terms: list[dict], ctx: Any, tolerance: float, max_it: int) -> dict:
    """linalg.py:162-279"""
    op = _Ops(var, terms, ctx)
    T = op.T
    x = var()
    itr = 0
    op.fill(x)
    r0 = op.residual(rhs, x)
    r = r0.clone()
    v = torch.zeros_like(x)
    p = torch.zeros_like(x)
    s = torch.empty_like(x)
    t = torch.empty_like(x)
    rho, alpha, omega = T(1.0), T(1.0), T(1.0)
    rho_next = op.dot(r0, r0)
    tol = float(np.sqrt(rho_next))
    x_old = x.clone()
    finished = False
    with np.errstate(all="ignore"):
        while not finished:
            x_old.copy_(x)
            beta = rho_next / rho * alpha / omega
            rho = rho_next
            op.axpy(p, p, -omega, v)                 # p = r + beta * (p - omega * v)
            op.axpy(p, r, beta, p)
            op.A(p, out=v)
            itr += 1
            alpha = _nan_to_num(rho / op.dot(r0, v))
            op.axpy(s, r, -alpha, v)                 # s = r - alpha * v
            tol = op.norm(s)                         # ||r - alpha v||
            if tol <= tolerance:
                op.axpy(x, x, alpha, p)
                op.fill(x)
                finished = True
                continue
            op.A(s, out=t)
            omega = _nan_to_num(op.dot(t, s) / op.dot(t, t))
            rho_next = -omega * op.dot(r0, t)
            op.axpy(x, x, alpha, p)                  # x = x + alpha * p + s * omega
            op.axpy(x, x, omega, s)
            op.fill(x)
            op.axpy(r, s, -omega, t)                 # r = s - omega * t
            tol = op.norm(r)                         # ||s - omega t||
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
