"""Krylov / relaxation solvers (mirrors ``pyapes/solver/linalg.py``).

``solve(var, rhs, Aop, eqs, config, mesh)`` keeps the reference signature; the loops
themselves are ``pa_cg`` / ``pa_bicgstab`` / ``pa_jacobi`` of libpyapes_hip: the
iteration, its two global reductions, the BC fill and the stop test all stay on the
device, the host only polls a done flag.
"""
from __future__ import annotations

import warnings
from typing import Any, Callable, TypedDict

import torch
from torch import Tensor

from ..backend import require_gpu
from ..hip import lib as L
from ..hip.context import context_for
from ..mesh.tools import boundary_slicer
from ..variables import Field
from .fdc import _adv_of, div_kind
from .tools import FDMSolverConfig
from .types import OPStype


class ReportType(TypedDict):
    """linalg.py:22-30"""

    itr: int
    tol: float
    converge: bool


def terms_of(eqs: dict[int, OPStype]) -> tuple[list[dict], list]:
    """ops dict -> kernel term list (+ the BC list the stencils were built against)."""
    terms: list[dict] = []
    bcs: list | None = None
    for key in eqs:
        op = eqs[key]
        name = op["name"].lower()
        if name == "ddt":
            continue
        spec = op["A_coeffs"]
        if bcs is None:
            bcs = spec.bcs
        if name == "laplacian":
            terms.append({"kind": L.OP_LAPLACIAN, "sign": op["sign"], "coeff": op["param"][0]})
        elif name == "grad":
            terms.append({"kind": L.OP_GRAD, "sign": op["sign"], "coeff": op["param"][0]})
        elif name == "div":
            var_j, cfg = op["param"]
            lim = cfg["div"]["limiter"].lower() if "limiter" in cfg["div"] else "none"
            terms.append({"kind": div_kind(lim, bool(cfg["div"].get("compat", False))), "sign": op["sign"],
                          "u": _adv_of(var_j, op["target"])})
        else:
            raise ValueError(f"pyapes_amd: unknown operator {op['name']}")
    return terms, (bcs or [])


def solve(var: Field, rhs: Tensor, Aop: Callable[..., Tensor] | None, eqs: dict[int, OPStype],
          config: FDMSolverConfig, mesh: Any) -> ReportType:
    """linalg.py:33-71.  ``Aop`` is accepted for signature compatibility; the operator is
    taken from ``eqs`` and applied inside the fused kernels."""
    method = config["method"]
    assert isinstance(method, str) and method is not None, "Linalg: solver method is not defined!"
    method = method.lower()
    if method not in ("cg", "bicgstab", "jacobi"):
        raise RuntimeError(
            f"Linalg: solver only supports CG, BICGSTAB and JACOBI. {method=} would be a typo or is not supported.")
    return _run(method, var, rhs, eqs, config, mesh)


def cg(var, rhs, Aop, eqs, config, mesh) -> ReportType:
    return _run("cg", var, rhs, eqs, config, mesh)


def bicgstab(var, rhs, Aop, eqs, config, mesh) -> ReportType:
    return _run("bicgstab", var, rhs, eqs, config, mesh)


def jacobi(var, rhs, Aop, eqs, config, mesh) -> ReportType:
    return _run("jacobi", var, rhs, eqs, config, mesh)


def _run_slab(method: str, var: Field, rhs: Tensor, eqs: dict[int, OPStype], config: FDMSolverConfig,
              mesh: Any) -> ReportType:
    """``Solver.solve()`` on ``Mesh(..., slab=(rank, world))``: the same call on every rank of the process group, the grid
    cut into slabs along axis 0 (pyapes_amd/slab.py: plane exchange with the two neighbours, scalar all-reduces).
    ``set_eq`` has adjusted the rhs already -- rank-locally, each rank for the Neumann layers it owns."""
    from ..slab import slab_solver
    if mesh.obstacle is not None and len(var.bcs) > 0:
        raise NotImplementedError  # linalg.py:287-292
    if var.dim != 1:
        raise NotImplementedError("pyapes_amd: solver equations are scalar (SURVEY Q7)")
    backend = context_for(mesh)
    if not getattr(backend, "is_standin", False):
        require_gpu(var(), f"linalg.{method}")
    if not var().is_contiguous():
        var.set_var_tensor(var().contiguous())
    tol, max_it = config["tol"], config["max_it"]
    boundary_slicer(mesh.dim, var.bcs)
    if any(bc.depends_on_var(var()) for bc in var.bcs):
        raise NotImplementedError("pyapes_amd: BC callables that read the iterate are single-GPU only (solver/host_stepped.py)")
    if config.get("save_old", False):
        raise NotImplementedError('pyapes_amd: {"save_old": True} is single-GPU only')
    terms, _ = terms_of(eqs)
    drv = slab_solver(method, mesh, var, rhs, terms, backend=backend, omega=float(config.get("omega", 1.0)))
    rep = drv.solve(tol, max_it, adjust_rhs=False)
    var.mark_old_stale("solved on a slab mesh")
    itr, rtol = int(rep.itr), float(rep.tol)
    if getattr(rep, "status", 0) != 0:
        raise RuntimeError(f"Invalid tolerance detected! tol: {rtol}")   # linalg.py:334-336
    hit_max = (itr > max_it) if method != "bicgstab" else (itr >= max_it and rtol > tol)
    if hit_max:
        warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
    elif config.get("report", False) and method != "bicgstab" and getattr(mesh, "slab", (0, 1))[0] == 0:
        _solution_report(itr, rtol, method.upper())
    if config.get("report", False) and method == "bicgstab" and getattr(mesh, "slab", (0, 1))[0] == 0:
        _solution_report(itr, rtol, "BICGSTAB")
    var.last_gpu_ms = None
    return {"itr": itr, "tol": rtol, "converge": bool(itr < max_it)}


def _run(method: str, var: Field, rhs: Tensor, eqs: dict[int, OPStype], config: FDMSolverConfig,
         mesh: Any) -> ReportType:
    if getattr(mesh, "slab", None) is not None:
        return _run_slab(method, var, rhs, eqs, config, mesh)
    require_gpu(var(), f"linalg.{method}")
    if mesh.obstacle is not None and len(var.bcs) > 0:
        raise NotImplementedError  # linalg.py:287-292
    if var.dim != 1:
        raise NotImplementedError("pyapes_amd: solver equations are scalar (SURVEY Q7)")
    if not var().is_contiguous():
        var.set_var_tensor(var().contiguous())
    tol, max_it = config["tol"], config["max_it"]
    # linalg.py:104 / 183: the reference resolves the periodic faces through the xyz letter table
    # (mesh/tools.py:11-13), so a periodic face of an rz mesh raises KeyError / IndexError there too
    boundary_slicer(mesh.dim, var.bcs)
    ctx = context_for(mesh)
    terms, _ = terms_of(eqs)
    # a callable BC value is evaluated ONCE per solve by the device loops; the reference re-evaluates it with
    # the current iterate inside every BC fill (bcs.py:203, 245).  The two agree unless the callable reads the
    # iterate -- detected here, and such a solve takes the host-stepped loop (solver/host_stepped.py), which
    # comes back to Python for every face of every fill exactly as the reference does.
    if any(bc.depends_on_var(var()) for bc in var.bcs):
        from . import host_stepped
        res = host_stepped.run(method, var, rhs, terms, ctx, tol, max_it)
        itr, rtol = int(res["itr"]), float(res["tol"])
        hit_max = (itr > max_it) if method != "bicgstab" else (itr >= max_it and rtol > tol)
        if hit_max:
            warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
        elif config.get("report", False) and method != "bicgstab":
            _solution_report(itr, rtol, method.upper())
        if config.get("report", False) and method == "bicgstab":
            _solution_report(itr, rtol, "BICGSTAB")
        var.last_gpu_ms = None
        return {"itr": itr, "tol": rtol, "converge": bool(itr < max_it)}
    ctx.bind_bcs(var(), var.bcs, 0)          # the BC fill uses the solved field's own list
    ctx.set_terms(terms)
    # Field.VARo: the reference's loops call var.save_old() at the top of every iteration (linalg.py:110, 210).
    # The device loops do that copy only on request ({"save_old": True}); otherwise VARo is marked stale and
    # reading it raises instead of returning something the reference would not have there.
    x_old = torch.empty_like(var()) if config.get("save_old", False) else None
    ctx.keep_old(None if x_old is None else x_old[0])
    try:
        rep = ctx.solve(method, var()[0], rhs[0] if rhs.dim() == mesh.dim + 1 else rhs, tol, max_it,
                        omega=float(config.get("omega", 1.0)))
    finally:
        ctx.keep_old(None)
    if x_old is not None and rep.itr >= 1:
        var.VARo = x_old
    elif x_old is None:
        var.mark_old_stale('solved on the device without {"fdm": {"save_old": True}}')
    hit_max = (rep.itr > max_it) if method != "bicgstab" else (rep.itr >= max_it and rep.tol > tol)
    if hit_max:
        warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
    elif config.get("report", False) and method != "bicgstab":
        _solution_report(rep.itr, rep.tol, method.upper())
    if config.get("report", False) and method == "bicgstab":
        _solution_report(rep.itr, rep.tol, "BICGSTAB")
    report: ReportType = {"itr": int(rep.itr), "tol": float(rep.tol), "converge": bool(rep.itr < max_it)}
    var.last_gpu_ms = rep.gpu_ms
    return report


def _solution_report(itr: int, tol: float, method: str) -> None:
    print(f"\n{method}: The solution  converged after {itr} iteration.")
    print(f"\ttolerance: {tol}")
