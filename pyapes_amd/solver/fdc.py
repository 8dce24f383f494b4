"""Explicit finite-difference discretisers (mirrors ``pyapes/solver/fdc.py``).

``FDC(config).laplacian(var)``, ``.grad(var)``, ``.div(var_j, var_i)`` evaluate the
2nd-order stencils on the current field and return a tensor; ``build_A_coeffs`` /
``adjust_rhs`` / ``apply`` keep their meaning.  All arithmetic is done by
``k_aop`` / ``k_grad`` / ``k_edge`` / ``k_rhs_adjust`` in ``csrc/pa_ops.hip`` and, for the general
Div (Jac advection, vector targets, edge=True in n-D) and ``DiffFlux``, ``csrc/pa_rfp.hip``.
"""
from __future__ import annotations

import warnings
from typing import Any

import torch
from torch import Tensor

from ..backend import require_gpu
from ..hip import lib as L
from ..hip.context import context_for
from ..geometry.basis import n2d_coord
from ..variables import Field
from ..variables.container import Hess, Jac
from .tools import StencilSpec
from .types import DiscretizerConfigType


def _limiter(cfg: dict | None) -> tuple[str, bool]:
    """fdc.py:697-705"""
    if cfg is not None and "limiter" in cfg:
        return cfg["limiter"].lower(), bool(cfg.get("compat", False))
    warnings.warn("FDM: no limiter is specified. Use `none` (central difference) as a default.")
    return "none", False


def div_kind(limiter: str, compat: bool) -> int:
    if limiter == "none":
        return L.OP_DIV_CENTRAL
    if limiter == "upwind":
        return L.OP_DIV_UPWIND_COMPAT if compat else L.OP_DIV_UPWIND
    if limiter == "quick":
        raise NotImplementedError("FDC Div: quick scheme is not implemented yet.")
    raise RuntimeError(f"FDC Div: {limiter=} is an unknown limiter type.")


def _adv_of(var_j: Any, var_i: Field) -> float | Tensor:
    """fdc.py:775-792: float stays a scalar (the kernels broadcast it), Tensor / Field must be
    field-shaped; a scalar target is advected by component 0 on every axis."""
    if isinstance(var_j, Field):
        var_j = var_j()
    if isinstance(var_j, Tensor):
        assert var_j.shape == var_i().shape, "FDC Div: adv shape must match var_i shape"
        return var_j
    if isinstance(var_j, (float, int)):
        return float(var_j)
    if isinstance(var_j, Jac):
        # fdc.py:726-733: a scalar target takes Jac[n2d[0]] on every axis (like adv[0])
        return var_j[n2d_coord(var_i.mesh.coord_sys)[0]].unsqueeze(0)
    raise NotImplementedError("FDC: var_j Hess is not implemented yet!")


def _div_plan(var_j: Any, var_i: Field, edge: bool) -> tuple[list, float, list, list]:
    """Resolve the reference's indexing of target and advection per mesh axis (fdc.py:93-102, 292-311,
    708-792) -> (x, u, u_int, u_edge): one entry per axis, ``None`` = the scalar ``u``.  Raises
    IndexError where the reference's own indexing does (edge=True with a scalar target and a float /
    same-shaped tensor / scalar-Field advection on a mesh with more than one axis)."""
    mesh = var_i.mesh
    nd, vd = mesh.dim, var_i.dim
    n2d = n2d_coord(mesh.coord_sys)
    if vd != 1 and vd < nd:
        raise IndexError(f"index {vd} is out of bounds for dimension 0 with size {vd}")   # var[idx], fdc.py:99
    if isinstance(var_j, Hess):
        raise NotImplementedError("FDC: var_j Hess is not implemented yet!")
    u = 0.0
    full: Tensor | None = None
    if isinstance(var_j, Field):
        full = var_j()
    elif isinstance(var_j, Tensor):
        assert var_j.shape == var_i().shape, "FDC Div: adv shape must match var_i shape"
        full = var_j
    elif isinstance(var_j, (float, int)):
        u = float(var_j)
    elif not isinstance(var_j, Jac):
        raise NotImplementedError("FDC: var_j Hess is not implemented yet!")

    def interior(i: int) -> Tensor | None:          # advection multiplying var component i
        if isinstance(var_j, Jac):
            return var_j[n2d[i]]
        return None if full is None else full[i]

    def at_edge(a: int) -> Tensor | None:           # _treat_edge's own choice, fdc.py:294-309
        if isinstance(var_j, Jac):
            return var_j[n2d[a]]
        if isinstance(var_j, Field):
            return var_j[a]
        if isinstance(var_j, Tensor):
            return var_j[a] if var_j.shape == var_i().shape else var_j
        var_i[a]                                      # torch.ones_like(var[dim]) * var_add
        return None

    x = [var_i()[0] if vd == 1 else var_i()[a] for a in range(nd)]
    u_int = [interior(0 if vd == 1 else a) for a in range(nd)]
    u_edge = [at_edge(a) if edge else None for a in range(nd)]
    return x, u, u_int, u_edge


class Discretizer:
    """Base of the three operators (fdc.py:25-168)."""

    _op_type = "Discretizer"

    def __init__(self):
        self.A_coeffs: StencilSpec | None = None
        self._rhs_args: tuple | None = None
        self._config: DiscretizerConfigType | None = None

    @property
    def op_type(self) -> str:
        return self._op_type

    @property
    def config(self) -> DiscretizerConfigType | None:
        return self._config

    def set_config(self, config: DiscretizerConfigType) -> None:
        self._config = config

    def reset(self) -> None:
        self.A_coeffs = None
        self._rhs_args = None

    @property
    def rhs_adj(self) -> Tensor | None:
        """RHS adjustment of the last call (evaluated on first access)."""
        if self._rhs_args is None:
            return None
        return self.adjust_rhs(*self._rhs_args)

    def _edge(self) -> bool:
        if self._config is not None and self.op_type.lower() in self._config:
            return bool(self._config[self.op_type.lower()].get("edge", False))  # type: ignore[literal-required]
        warnings.warn("FDC: config is not defined! Using default config (edge=False).")
        return False

    def __call__(self, *args: Any) -> Tensor:
        if len(args) == 1:
            assert isinstance(args[0], Field), "FDC: only `Field` is allowed for var!"
            self.A_coeffs = self.build_A_coeffs(args[0])
            self._rhs_args = (args[0],)
            return self.apply(self.A_coeffs, args[0])
        assert isinstance(args[1], Field), "FDC: only `Field` is allowed for var_i!"
        self.A_coeffs = self.build_A_coeffs(args[0], args[1], config=self._config)
        self._rhs_args = (args[0], args[1], self._config)
        self.var_addition = args[0]
        return self.apply(self.A_coeffs, args[1])


def _rhs_adjust(var: Field, term: dict, bcs: list) -> Tensor:
    """rhs_adj tensor of one operator: the kernel adds into zeros."""
    ctx = context_for(var.mesh)
    if not getattr(ctx, "is_standin", False):   # (tests hang a torch stand-in for the C calls on a CPU slab mesh)
        require_gpu(var(), "adjust_rhs")
    out = torch.zeros_like(var())
    for d in range(var.dim):
        ctx.bind_bcs(var(), bcs, d, for_rhs=True)
        ctx.set_terms([term])
        ctx.rhs_adjust(out[d])
    return out


class Laplacian(Discretizer):
    _op_type = "Laplacian"

    @staticmethod
    def build_A_coeffs(var: Field) -> StencilSpec:
        return StencilSpec("Laplacian", list(var.bcs) if var.bcs is not None else [])

    @staticmethod
    def adjust_rhs(var: Field) -> Tensor:
        """``+(2/3) V n / dx`` on the first interior plane of every neumann face (fdc.py:426-458)."""
        return _rhs_adjust(var, {"kind": L.OP_LAPLACIAN}, var.bcs or [])

    def apply(self, A_coeffs: StencilSpec, var: Field) -> Tensor:
        assert A_coeffs is not None, "FDC: A_A_coeffs is not defined!"
        require_gpu(var(), "FDC.laplacian")
        ctx = context_for(var.mesh)
        edge = self._edge()
        out = torch.empty_like(var())
        for d in range(var.dim):
            ctx.bind_bcs(var(), A_coeffs.bcs, d)
            ctx.laplacian(var()[d], edge, out=out[d])
        return out


class Grad(Discretizer):
    """Returns ``(var.dim, mesh.dim, *nx)`` (fdc.py:461-502)."""

    _op_type = "Grad"

    @staticmethod
    def build_A_coeffs(var: Field) -> StencilSpec:
        return StencilSpec("Grad", list(var.bcs) if var.bcs is not None else [])

    @staticmethod
    def adjust_rhs(var: Field) -> Tensor:
        return _rhs_adjust(var, {"kind": L.OP_GRAD}, var.bcs or [])

    def apply(self, A_coeffs: StencilSpec, var: Field) -> Tensor:
        assert A_coeffs is not None, "FDC: A_A_coeffs is not defined!"
        require_gpu(var(), "FDC.grad")
        ctx = context_for(var.mesh)
        edge = self._edge()
        out = torch.empty((var.dim, var.mesh.dim, *var.mesh.nx), dtype=var().dtype, device=var().device)
        for d in range(var.dim):
            ctx.bind_bcs(var(), A_coeffs.bcs, d)
            ctx.grad(var()[d], edge, out=out[d])
        return out


class Div(Discretizer):
    """``div(var_j, var_i)``: central (limiter "none") or upwind (fdc.py:612-694).

    limiter "upwind" evaluates the first-order upwind derivative the reference's own test
    states (tests/test_fdm.py:239); ``{"div": {"limiter": "upwind", "compat": True}}``
    reproduces the reference's literal (defective) output instead.
    """

    _op_type = "Div"

    @staticmethod
    def build_A_coeffs(var_j: Any, var_i: Field, config: DiscretizerConfigType | None = None) -> StencilSpec:
        assert config is not None and "div" in config, "FDC Div: config should contain 'div' key."
        limiter, compat = _limiter(config["div"])
        div_kind(limiter, compat)
        return StencilSpec("Div", list(var_i.bcs) if var_i.bcs is not None else [], limiter, compat, var_j)

    @staticmethod
    def adjust_rhs(var_j: Any, var_i: Field, config: DiscretizerConfigType | None = None) -> Tensor:
        assert config is not None and "div" in config, "FDC Div: config should contain 'div' key."
        limiter, compat = _limiter(config["div"])
        return _rhs_adjust(var_i, {"kind": div_kind(limiter, compat), "u": _adv_of(var_j, var_i)},
                           var_i.bcs or [])

    def apply(self, A_coeffs: StencilSpec, var: Field) -> Tensor:
        assert A_coeffs is not None, "FDC: A_A_coeffs is not defined!"
        require_gpu(var(), "FDC.div")
        edge = self._edge()
        kind = div_kind(A_coeffs.limiter, A_coeffs.compat)
        ctx = context_for(var.mesh)
        ctx.bind_bcs(var(), A_coeffs.bcs, 0)
        out = torch.empty((1, *var.mesh.nx), dtype=var().dtype, device=var().device)
        var_j = A_coeffs.var_j
        if var.dim == 1 and not edge and not isinstance(var_j, (Jac, Hess)):
            ctx.div(kind, _adv_of(var_j, var), var()[0], out=out[0])      # the solver's operator, tiled
            return out
        if not var().is_contiguous():
            var.set_var_tensor(var().contiguous())
        x, u, u_int, u_edge = _div_plan(var_j, var, edge)
        ctx.div_general(kind, edge, x, u, u_int, u_edge, out[0])
        return out


class DiffFlux:
    """``D_ij d(phi)/dx_j`` of a scalar field as a vector ``Field`` (fdc.py:818-856); the r row of an
    rz mesh carries a factor r.  No boundary treatment beyond the edge=True gradient."""

    @staticmethod
    def __call__(diff: Hess, var: Field) -> Field:
        mesh = var.mesh
        n2d = n2d_coord(mesh.coord_sys)
        jac = jacobian(var)
        nd = mesh.dim
        flux = Field("DiffFlux", len(jac), mesh, None)
        D = [diff[n2d[i] + n2d[j]].contiguous() for i in range(nd) for j in range(nd)]
        J = [jac[n2d[j]].contiguous() for j in range(nd)]
        out = torch.empty((nd, *mesh.nx), dtype=var().dtype, device=var().device)
        context_for(mesh).diff_flux(D, J, out)
        flux.set_var_tensor(out)
        return flux


class FDC:
    """Collection of the explicit operators; every instance has its own operator objects
    (the reference shares class-level singletons)."""

    def __init__(self, config: DiscretizerConfigType | None = None):
        self.div = Div()
        self.laplacian = Laplacian()
        self.grad = Grad()
        self.diffFlux = DiffFlux()
        self.config = config
        if config is not None:
            for c in config:
                getattr(self, c).set_config(config)

    def update_config(self, scheme: str, target: str, val: Any) -> None:
        if self.config is not None:
            self.config.setdefault(scheme, {})[target] = val  # type: ignore[index]
        else:
            self.config = {scheme: {target: val}}  # type: ignore[assignment,misc]
        for c in self.config:  # type: ignore[union-attr]
            getattr(self, c).set_config(self.config)


def _edge_grad_of(mesh, field_nd: Tensor) -> Tensor:
    """``(mesh.dim, *nx)``: central gradient with the one-sided 2nd-order boundary formulas of
    ``_treat_edge`` (fdc.py:260-288) of one scalar array -- no BC rows (container field without BCs)."""
    require_gpu(field_nd, "jacobian / hessian")
    ctx = context_for(mesh)
    ctx.bind_bcs(field_nd, [], 0)
    return ctx.grad(field_nd.contiguous(), True)


def jacobian(var: Field) -> Jac:
    """First derivatives of a scalar field (fdc.py:896-914): ``Jac(x=.., y=.., z=..)``."""
    assert var().shape[0] == 1, "Scalar: var must be a scalar field."
    n2d = n2d_coord(var.mesh.coord_sys)
    g = _edge_grad_of(var.mesh, var()[0])
    return Jac(**{n2d[i]: g[i] for i in range(var.mesh.dim)})


def hessian(var: Field) -> Hess:
    """Second derivatives of a scalar field as the gradient of each Jacobian component, upper
    triangle only (fdc.py:917-944): ``Hess(xx=.., xy=.., ...)``."""
    assert var().shape[0] == 1, "Scalar: var must be a scalar field."
    n2d = n2d_coord(var.mesh.coord_sys)
    nd = var.mesh.dim
    g = _edge_grad_of(var.mesh, var()[0])
    data = {}
    for i in range(nd):
        gi = _edge_grad_of(var.mesh, g[i])
        for j in range(i, nd):
            data[n2d[i] + n2d[j]] = gi[j]
    return Hess(**data)
