#!/usr/bin/env python3
"""CPU oracle for the pyapes FDM stencil + Krylov hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``pyapes_amd/`` imports it, and the product path has no CPU
fallback: it raises when the HIP library is missing.

What it is: a from-scratch torch-CPU restatement of the *literal* algorithm of
the reference (kyoungseoun-chung/pyapes v0.2.13), op for op -- dense coefficient
tables, ``torch.roll`` stencils, boolean-mask boundary fill in list order, the
``|x_new - x_old|`` stop test -- written from SURVEY.md Appendix A.  Because it
keeps the reference's operation sequence (including its temporaries) it also
serves as the ``cpu_baseline`` (kind "port") timed by ``bench.py``.

Parity pin: ``tests/golden/make_golden.py`` (run in the build container only,
where /root/reference is importable) drives the reference and this file on the
same seeded inputs and stores the reference's outputs as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks this file against those vectors
(bit-exact for operator / BC-fill outputs, <=1e-13 rel for solver iterates).

Reference citations (relative to the reference repo root):
  mesh / masks      pyapes/mesh/_mesh.py:30-117, 321-399
  BC objects        pyapes/variables/bcs.py:70-95 (shifted masks), 200-280 (apply)
  interior slicer   pyapes/mesh/tools.py:7-20
  stencil tables    pyapes/solver/tools.py:29-108, pyapes/solver/fdc.py:376-423,
                    480-492, 543-609, 623-664, 708-772
  stencil apply     pyapes/solver/fdc.py:67-118, 171-200
  rhs adjustment    pyapes/solver/fdc.py:426-458, 505-540, 667-694
  edge treatment    pyapes/solver/fdc.py:203-366
  operator sum      pyapes/solver/ops.py:122-154, pyapes/solver/fdm.py:159-169
  CG / BiCGSTAB     pyapes/solver/linalg.py:74-159, 162-279, 282-338
Functions marked [NEW] have no reference counterpart.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

import torch
from torch import Tensor

FACES = ["xl", "xu", "yl", "yu", "zl", "zu"]
FACES_RZ = ["rl", "ru", "zl", "zu"]          # geometry/basis.py:18 (Cylinder: axis 0 = r, axis 1 = z)
_AXIS = {"x": 0, "y": 1, "z": 2}
_AXIS_RZ = {"r": 0, "z": 1}                   # geometry/basis.py:10


def _tdtype(dtype: str | int) -> torch.dtype:
    if dtype in ("double", "d", 64):
        return torch.float64
    if dtype in ("single", "s", 32):
        return torch.float32
    raise ValueError(f"oracle: unknown dtype {dtype!r}")


# --------------------------------------------------------------------------
# mesh (pyapes/mesh/_mesh.py:30-117)
# --------------------------------------------------------------------------
class OMesh:
    """Node-based equidistant box mesh; ``spacing`` = node counts (ints) or dx (floats).
    ``coord="rz"``: axisymmetric (Cylinder) mesh, 2-D, axis 0 = r >= 0 (_mesh.py:48-49, 121-131)."""

    def __init__(self, lower: Sequence[float], upper: Sequence[float],
                 spacing: Sequence[int | float], dtype: str | int = "double", coord: str = "xyz"):
        assert coord in ("xyz", "rz")
        self.coord = coord
        if coord == "rz":
            assert len(lower) == 2 and lower[0] >= 0
        self.dtype = _tdtype(dtype)
        self.lower = [float(v) for v in lower]
        self.upper = [float(v) for v in upper]
        self.dim = len(self.lower)
        lo = torch.tensor(self.lower, dtype=self.dtype)
        up = torch.tensor(self.upper, dtype=self.dtype)
        lx = up - lo
        if any(isinstance(s, int) for s in spacing):
            self.nx = [int(s) for s in spacing]
            self.dx_list = [float(l / (n - 1.0)) for l, n in zip(lx, self.nx)]
        else:
            self.dx_list = [float(s) for s in spacing]
            self.nx = [int(l / d + 1.0) for l, d in zip(lx, self.dx_list)]
        self.x = [
            torch.linspace(lo[i].item(), up[i].item(), self.nx[i], dtype=self.dtype)
            for i in range(self.dim)
        ]
        self.grid = torch.meshgrid(self.x, indexing="ij")

    @property
    def dx(self) -> Tensor:
        return torch.tensor(self.dx_list, dtype=self.dtype)

    @property
    def letters(self) -> str:
        """axis letters (geometry/basis.py:8,13)"""
        return "rz" if self.coord == "rz" else "xyz"

    @property
    def faces(self) -> list[str]:
        return FACES_RZ if self.coord == "rz" else FACES[:2 * self.dim]

    def axis_of(self, face: str) -> int:
        return (_AXIS_RZ if self.coord == "rz" else _AXIS)[face[0]]

    @property
    def R(self) -> Tensor:
        assert self.coord == "rz"
        return self.grid[0]

    def face_mask(self, face: str) -> Tensor:
        """Whole boundary plane incl. edges/corners (_mesh.py:321-399)."""
        a = self.axis_of(face)
        m = torch.zeros(*self.nx, dtype=torch.bool)
        idx: list[Any] = [slice(None)] * self.dim
        idx[a] = 0 if face[1] == "l" else self.nx[a] - 1
        m[tuple(idx)] = True
        return m


# --------------------------------------------------------------------------
# boundary conditions (pyapes/variables/bcs.py)
# --------------------------------------------------------------------------
@dataclass
class OBC:
    face: str
    type: str            # dirichlet | neumann | symmetry | periodic
    val: Any             # number | list | Tensor | callable(grid, mask, var, opt) | None
    mesh: OMesh
    opt: Any = None
    axis: int = field(init=False)
    n_dir: int = field(init=False)

    def __post_init__(self):
        self.type = self.type.lower()
        self.axis = self.mesh.axis_of(self.face)
        self.n_dir = -1 if self.face[1] == "l" else 1
        self.mask = self.mesh.face_mask(self.face)
        # bcs.py:84-93: rolled copies of the face mask along the face normal
        self.prev = torch.roll(self.mask, -self.n_dir, self.axis)
        self.prev2 = torch.roll(self.mask, -2 * self.n_dir, self.axis)
        self.fwd = torch.roll(self.mask, self.n_dir, self.axis)
        self.fwd2 = torch.roll(self.mask, 2 * self.n_dir, self.axis)
        self.n_vec = torch.zeros(3, dtype=self.mesh.dtype)
        self.n_vec[self.axis] = self.n_dir

    # bcs.py:200-280
    def apply(self, var: Tensor, d: int) -> None:
        grid = self.mesh.grid
        if self.type == "dirichlet":
            v = self.val
            if callable(v):
                var[d, self.mask] = v(grid, self.mask, var, self.opt)
            elif isinstance(v, list):
                var[d, self.mask] = v[d]
            elif isinstance(v, (int, float)):
                var[d, self.mask] = float(v)
            elif isinstance(v, Tensor):
                var[d, self.mask] = v
            else:
                raise TypeError("oracle: bad dirichlet value")
        elif self.type == "neumann":
            dx = grid[self.axis][self.mask] - grid[self.axis][self.prev]
            vp = var[d][self.prev]
            vpp = var[d][self.prev2]
            v = self.val
            if callable(v):
                c = v(grid, self.mask, var, self.opt)
            elif isinstance(v, list):
                c = v[d]
            elif isinstance(v, (int, float)):
                c = float(v)
            elif isinstance(v, Tensor):
                c = v
            else:
                raise TypeError("oracle: bad neumann value")
            var[d][self.mask] = 4 / 3 * vp - 1 / 3 * vpp + 2 / 3 * c * dx * self.n_dir
        elif self.type == "symmetry":
            var[d, self.mask] = var[d, self.prev]
        elif self.type == "periodic":
            if self.n_dir < 0:
                vp = var[d, self.prev]
                vf = var[d, self.fwd]
                vff = var[d, self.fwd2]
                var[d, self.mask] = vp - vf + vff
            else:
                var[d, self.mask] = var[d, self.fwd]
        else:
            raise ValueError(f"oracle: unknown bc type {self.type}")


def make_bcs(mesh: OMesh, cfg: Sequence[dict]) -> list[OBC]:
    """cfg: list of {bc_face, bc_type, bc_val[, bc_val_opt]} in application order."""
    return [OBC(c["bc_face"], c["bc_type"], c["bc_val"], mesh, c.get("bc_val_opt"))
            for c in cfg]


def homogeneous_cfg(dim: int, val: Any, typ: str, faces: Sequence[str] = FACES) -> list[dict]:
    return [{"bc_face": faces[i], "bc_type": typ,
             "bc_val": val[i] if isinstance(val, list) else val} for i in range(2 * dim)]


def mixed_cfg(vals: Sequence[Any], types: Sequence[str], faces: Sequence[str] = FACES) -> list[dict]:
    return [{"bc_face": faces[i], "bc_type": t, "bc_val": v}
            for i, (v, t) in enumerate(zip(vals, types))]


def bc_fill(var: Tensor, bcs: Sequence[OBC]) -> Tensor:
    """linalg.py:282-299: for every component, every face in list order."""
    for d in range(var.shape[0]):
        for bc in bcs:
            bc.apply(var, d)
    return var


def interior_slicer(ndim: int, bcs: Sequence[OBC]) -> tuple[slice, ...]:
    """mesh/tools.py:7-20.  Literal: the face letter goes through the xyz table whatever the mesh is,
    so a periodic face of an rz mesh raises here (KeyError 'r' / IndexError for 'z' -> 2)."""
    lim: list[list[int | None]] = [[1, -1] for _ in range(ndim)]
    for bc in bcs:
        if bc.type == "periodic":
            lim[_AXIS[bc.face[0]]][0 if bc.face[1] == "l" else 1] = None
    return tuple(slice(*l) for l in lim)


def _bc_value(bc: OBC, var: Tensor, d: int) -> Any:
    """fdc.py:803-817."""
    v = bc.val
    if callable(v):
        return v(bc.mesh.grid, bc.mask, var, bc.n_vec)
    if isinstance(v, list):
        return v[d]
    if isinstance(v, (int, float)):
        return v
    if v is None:
        return 0.0
    raise ValueError("oracle: unknown bc value")


# --------------------------------------------------------------------------
# coefficient tables [App, Ap, Ac, Am, Amm], each a list over mesh axes of
# tensors shaped like var (dim, *nx)         (solver/tools.py:29-108)
# --------------------------------------------------------------------------
def _nn(t: Tensor) -> Tensor:
    return torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0)


def _base_tables(var: Tensor, ndim: int, p: float, c: float, m: float):
    z = lambda: [torch.zeros_like(var) for _ in range(ndim)]  # noqa: E731
    return [z(),
            [p * torch.ones_like(var) for _ in range(ndim)],
            [c * torch.ones_like(var) if c != 0.0 else torch.zeros_like(var) for _ in range(ndim)],
            [m * torch.ones_like(var) for _ in range(ndim)],
            z()]


def laplacian_tables(var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None):
    """fdc.py:376-423; rz base rows tools.py:86-107."""
    App, Ap, Ac, Am, Amm = _base_tables(var, mesh.dim, 1.0, -2.0, 1.0)
    if mesh.coord == "rz":
        dr = mesh.dx[0]
        scale = _nn(dr / (2 * mesh.R))
        Ap[0] = (1 + scale) * torch.ones_like(var)
        Am[0] = (1 - scale) * torch.ones_like(var)
    dx = mesh.dx
    for i in range(var.shape[0]):
        for j in range(mesh.dim):
            if bcs is None:
                continue
            for bc in bcs:
                if bc.n_vec[j] == 0:
                    continue
                if bc.type in ("neumann", "symmetry"):
                    dr = mesh.dx[j] if j == 0 else 0.0
                    r = mesh.grid[j][bc.prev]
                    alpha = _nn(2 / 3 * dr / r) if mesh.coord == "rz" else torch.zeros_like(r)
                    if bc.n_dir < 0:
                        Ap[j][i][bc.prev] = 2 / 3 + alpha
                        Ac[j][i][bc.prev] = -(2 / 3 + alpha)
                        Am[j][i][bc.prev] = 0.0
                    else:
                        Ap[j][i][bc.prev] = 0.0
                        Ac[j][i][bc.prev] = -(2 / 3 + alpha)
                        Am[j][i][bc.prev] = 2 / 3 + alpha
            Ap[j][i] /= dx[j] ** 2
            Ac[j][i] /= dx[j] ** 2
            Am[j][i] /= dx[j] ** 2
    return [App, Ap, Ac, Am, Amm]


def laplacian_rhs_adjust(var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None) -> Tensor:
    """fdc.py:426-458 (literal, incl. the upper-face sign)."""
    adj = torch.zeros_like(var)
    dx = mesh.dx
    for i in range(var.shape[0]):
        if bcs is None:
            continue
        for j in range(mesh.dim):
            for bc in bcs:
                if bc.type == "neumann":
                    dr = mesh.dx[j] if j == 0 else 0.0
                    r = mesh.grid[j][bc.prev]
                    alpha = _nn(1 / 3 * dr / r) if mesh.coord == "rz" else torch.zeros_like(r)
                    at_bc = _bc_value(bc, var, i)
                    adj[i][bc.prev] += (2 / 3 - alpha) * (at_bc * bc.n_vec[j]) / dx[j]
    return adj


def _grad_adjust(var: Tensor, mesh: OMesh, bcs: Sequence[OBC], tabs, comp: int,
                 gamma: tuple[Tensor, ...] | None = None) -> None:
    """fdc.py:543-609."""
    if gamma is None:
        gmin = torch.ones_like(var)
        gmax = torch.ones_like(var)
    else:
        gmin = gamma[0]
        gmax = gamma[0] if len(gamma) == 1 else gamma[1]
    Ap, Ac, Am = tabs
    dx = mesh.dx
    for j in range(mesh.dim):
        for bc in bcs:
            if bc.n_vec[j] == 0:
                continue
            if bc.type in ("neumann", "symmetry"):
                gx = gmax[comp][bc.prev]
                gn = gmin[comp][bc.prev]
                if bc.n_dir < 0:
                    Ap[j][comp][bc.prev] += 1 / 3 * gx
                    Ac[j][comp][bc.prev] -= 1 / 3 * gn
                    Am[j][comp][bc.prev] = 0.0
                else:
                    Ap[j][comp][bc.prev] = 0.0
                    Ac[j][comp][bc.prev] += 1 / 3 * gn
                    Am[j][comp][bc.prev] -= 1 / 3 * gx
            elif bc.type == "periodic":
                if bc.n_dir < 0:
                    Am[j][comp][bc.prev] = 0.0
                else:
                    Ap[j][comp][bc.prev] = 0.0
        Ap[j][comp] /= 2.0 * dx[j]
        Ac[j][comp] /= 2.0 * dx[j]
        Am[j][comp] /= 2.0 * dx[j]


def grad_tables(var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None):
    """fdc.py:480-492."""
    App, Ap, Ac, Am, Amm = _base_tables(var, mesh.dim, 1.0, 0.0, -1.0)
    if bcs is not None:
        for i in range(var.shape[0]):
            _grad_adjust(var, mesh, bcs, [Ap, Ac, Am], i)
    return [App, Ap, Ac, Am, Amm]


def _grad_rhs(var: Tensor, mesh: OMesh, bcs: Sequence[OBC], adj: Tensor, comp: int,
              gamma: tuple[Tensor, ...] | None = None) -> None:
    """fdc.py:505-540."""
    if gamma is None:
        gmin = torch.ones_like(var)
        gmax = torch.ones_like(var)
    else:
        gmin = 2.0 * gamma[0]
        gmax = 2.0 * (gamma[0] if len(gamma) == 1 else gamma[1])
    for j in range(mesh.dim):
        for bc in bcs:
            if bc.type == "neumann":
                at_bc = _bc_value(bc, var, comp)
                g = gmax if bc.n_dir < 0 else gmin
                adj[comp][bc.prev] -= (1 / 3) * (at_bc * bc.n_vec[j]) * g[comp][bc.prev]


def grad_rhs_adjust(var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None) -> Tensor:
    adj = torch.zeros_like(var)
    if bcs is not None:
        for i in range(var.shape[0]):
            _grad_rhs(var, mesh, bcs, adj, i)
    return adj


def adv_tensor(u: Any, var: Tensor) -> Any:
    """fdc.py:775-792.  float | Tensor -> Tensor shaped like var; a Jac (dict axis letter -> Tensor,
    variables/container.py) is passed through, as Div.build_A_coeffs does (fdc.py:640-643)."""
    if isinstance(u, dict):
        return u
    if isinstance(u, float):
        return torch.ones_like(var) * u
    assert u.shape == var.shape, "oracle: adv shape must match var shape"
    return u


def _adv_comp(adv: Any, i: int, mesh: "OMesh") -> Tensor:
    """fdc.py:726-733 / 758-765: component of the advection term that multiplies var component i --
    adv[i] for tensors, adv[n2d[i]] for a Jac (n2d = axis letters of the coordinate system)."""
    if isinstance(adv, dict):
        return adv[mesh.letters[i]]
    return adv[i]


def div_tables(u: float | Tensor, var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None,
               limiter: str = "none"):
    """fdc.py:623-664, 708-772 (literal; 'upwind' is the reference's defective form, Q3)."""
    adv = adv_tensor(u, var)
    App, Ap, Ac, Am, Amm = _base_tables(var, mesh.dim, 1.0, 0.0, -1.0)
    if mesh.coord == "rz":  # tools.py:64-78: the u phi / r term of the axisymmetric divergence
        Ac[0] = _nn(2 * mesh.dx[0] / mesh.R) * torch.ones_like(var)
    limiter = limiter.lower()
    if limiter == "none":
        advection = torch.zeros_like(var[0])
        for i in range(var.shape[0]):
            for j in range(mesh.dim):
                advection = _adv_comp(adv, i, mesh)
                Ap[j][i] *= torch.roll(advection, -1, dims=j)
                Ac[j][i] *= advection
                Am[j][i] *= torch.roll(advection, 1, dims=j)
            _grad_adjust(var, mesh, bcs or [], [Ap, Ac, Am], i, (advection,))
    elif limiter == "upwind":
        zeros = torch.zeros_like(var[0])
        for i in range(var.shape[0]):
            for j in range(mesh.dim):
                advection = _adv_comp(adv, i, mesh)
                Ap[j][i] = 2.0 * torch.min(advection, zeros)
                Ac[j][i] *= 2.0 * advection
                Am[j][i] = 2.0 * torch.max(advection, zeros)
    else:
        raise RuntimeError(f"oracle: unknown limiter {limiter}")
    return [App, Ap, Ac, Am, Amm]


def div_rhs_adjust(u: float | Tensor, var: Tensor, mesh: OMesh, bcs: Sequence[OBC] | None,
                   limiter: str = "none") -> Tensor:
    """fdc.py:667-694."""
    adj = torch.zeros_like(var)
    if bcs is not None:
        adv = adv_tensor(u, var)
        if limiter.lower() == "none":
            for i in range(var.shape[0]):
                _grad_rhs(var, mesh, bcs, adj, i, (adv,))
        else:
            z = torch.zeros_like(var)
            for i in range(var.shape[0]):
                _grad_rhs(var, mesh, bcs, adj, i, (torch.min(adv, z), torch.max(adv, z)))
    return adj


# --------------------------------------------------------------------------
# stencil application (fdc.py:67-118, 171-200)
# --------------------------------------------------------------------------
def _axis_sum(tabs, var: Tensor, comp: int, axis: int) -> Tensor:
    out = torch.zeros_like(var[0])
    for k, c in enumerate(tabs):
        if var.shape[0] == 1:
            coeff, vi = c[axis][0], 0
        else:
            coeff, vi = c[axis][comp], comp
        out += coeff * torch.roll(var[vi], -2 + k, axis)
    return out


def apply_laplacian(tabs, var: Tensor, ndim: int) -> Tensor:
    out = torch.zeros_like(var)
    for i in range(var.shape[0]):
        for a in range(ndim):
            out[i] += _axis_sum(tabs, var, i, a)
    return out


def apply_grad(tabs, var: Tensor, ndim: int) -> Tensor:
    comps = []
    for i in range(var.shape[0]):
        comps.append(torch.stack([_axis_sum(tabs, var, i, a) for a in range(ndim)]))
    return torch.stack(comps)


def apply_div(tabs, var: Tensor, ndim: int, edge: tuple | None = None) -> Tensor:
    """fdc.py:93-102.  edge = (mesh, var_add): the per-axis edge treatment of edge=True."""
    out = torch.zeros_like(var[0]).unsqueeze(0)
    for a in range(ndim):
        disc = _axis_sum(tabs, var, a, a)
        if edge is not None:
            edge_div(disc, var, edge[0], a, edge[1])
        out[0] += disc
    return out


def edge_div(disc: Tensor, var: Tensor, mesh: "OMesh", dim: int, var_add: Any) -> None:
    """fdc.py:290-361: the two end planes of axis ``dim`` of this axis' contribution get the one-sided
    2nd-order derivative times the advection value (+ the rz terms, literal: the lower one without
    the advection factor).  Raises IndexError exactly where the reference's indexing does (scalar
    field with float / same-shaped tensor advection on an axis > 0)."""
    nd = mesh.dim
    if isinstance(var_add, Tensor):
        adv = var_add[dim] if var_add.shape == var.shape else var_add
    elif isinstance(var_add, float):
        adv = torch.ones_like(var[dim]) * var_add
    elif isinstance(var_add, dict):
        adv = var_add[mesh.letters[dim]]
    elif var_add is None:
        adv = torch.ones_like(var[dim])
    else:
        raise NotImplementedError
    target = var[0] if var.shape[0] == 1 else var[dim]
    for side in (0, 1):
        s = [[slice(None)] * nd for _ in range(3)]
        for q in range(3):
            s[q][dim] = q if side == 0 else -1 - q
        s1 = tuple(s[0])
        t0, t1, t2 = (target[tuple(s[q])] for q in range(3))
        if side == 0:
            disc[s1] = -(3 / 2 * t0 - 2.0 * t1 + 1 / 2 * t2) / (mesh.dx[dim]) * adv[s1]
            if mesh.coord == "rz" and dim == 0:
                disc[s1] += _nn(t0 / mesh.R[s1])
        else:
            disc[s1] = (3 / 2 * t0 - 2.0 * t1 + 1 / 2 * t2) / (mesh.dx[dim]) * adv[s1]
            if mesh.coord == "rz" and dim == 0:
                disc[s1] += _nn(t0 * adv[s1] / mesh.R[s1])


# fdc.py:896-944: first / second derivatives of a scalar array as edge=True gradients of a BC-free field
def jacobian(field: Tensor, mesh: "OMesh") -> dict:
    v = field.unsqueeze(0)
    g = apply_grad(grad_tables(v, mesh, []), v, mesh.dim)
    edge_grad(g, v, mesh)
    return {mesh.letters[i]: g[0][i] for i in range(mesh.dim)}


def hessian(field: Tensor, mesh: "OMesh") -> dict:
    jac = jacobian(field, mesh)
    out = {}
    for i in range(mesh.dim):
        gi = jacobian(jac[mesh.letters[i]], mesh)
        for j in range(i, mesh.dim):
            out[mesh.letters[i] + mesh.letters[j]] = gi[mesh.letters[j]]
    return out


def _hkey(h: dict, key: str) -> Tensor:
    return h["".join(sorted(key))]      # variables/container.py:43-58


def diff_flux(hess: dict, field: Tensor, mesh: "OMesh") -> Tensor:
    """fdc.py:818-856: D_ij d(phi)/dx_j as a vector field (mesh.dim, *nx); the r row carries a factor r."""
    jac = jacobian(field, mesh)
    L = mesh.letters
    out = torch.zeros(mesh.dim, *mesh.nx, dtype=mesh.dtype)
    for i in range(mesh.dim):
        acc = torch.zeros_like(field)
        for j in range(mesh.dim):
            d = _hkey(hess, L[i] + L[j])
            if L[i] == "r":
                d = mesh.grid[0] * d
            acc += d * jac[L[j]]
        out[i] = acc
    return out


# --------------------------------------------------------------------------
# Rosenbluth-Fokker-Planck explicit operators, rz only (solver/rfp.py)
# --------------------------------------------------------------------------
def rfp_friction(jacH: dict, pdf: Tensor, mesh: "OMesh") -> Tensor:
    """rfp.py:19-82: div(grad(H) f) in conservative form on face-averaged values, zero normal flux
    rows written afterwards in the order r=0, r=R, z=0, z=Z."""
    assert mesh.coord == "rz"
    Hr, Hz = jacH["r"], jacH["z"]
    dx = mesh.dx
    R = mesh.R
    up = lambda t, a: torch.roll(t, -1, a)   # noqa: E731  value at index + 1
    dn = lambda t, a: torch.roll(t, 1, a)    # noqa: E731  value at index - 1
    Arp, Arm = (up(Hr, 0) + Hr) / 2.0, (Hr + dn(Hr, 0)) / 2.0
    Azp, Azm = (up(Hz, 1) + Hz) / 2.0, (Hz + dn(Hz, 1)) / 2.0
    Prp, Prm = (up(pdf, 0) + pdf) / 2.0, (pdf + dn(pdf, 0)) / 2.0
    Pzp, Pzm = (up(pdf, 1) + pdf) / 2.0, (pdf + dn(pdf, 1)) / 2.0
    r_p, r_m = (up(R, 0) + R) / 2, (R + dn(R, 0)) / 2
    zflux = Azp * Pzp - Azm * Pzm
    rflux = (r_p * Arp * Prp - r_m * Arm * Prm) / (R * dx[0])
    out = zflux / dx[1] + rflux
    out[0, :] = zflux[0, :] / (dx[1])
    out[-1, :] = zflux[-1, :] / (dx[1]) + 2.0 * ((-r_m * Arm * Prm) / (R * dx[0]))[-1, :]
    out[:, 0] = 2.0 * (Azp * Pzp)[:, 0] / (dx[1]) + _nn(rflux)[:, 0]
    out[:, -1] = 2.0 * (-Azm * Pzm)[:, -1] / (dx[1]) + _nn(rflux)[:, -1]
    return out


def _sh(t: Tensor, di: int, dj: int) -> Tensor:
    """value at (i + di, j + dj), wrap-around"""
    return torch.roll(t, (-di, -dj), (0, 1))


def _face_grad(t: Tensor, p: tuple[int, int], m: tuple[int, int], h: Tensor) -> Tensor:
    """rfp.py:221-230"""
    return (_sh(t, *p) - _sh(t, *m)) / h


def _cell_avg(t: Tensor, ui: int, uj: int) -> Tensor:
    """rfp.py:233-251: mean over the cell whose upper corner is (i + ui, j + uj)"""
    return (_sh(t, ui, uj) + _sh(t, ui, uj - 1) + _sh(t, ui - 1, uj) + _sh(t, ui - 1, uj - 1)) / 4


def rfp_diffusion(hessG: dict, pdf: Tensor, mesh: "OMesh") -> Tensor:
    """rfp.py:85-218: div(D grad f), D = hess(G), symmetric differences; the mixed term uses cell-centre
    averages of D_rz; boundary rows written afterwards in the order r=0, r=R, z=0, z=Z."""
    assert mesh.coord == "rz"
    Drr, Dzz, Drz = _hkey(hessG, "rr"), _hkey(hessG, "zz"), _hkey(hessG, "rz")
    dx = mesh.dx
    R = mesh.grid[0]
    f = pdf
    rr_p = (_sh(Drr, 1, 0) + Drr) * (_sh(f, 1, 0) - f) / (2.0 * dx[0])
    rr_m = (_sh(Drr, -1, 0) + Drr) * (f - _sh(f, -1, 0)) / (2.0 * dx[0])
    zz_p = (_sh(Dzz, 0, 1) + Dzz) * (_sh(f, 0, 1) - f) / (2.0 * dx[1])
    zz_m = (_sh(Dzz, 0, -1) + Dzz) * (f - _sh(f, 0, -1)) / (2.0 * dx[1])
    c_pp, c_pm, c_mp, c_mm = _cell_avg(Drz, 1, 1), _cell_avg(Drz, 1, 0), _cell_avg(Drz, 0, 1), _cell_avg(Drz, 0, 0)
    G = _face_grad
    rz_r_zp = 0.25 * c_pp * (G(f, (1, 0), (0, 0), dx[0]) + G(f, (1, 1), (0, 1), dx[0])) \
        + 0.25 * c_mp * (G(f, (0, 0), (-1, 0), dx[0]) + G(f, (0, 1), (-1, 1), dx[0]))
    rz_r_zm = 0.25 * c_pm * (G(f, (1, -1), (0, -1), dx[0]) + G(f, (1, 0), (0, 0), dx[0])) \
        + 0.25 * c_mm * (G(f, (0, -1), (-1, -1), dx[0]) + G(f, (0, 0), (-1, 0), dx[0]))
    rz_z_rp = 0.25 * c_pp * (G(f, (0, 1), (0, 0), dx[1]) + G(f, (1, 1), (1, 0), dx[1])) \
        + 0.25 * c_mp * (G(f, (0, 0), (0, -1), dx[1]) + G(f, (1, 0), (1, -1), dx[1]))
    rz_z_rm = 0.25 * c_pm * (G(f, (-1, 1), (-1, 0), dx[1]) + G(f, (0, 1), (0, 0), dx[1])) \
        + 0.25 * c_mm * (G(f, (-1, 0), (-1, -1), dx[1]) + G(f, (0, 0), (0, -1), dx[1]))
    r_p, r_m = (_sh(R, 1, 0) + R) / 2, (R + _sh(R, -1, 0)) / 2
    rad_mixed = (r_p * rz_z_rp - r_m * rz_z_rm) / (R * dx[0])
    rad_rr = (r_p * rr_p - r_m * rr_m) / (R * dx[0])
    out = (zz_p - zz_m) / dx[1] + (rz_r_zp - rz_r_zm) / dx[1] + rad_mixed + rad_rr
    out[0, :] = (zz_p - zz_m)[0, :] / dx[1] + 2.0 * (rz_r_zp - rz_r_zm)[0, :] / dx[1]
    out[-1, :] = (((zz_p - zz_m) / dx[1] + (rz_r_zp - rz_r_zm) / dx[1])[-1, :]
                  + 2.0 * ((-r_m * rz_z_rm) / (R * dx[0]))[-1, :]
                  + 2.0 * ((-r_m * rr_m) / (R * dx[0]))[-1, :])
    out[:, 0] = 2.0 * ((zz_p) / dx[1] + (rz_r_zp) / dx[1])[:, 0] + _nn(rad_mixed + rad_rr)[:, 0]
    out[:, -1] = 2.0 * ((-zz_m) / dx[1] + (-rz_r_zm) / dx[1])[:, -1] + _nn(rad_mixed + rad_rr)[:, -1]
    return out


def minmod(a: Tensor, b: Tensor) -> Tensor:
    """rfp.py:268-286"""
    val = torch.zeros_like(a)
    m = torch.logical_and(a.ge(0.0), b.ge(0.0))
    val[m] = torch.min(a[m], b[m])
    m = torch.logical_and(a.lt(0.0), b.lt(0.0))
    val[m] = torch.max(a[m], b[m])
    val[(a * b).le(0.0)] = 0.0
    return val


def mc_limiter(a: Tensor, b: Tensor) -> Tensor:
    """rfp.py:262-265"""
    return minmod(2.0 * minmod(a, b), (a + b) / 2.0)


# fdc.py:203-366 (explicit one-sided boundary formulas, edge=True)
def edge_laplacian(out: Tensor, var: Tensor, mesh: OMesh) -> None:
    nd = mesh.dim
    for comp in range(var.shape[0]):
        for a in range(nd):
            for side in (0, 1):
                s = [[slice(None)] * nd for _ in range(4)]
                for q in range(4):
                    s[q][a] = q if side == 0 else -1 - q
                v0, v1, v2, v3 = (var[comp][tuple(s[q])] for q in range(4))
                out[comp][tuple(s[0])] = (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3) / (mesh.dx[a] ** 2)


def edge_grad(out: Tensor, var: Tensor, mesh: OMesh) -> None:
    nd = mesh.dim
    for comp in range(out.shape[0]):
        for a in range(nd):
            for side in (0, 1):
                s = [[slice(None)] * nd for _ in range(3)]
                for q in range(3):
                    s[q][a] = q if side == 0 else -1 - q
                v0, v1, v2 = (var[comp][tuple(s[q])] for q in range(3))
                e = (3 / 2 * v0 - 2.0 * v1 + 1 / 2 * v2) / (mesh.dx[a])
                out[comp][a][tuple(s[0])] = -e if side == 0 else e


# --------------------------------------------------------------------------
# operator sum (ops.py:122-154)
# --------------------------------------------------------------------------
@dataclass
class OTerm:
    kind: str                    # "laplacian" | "grad" | "div"
    tabs: Any
    param: Any = None            # laplacian/grad multiplier (float | Tensor | None)
    sign: float = 1.0


def Aop(var: Tensor, terms: Sequence[OTerm], ndim: int) -> Tensor:
    res = torch.zeros_like(var)
    for t in terms:
        if t.kind == "laplacian":
            ax = apply_laplacian(t.tabs, var, ndim)
            if t.param is not None:
                ax = ax * t.param
        elif t.kind == "grad":
            ax = apply_grad(t.tabs, var, ndim)
            if t.param is not None:
                ax = ax * t.param
        elif t.kind == "div":
            ax = apply_div(t.tabs, var, ndim)
        else:
            raise ValueError(t.kind)
        ax = ax * t.sign
        if t.kind == "grad":
            ax = ax.view(var.size())
        res += ax
    return res


def _nan_to_num(t: Tensor) -> Tensor:
    return torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0)


def _tolerance(a: Tensor, b: Tensor) -> float:
    """linalg.py:321-338."""
    tol = torch.zeros(a.shape[0], dtype=a.dtype)
    for d in range(a.shape[0]):
        tol[d] = torch.linalg.norm(a[d] - b[d])
    if torch.isnan(tol) or torch.isinf(tol):
        raise RuntimeError(f"Invalid tolerance detected! tol: {tol}")
    return torch.max(tol).item()


# --------------------------------------------------------------------------
# Krylov loops (linalg.py:74-279)
# --------------------------------------------------------------------------
def cg(x: Tensor, rhs: Tensor, terms: Sequence[OTerm], mesh: OMesh, bcs: Sequence[OBC],
       tol: float, max_it: int, history: list | None = None, snapshots: dict | None = None):
    """Returns (x, report).  x is a new tensor per iteration like the reference
    (set_var_tensor rebinding), the final one is returned."""
    nd = mesh.dim
    axes = list(range(1, nd + 1))
    S = interior_slicer(nd, bcs)
    cur_tol, itr = 1.0, 0
    bc_fill(x, bcs)
    Ad = torch.zeros_like(rhs)
    r = torch.zeros_like(x)
    for i in range(x.shape[0]):
        r[i][S] = rhs[i][S] - Aop(x, terms, nd)[i][S]
    d = r.clone()
    while cur_tol > tol:
        x_old = x.clone()
        for i in range(x.shape[0]):
            Ad[i][S] = Aop(d, terms, nd)[i][S]
        alpha = _nan_to_num(torch.sum(r * r, dim=axes) / torch.sum(d * Ad, dim=axes))
        x = x + alpha * d
        bc_fill(x, bcs)
        beta_denom = torch.sum(r * r, dim=axes)
        r -= alpha * Ad
        cur_tol = _tolerance(x, x_old)
        beta = torch.sum(r * r, dim=axes) / beta_denom
        d = r + beta * d
        itr += 1
        if history is not None:
            history.append(cur_tol)
        if snapshots is not None and itr in snapshots:
            snapshots[itr] = {"x": x.clone(), "r": r.clone(), "d": d.clone()}
        if itr > max_it:
            warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
            break
    return x, {"itr": itr, "tol": cur_tol, "converge": itr < max_it}


def bicgstab(x: Tensor, rhs: Tensor, terms: Sequence[OTerm], mesh: OMesh, bcs: Sequence[OBC],
             tol: float, max_it: int):
    nd = mesh.dim
    axes = list(range(1, nd + 1))
    S = interior_slicer(nd, bcs)
    itr = 0
    bc_fill(x, bcs)
    r0 = torch.zeros_like(x)
    for i in range(x.shape[0]):
        r0[i][S] = rhs[i][S] - Aop(x, terms, nd)[i][S]
    r = r0.clone()
    t = torch.zeros_like(x)
    v = torch.zeros_like(x)
    p = torch.zeros_like(x)
    s = torch.zeros_like(x)
    rho: Any = 1.0
    alpha: Any = 1.0
    omega: Any = 1.0
    rho_next = torch.sum(r0 * r0, dim=axes)
    cur_tol = torch.sqrt(rho_next.max()).item()
    finished = False
    while not finished:
        beta = rho_next / rho * alpha / omega
        rho = rho_next
        p = r + beta * (p - omega * v)
        for i in range(x.shape[0]):
            v[i][S] = Aop(p, terms, nd)[i][S]
        itr += 1
        alpha = _nan_to_num(rho / torch.sum(r0 * v, dim=axes))
        s = r - alpha * v
        cur_tol = _tolerance(r, alpha * v)
        if cur_tol <= tol:
            x = x + alpha * p
            bc_fill(x, bcs)
            finished = True
            continue
        for i in range(x.shape[0]):
            t[i][S] = Aop(s, terms, nd)[i][S]
        omega = _nan_to_num(torch.sum(t * s, dim=axes) / torch.sum(t * t, dim=axes))
        rho_next = -omega * torch.sum(r0 * t, dim=axes)
        x = x + alpha * p + s * omega
        bc_fill(x, bcs)
        r = s - omega * t
        cur_tol = _tolerance(s, omega * t)
        if cur_tol <= tol:
            finished = True
        if itr >= max_it:
            warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
            break
    return x, {"itr": itr, "tol": cur_tol, "converge": itr < max_it}


# --------------------------------------------------------------------------
# [NEW] pieces required by north_star that the reference lacks
# --------------------------------------------------------------------------
def laplacian_diag(var: Tensor, tabs, ndim: int) -> Tensor:
    """[NEW] diagonal of the Laplacian table: sum over axes of Ac."""
    dg = torch.zeros_like(var)
    for i in range(var.shape[0]):
        for a in range(ndim):
            dg[i] += tabs[2][a][i]
    return dg


def jacobi(x: Tensor, rhs: Tensor, terms: Sequence[OTerm], mesh: OMesh, bcs: Sequence[OBC],
           tol: float, max_it: int, omega: float = 1.0):
    """[NEW] weighted Jacobi  x <- x + omega (b - A x)/diag(A)  on the interior set,
    with the SAME BC fill, interior slicer, stop test (|x_new - x_old|_2 over all
    nodes), K+1 iteration rule and report as the reference CG (linalg.py:74-159).
    diag(A) = sum_k sign_k * param_k * sum_axes Ac  (laplacian terms only)."""
    nd = mesh.dim
    S = interior_slicer(nd, bcs)
    diag = torch.zeros_like(x)
    for t in terms:
        assert t.kind == "laplacian", "oracle jacobi: laplacian terms only"
        dg = laplacian_diag(x, t.tabs, nd)
        if t.param is not None:
            dg = dg * t.param
        diag += dg * t.sign
    cur_tol, itr = 1.0, 0
    bc_fill(x, bcs)
    while cur_tol > tol:
        x_old = x.clone()
        res = torch.zeros_like(x)
        for i in range(x.shape[0]):
            res[i][S] = (rhs[i][S] - Aop(x, terms, nd)[i][S]) / diag[i][S]
        x = x + omega * res
        bc_fill(x, bcs)
        cur_tol = _tolerance(x, x_old)
        itr += 1
        if itr > max_it:
            warnings.warn(f"Maximum iteration reached! max_it: {max_it}", RuntimeWarning)
            break
    return x, {"itr": itr, "tol": cur_tol, "converge": itr < max_it}


def div_upwind_intended(u: float | Tensor, var: Tensor, mesh: OMesh) -> Tensor:
    """[NEW] first-order upwind advection as the reference's own test states it
    (tests/test_fdm.py:239):  sum_a [ u+ (phi_i - phi_{i-1}) + u- (phi_{i+1} - phi_i) ] / dx_a,
    u+ = max(u,0), u- = min(u,0); scalar phi uses adv[0] on every axis.
    Evaluated in flux-difference form with the reciprocal spacing fl(1/dx_a) (one multiply per
    axis instead of two divisions per node; the product path uses the same operation order).
    Wrap-around neighbours like every other roll stencil; boundary rows are
    meaningless and overwritten by the BC fill of the time-march."""
    adv = adv_tensor(u, var)
    out = torch.zeros_like(var[0]).unsqueeze(0)
    zeros = torch.zeros_like(var[0])
    dx = mesh.dx
    for a in range(mesh.dim):
        ai = a if var.shape[0] > 1 else 0
        up = torch.max(adv[ai], zeros)
        um = torch.min(adv[ai], zeros)
        phi = var[ai]
        inv = torch.ones((), dtype=var.dtype) / dx[a]
        bwd = phi - torch.roll(phi, 1, a)
        fwd = torch.roll(phi, -1, a) - phi
        out[0] += (up * bwd + um * fwd) * inv
    if mesh.coord == "rz":  # + u phi / r, written like the central scheme's Ac row (tools.py:64-78)
        ac = _nn(2 * dx[0] / mesh.R) * torch.ones_like(var[0])
        ac = ac * adv[0]
        ac = ac / (2.0 * dx[0])
        out[0] += ac * var[0]
    return out


def euler_step(phi: Tensor, u: float | Tensor, nu: float, dt: float, mesh: OMesh,
               bcs: Sequence[OBC], limiter: str = "upwind") -> Tensor:
    """[NEW] explicit Euler step of  d(phi)/dt + div(u phi) = nu lap(phi):
    phi <- B( phi + dt * ( nu * lap(phi) - adv(phi) ) ) on the interior set,
    lap/adv from the explicit operators (edge=False) evaluated on the BC-filled phi."""
    nd = mesh.dim
    S = interior_slicer(nd, bcs)
    lap = apply_laplacian(laplacian_tables(phi, mesh, bcs), phi, nd)
    if limiter == "upwind":
        adv = div_upwind_intended(u, phi, mesh)
    elif limiter == "none":
        adv = apply_div(div_tables(u, phi, mesh, bcs, "none"), phi, nd)
    else:
        raise ValueError(limiter)
    new = phi.clone()
    for i in range(phi.shape[0]):
        new[i][S] = phi[i][S] + dt * (nu * lap[i][S] - adv[i][S])
    bc_fill(new, bcs)
    return new


# --------------------------------------------------------------------------
# analytic Poisson inputs (formulas of pyapes/testing/poisson.py:20-87)
# --------------------------------------------------------------------------
def poisson_rhs(mesh: OMesh) -> Tensor:
    g = mesh.grid
    rhs = torch.zeros(1, *mesh.nx, dtype=mesh.dtype)
    if mesh.dim == 1:
        rhs[0] = 1.0 - 2.0 * g[0] ** 2
    elif mesh.dim == 2:
        rhs[0] = 6.0 * g[0] * g[1] * (1.0 - g[1]) - 2.0 * (g[0] ** 3)
    else:
        rhs[0] = torch.sin(math.pi * g[0]) * torch.sin(math.pi * g[1]) * torch.sin(math.pi * g[2])
    return rhs


def poisson_exact(mesh: OMesh) -> Tensor:
    g = mesh.grid
    if mesh.dim == 1:
        return 7.0 / 9.0 - 2.0 / 9.0 * g[0] + g[0] ** 2 / 2.0 - g[0] ** 4 / 6.0
    if mesh.dim == 2:
        return g[1] * (1.0 - g[1]) * (g[0] ** 3)
    return (-1.0 / (3 * math.pi ** 2) * torch.sin(math.pi * g[0]) * torch.sin(math.pi * g[1])
            * torch.sin(math.pi * g[2]))


def _p1(grid, mask, *_):
    return 7.0 / 9.0 - 2.0 / 9.0 * grid[0][mask] + grid[0][mask] ** 2 / 2.0 - grid[0][mask] ** 4 / 6.0


def _p2(grid, mask, *_):
    return grid[1][mask] * (1.0 - grid[1][mask]) * (grid[0][mask] ** 3)


def poisson_cfg(dim: int) -> list[dict]:
    val: Any = _p1 if dim == 1 else (_p2 if dim == 2 else 0.0)
    return [{"bc_face": FACES[i], "bc_type": "dirichlet", "bc_val": val} for i in range(2 * dim)]


# BC values that READ THE FIELD they are given -- the reference evaluates a callable inside every BC fill with the
# current, partly filled iterate (bcs.py:200-213, 223-253).  One set of functions drives the reference
# (tests/golden/make_golden.py), this oracle and the product (tests/helpers.py), on CPU and GPU tensors alike.
def _robin_lo(grid, mask, var, *_):      # Dirichlet value tied to the first interior plane next to a LOWER face of axis 0
    return 0.4 * var[0][torch.roll(mask, 1, 0)] + 0.2 * grid[-1][mask]


def _robin_hi_flux(grid, mask, var, *_):  # Neumann flux proportional to the first interior plane next to an UPPER face of axis 0
    return 0.3 * var[0][torch.roll(mask, -1, 0)] - 0.1


def _robin_edge(grid, mask, var, *_):    # reads the plane next to a lower face of the LAST axis: sees what the axis-0 faces wrote
    return 0.25 * var[0][torch.roll(mask, 1, mask.dim() - 1)] + grid[0][mask]


def robin_cfg(dim: int) -> list[dict]:
    """faces in factory order: xl Dirichlet(var), xu Neumann(var), then Dirichlet 0 / ... , the lower face of the last
    axis Dirichlet(var) again (its corner nodes read nodes the earlier faces of the same fill have written)"""
    cfg = [{"bc_face": "xl", "bc_type": "dirichlet", "bc_val": _robin_lo},
           {"bc_face": "xu", "bc_type": "neumann", "bc_val": _robin_hi_flux}]
    for a in range(1, dim):
        lo, hi = FACES[2 * a], FACES[2 * a + 1]
        cfg.append({"bc_face": lo, "bc_type": "dirichlet", "bc_val": _robin_edge if a == dim - 1 else 0.0})
        cfg.append({"bc_face": hi, "bc_type": "dirichlet", "bc_val": 0.5})
    return cfg


# axisymmetric Poisson problem of the reference's tests/test_solver.py:309-358:
# u = exp(-z) cos(r) on Cylinder[0:1, 0:1]; rl neumann 0, the other faces dirichlet (exact values)
def _rz_ru(grid, mask, *_):
    return torch.exp(-grid[1][mask]) * math.cos(1)


def _rz_zl(grid, mask, *_):
    return torch.cos(grid[0][mask])


def _rz_zu(grid, mask, *_):
    return torch.cos(grid[0][mask]) * math.exp(-1)


def poisson_rz_cfg() -> list[dict]:
    return [{"bc_face": "rl", "bc_type": "neumann", "bc_val": 0.0},
            {"bc_face": "ru", "bc_type": "dirichlet", "bc_val": _rz_ru},
            {"bc_face": "zl", "bc_type": "dirichlet", "bc_val": _rz_zl},
            {"bc_face": "zu", "bc_type": "dirichlet", "bc_val": _rz_zu}]


def poisson_rz_rhs(mesh: OMesh) -> Tensor:
    R, Z = mesh.grid
    rhs = torch.zeros(1, *mesh.nx, dtype=mesh.dtype)
    rhs[0] = -torch.sin(R) / (R * torch.exp(Z))
    rhs[0][R.eq(0.0)] = -1.0 / torch.exp(Z[R.eq(0.0)])
    return rhs


def poisson_rz_exact(mesh: OMesh) -> Tensor:
    return torch.exp(-mesh.grid[1]) * torch.cos(mesh.grid[0])


# --------------------------------------------------------------------------
# convenience: the equation  sign*laplacian(coeff, x) == rhs  end to end
# (fdm.py:124-169 + ops.py:47-81 + linalg.py:33-71)
# --------------------------------------------------------------------------
def solve_poisson(mesh: OMesh, bc_cfg: Sequence[dict], rhs: Tensor, x0: Tensor | None = None,
                  method: str = "cg", tol: float = 1e-6, max_it: int = 1000,
                  coeff: float | None = 1.0, sign: float = 1.0, **kw):
    """NOTE: like the reference (ops.py:61-77) the caller's ``rhs`` is modified in
    place by the rhs adjustment."""
    bcs = make_bcs(mesh, bc_cfg)
    x = torch.zeros(1, *mesh.nx, dtype=mesh.dtype) if x0 is None else x0
    tabs = laplacian_tables(x, mesh, bcs)
    rhs += laplacian_rhs_adjust(x, mesh, bcs)
    terms = [OTerm("laplacian", tabs, coeff, sign)]
    fn = {"cg": cg, "bicgstab": bicgstab, "jacobi": jacobi}[method.lower()]
    return fn(x, rhs, terms, mesh, bcs, tol, max_it, **kw)
