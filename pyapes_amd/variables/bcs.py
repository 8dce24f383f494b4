"""Boundary conditions (mirrors ``pyapes/variables/bcs.py``).

A BC object here is a small record (face, type, value); the ghost / boundary-node
fill itself is ``k_bc_face`` in ``csrc/pa_bc.hip`` and is reached through
``BC.apply`` (one face) or ``HipContext.apply_bcs`` (all faces in list order, what
``linalg._apply_bc_otf`` does).  The shifted boolean masks of the reference
(bcs.py:84-95) exist as lazy properties for API compatibility only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, NamedTuple, TypedDict

import torch
from torch import Tensor

from ..backend import DType, require_gpu
from ..geometry.basis import FDIR, FDIR_RZ, d2n_coord

BC_val_type = int | float | list | Callable | Tensor | None


class BCConfig(TypedDict, total=False):
    bc_face: str
    bc_type: str
    bc_val: Any
    bc_val_opt: dict[str, Tensor] | None


def _check_val(v: Any) -> None:
    if callable(v) or v is None or isinstance(v, (int, float, list, Tensor)):
        return
    raise TypeError(f"BC: wrong bc variable -> {type(v)} is not supported!")


@dataclass
class BC:
    """One boundary face of one field."""

    bc_id: str
    bc_val: Any
    bc_val_opt: dict[str, Tensor] | None
    bc_face: str
    bc_var_name: str
    bc_coord_sys: str
    mesh_dim: int
    dtype: DType
    device: torch.device
    mesh: Any = None

    def __post_init__(self):
        _check_val(self.bc_val)
        self._bc_face_dim = d2n_coord(self.bc_coord_sys)[self.bc_face[0]]   # bcs.py:72-75
        self._bc_n_dir = -1 if self.bc_face[-1] == "l" else 1
        self._bc_type = self.__class__.__name__.lower()
        self._bc_n_vec = torch.zeros(3, dtype=self.dtype.float, device=self.device)
        self._bc_n_vec[self._bc_face_dim] = self._bc_n_dir

    # -- reference-compatible read-only views ---------------------------------
    @property
    def bc_mask(self) -> Tensor:
        return self.mesh.d_mask[self.bc_face]

    def bc_mask_shift(self, shift: int) -> Tensor:
        return torch.roll(self.bc_mask, shift, self.bc_face_dim)

    @property
    def bc_mask_prev(self) -> Tensor:
        return self.bc_mask_shift(-self.bc_n_dir)

    @property
    def bc_mask_prev2(self) -> Tensor:
        return self.bc_mask_shift(-2 * self.bc_n_dir)

    @property
    def bc_mask_forward(self) -> Tensor:
        return self.bc_mask_shift(self.bc_n_dir)

    @property
    def bc_mask_forward2(self) -> Tensor:
        return self.bc_mask_shift(2 * self.bc_n_dir)

    @property
    def bc_n_vec(self) -> Tensor:
        return self._bc_n_vec

    @property
    def bc_treat(self) -> bool:
        return self._bc_type in ("neumann", "symmetry")

    @property
    def bc_type(self) -> str:
        return self._bc_type

    @property
    def type(self) -> str:
        return self._bc_type

    @property
    def bc_face_dim(self) -> int:
        return self._bc_face_dim

    @property
    def bc_n_dir(self) -> int:
        return self._bc_n_dir

    # -- value resolution ------------------------------------------------------
    def resolve(self, var: Tensor, comp: int, for_rhs: bool = False) -> tuple[float, Tensor | None]:
        """-> (scalar, face_array|None) for component ``comp``.

        Callables are evaluated HERE, once per call site (the fused solver loops run on the
        device, so a callable is frozen for the duration of one ``solve()``); signature as in
        the reference: ``f(grid, mask, var, bc_val_opt)`` (bcs.py:203-205) or, for the rhs
        adjustment, ``f(grid, mask, var, n_vec)`` (fdc.py:806-807).  Tensor values and callable
        results are in boolean-mask gather order = C order of the face plane.
        """
        v = self.bc_val
        if callable(v):
            out = v(self.mesh.grid, self.bc_mask, var, self._bc_n_vec if for_rhs else self.bc_val_opt)
            v = out
        if isinstance(v, list):
            v = v[comp]
        if v is None:
            # bcs.py:200, 224: Dirichlet / Neumann assert a value; Symmetry / Periodic have none
            assert self._bc_type not in ("dirichlet", "neumann"), "BC: bc_val is not specified!"
            return 0.0, None
        if isinstance(v, (int, float)):
            return float(v), None
        if isinstance(v, Tensor):
            if v.numel() == 1:
                return float(v), None
            arr = v.to(device=self.device, dtype=self.dtype.float).contiguous().reshape(-1)
            return 0.0, arr
        raise TypeError(f"{self._bc_type}: bc_val must be float, int, callable, list or Tensor!")

    def depends_on_var(self, var: Tensor) -> bool:
        """Does a callable ``bc_val`` read the field it is given?  The reference calls it inside EVERY BC fill
        with the current iterate (bcs.py:203, 245 from linalg.py:125); the device solvers evaluate it once
        per solve, which is the same thing only if the result does not depend on ``var``.  Probed by calling
        it on the field and on a shifted, scaled copy."""
        v = self.bc_val
        if not callable(v):
            return False
        # probed once per (BC object, callable): two evaluations and a device round trip per face would otherwise
        # be paid by every solve of a time-stepping loop (128^2, four callable faces: 2.1 ms per solve against
        # 0.15 ms of GPU work); whether a callable reads its `var` argument is a property of the callable
        probe = getattr(self, "_dep_probe", None)
        if probe is not None and probe[0] is v:
            return probe[1]
        a = v(self.mesh.grid, self.bc_mask, var, self.bc_val_opt)
        b = v(self.mesh.grid, self.bc_mask, var * 1.5 + 0.25, self.bc_val_opt)
        a, b = torch.as_tensor(a), torch.as_tensor(b)
        res = a.shape != b.shape or not torch.equal(a.to(b.device), b)
        self._dep_probe = (v, res)
        return res

    def apply(self, var: Tensor, grid: Any, var_dim: int) -> None:
        """Fill this ONE face of component ``var_dim`` in place (bcs.py:186-194)."""
        require_gpu(var, "BC.apply")
        from ..hip.context import context_for
        context_for(self.mesh).apply_bcs(var, [self], comps=[var_dim])


class Dirichlet(BC):
    """``x[face] = g`` (bcs.py:200-213)."""


class Neumann(BC):
    """``x[face] = 4/3 x[prev] - 1/3 x[prev2] + 2/3 V |dx|`` (bcs.py:223-253)."""


class Symmetry(BC):
    """``x[face] = x[prev]`` (bcs.py:259-262)."""


class Periodic(BC):
    """lower: ``x[0] = x[1] - x[N-1] + x[N-2]``; upper: ``x[N-1] = x[0]`` (bcs.py:268-280)."""


class BCContainer(TypedDict, total=False):
    bc_type: str
    bc_val: Any
    bc_val_opt: dict[str, Tensor] | None


class BoxBoundary(NamedTuple):
    """``BoxBoundary(xl={"bc_type": "dirichlet", "bc_val": 0.4}, ...)()`` -> list of BCConfig."""

    xl: BCContainer | None = None
    xu: BCContainer | None = None
    yl: BCContainer | None = None
    yu: BCContainer | None = None
    zl: BCContainer | None = None
    zu: BCContainer | None = None

    def __call__(self) -> list[BCConfig]:
        cfg: list[BCConfig] = []
        for face in FDIR:
            d = getattr(self, face)
            if d is not None:
                cfg.append({"bc_face": face, "bc_type": d["bc_type"], "bc_val": d["bc_val"],
                            "bc_val_opt": d.get("bc_val_opt")})
        return cfg


class CylinderBoundary(NamedTuple):
    """``CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0.0}, ...)()`` (bcs.py:301-328)."""

    rl: BCContainer | None = None
    ru: BCContainer | None = None
    zl: BCContainer | None = None
    zu: BCContainer | None = None

    def __call__(self) -> list[BCConfig]:
        cfg: list[BCConfig] = []
        for face in FDIR_RZ:
            d = getattr(self, face)
            if d is not None:
                cfg.append({"bc_face": face, "bc_type": d["bc_type"], "bc_val": d["bc_val"],
                            "bc_val_opt": d.get("bc_val_opt")})
        return cfg


def mixed_bcs(bc_val: list, bc_type: list[str]) -> list[BCConfig]:
    """One entry per face in ``FDIR`` order (bcs.py:385-408)."""
    return [{"bc_face": FDIR[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
            for i, (v, t) in enumerate(zip(bc_val, bc_type))]


def homogeneous_bcs(dim: int, bc_val: Any, bc_type: str) -> list[BCConfig]:
    """Same type on all ``2*dim`` faces (bcs.py:411-440)."""
    return [{"bc_face": FDIR[i], "bc_type": bc_type,
             "bc_val": bc_val[i] if isinstance(bc_val, list) else bc_val, "bc_val_opt": None}
            for i in range(dim * 2)]


class BC_HD:
    def __new__(cls, dim: int, bc_val: float):
        return homogeneous_bcs(dim, bc_val, "dirichlet")


class BC_HN:
    def __new__(cls, dim: int, bc_val: float):
        return homogeneous_bcs(dim, bc_val, "neumann")


BC_type = Dirichlet | Neumann | Symmetry | Periodic
BC_FACTORY: dict[str, type[BC]] = {"dirichlet": Dirichlet, "neumann": Neumann,
                                   "symmetry": Symmetry, "periodic": Periodic}
