"""``Field``: a ``(dim, *nx)`` tensor + its mesh + its ordered BC list
(mirrors ``pyapes/variables/fields.py:20-422``)."""
from __future__ import annotations

import copy
from typing import Any

import torch
from torch import Tensor

from ..mesh import Mesh
from .bcs import BC_FACTORY, BC, BCConfig
from .fieldops import FieldArithmetic


class Field(FieldArithmetic):
    """``Field("p", 1, mesh, {"domain": bcs, "obstacle": None}, init_val=0.0)``."""

    def __init__(self, name: str, dim: int, mesh: Mesh,
                 bc_config: dict[str, list[BCConfig] | None] | None,
                 init_val: Any = None, object_interp: bool = False):
        self.name = name
        self.dim = dim
        self.mesh = mesh
        self.bc_config = bc_config
        self.init_val = init_val
        self.object_interp = object_interp
        self._VAR = torch.zeros(dim, *mesh.nx, dtype=mesh.dtype.float, device=mesh.device)
        if init_val is not None:
            if isinstance(init_val, float):
                self._VAR += init_val
            elif isinstance(init_val, list):
                assert dim == len(init_val), "Field: init_val should match with Field dimension!"
                for d in range(dim):
                    if isinstance(init_val[d], (float, Tensor)):
                        self._VAR[d] += init_val[d]
                    else:
                        raise ValueError(f"Field: {type(init_val[d])} is an unsupported init_val type!")
            elif isinstance(init_val, Tensor):
                assert dim == init_val.size(0), "Field: init_val should match with Field dimension!"
                for d in range(dim):
                    self._VAR[d] += init_val[d].to(self._VAR.device)
            elif isinstance(init_val, str) and init_val.lower() == "random":
                self._VAR = torch.rand_like(self._VAR)
            else:
                raise ValueError("Field: unsupported data type!")
        if self.bc_config is not None:
            if "domain" not in self.bc_config:
                raise ValueError("Field: domain must be defined!")
            self.bc_config.setdefault("obstacle", None)
        self.set_bcs()

    # -- time stamps (fields.py:109-145) ---------------------------------------
    def set_time(self, dt: float, init_val: float | None = None) -> None:
        self._t = init_val if init_val is not None else 0.0
        self._dt = dt

    def update_time(self, dt: float | None = None) -> None:
        self._t += self._dt if dt is None else dt

    @property
    def t(self) -> float:
        return self._t

    @property
    def dt(self) -> float:
        return self._dt

    def save_old(self) -> None:
        self._VARo = self._VAR.clone()
        self._VARo_stale = None

    @property
    def VARo(self) -> Tensor:
        """fields.py:129-136.  After a device solve without ``save_old`` the reference would hold the
        iterate before the last iteration here; rather than hand out something else, reading it raises."""
        why = getattr(self, "_VARo_stale", None)
        if why:
            raise RuntimeError(f"pyapes_amd: Field.VARo of '{self.name}' is not available: {why}")
        return self._VARo

    @VARo.setter
    def VARo(self, other: Tensor) -> None:
        self._VARo = other
        self._VARo_stale = None

    def mark_old_stale(self, why: str) -> None:
        self._VARo_stale = why

    # -- container -------------------------------------------------------------
    @property
    def mesh_axis(self) -> list[int]:
        return [i + 1 for i in range(self.mesh.dim)]

    @property
    def dx(self) -> Tensor:
        return self.mesh.dx

    @property
    def nx(self) -> torch.Size:
        return self.mesh.nx

    @property
    def VAR(self) -> Tensor:
        return self._VAR

    @VAR.setter
    def VAR(self, other: Tensor) -> None:
        self._VAR = other

    @property
    def size(self) -> torch.Size:
        return self._VAR.size()

    def copy(self, name: str | None = None) -> "Field":
        """New field with the same mesh (shared, immutable) and BC list and a cloned tensor."""
        new = copy.copy(self)
        new._VAR = self._VAR.clone()
        new.bcs = list(self.bcs)
        if name is not None:
            new.name = name
        return new

    def zeros_like(self, name: str | None = None) -> "Field":
        new = self.copy(name)
        new._VAR = torch.zeros_like(self._VAR)
        return new

    def zeros_like_tensor(self) -> Tensor:
        return torch.zeros_like(self._VAR)

    def sum(self, dim: int = 0) -> Tensor:
        return torch.sum(self._VAR, dim=dim)

    def set_var_tensor(self, val: Tensor, insert: int | None = None) -> "Field":
        """Rebind (same shape) or broadcast-assign per component (fields.py:209-235)."""
        if self.size == val.shape:
            self._VAR = val
        else:
            for i in range(self.dim):
                if insert is None or i == insert:
                    self._VAR[i] = val
        return self

    def __getitem__(self, idx: int | slice) -> Tensor:
        return self._VAR if isinstance(idx, slice) else self._VAR[idx]

    def __setitem__(self, idx: int | slice, val: Tensor) -> None:
        if isinstance(idx, slice):
            self._VAR = val
        else:
            self._VAR[idx] = val

    def __call__(self) -> Tensor:
        return self._VAR

    def volume_integral(self, target: Tensor | None = None) -> Tensor:
        if target is None:
            target = torch.ones_like(self._VAR[0])
        val = torch.zeros(self.dim, device=self._VAR.device, dtype=self._VAR.dtype)
        for i in range(self.dim):
            val[i] = torch.sum(target * self._VAR[i] * self.mesh.dx.prod())
        return val

    # -- BCs (fields.py:361-422) -------------------------------------------------
    def get_bc(self, bc_id: str) -> BC | None:
        found = [bc for bc in self.bcs if bc.bc_id == bc_id]
        if len(found) > 1:
            raise KeyError(f"Field: bc_id {bc_id} returned multiple bcs. Check id once again!")
        return found[0] if found else None

    def set_bcs(self) -> None:
        self.bcs: list[BC] = []
        if self.bc_config is None:
            return
        if self.bc_config["domain"] is not None:
            d_bc = self.bc_config["domain"]
            n_faces = len(self.mesh.domain.config)
            assert n_faces == len(d_bc), \
                f"Field: domain config ({n_faces}) mismatch with bc config ({len(d_bc)})!"
            for bc in d_bc:
                face = bc["bc_face"]
                self.bcs.append(BC_FACTORY[str(bc["bc_type"])](
                    bc_id=f"d-{face}", bc_val=bc["bc_val"], bc_val_opt=bc.get("bc_val_opt"),
                    bc_face=face, bc_var_name=self.name, bc_coord_sys=self.mesh.coord_sys,
                    mesh_dim=self.mesh.dim, dtype=self.mesh.dtype, device=self.mesh.device,
                    mesh=self.mesh))
        if self.mesh.obstacle is not None and self.bc_config.get("obstacle") is not None:
            raise NotImplementedError

    def apply_bcs(self) -> "Field":
        """All faces in list order, every component (what linalg._apply_bc_otf does)."""
        from ..hip.context import context_for
        if self.bcs:
            context_for(self.mesh).apply_bcs(self._VAR, self.bcs)
        return self
