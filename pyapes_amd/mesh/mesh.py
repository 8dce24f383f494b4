"""Equidistant node-based box mesh (mirrors ``pyapes/mesh/_mesh.py:19-117``).

Differences that matter at 512^3 and beyond: ``grid`` is a tuple of broadcast
views (no n^3 coordinate arrays) and the face masks are materialised lazily --
the HIP kernels take face ids, not boolean masks.  ``slab=(rank, world)`` (new)
makes this rank's mesh one slab of the global box along axis 0.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ..backend import DTYPE_DOUBLE, DTYPE_SINGLE, TORCH_DEVICE, DType, TorchDevice
from ..geometry import GeoTypeIdentifier
from ..geometry.basis import Geometry, d2n_coord


class _LazyMasks(dict):
    """``d_mask[face]`` -> bool tensor of the whole boundary plane, built on first use."""

    def __init__(self, mesh: "Mesh"):
        super().__init__()
        self._mesh = mesh
        self._faces = [c["face"] for c in mesh.domain.config.values()]

    def __missing__(self, face: str) -> Tensor:
        if face not in self._faces:
            raise KeyError(face)
        m = self._mesh
        mask = torch.zeros(*m.nx, dtype=torch.bool, device=m.device)
        idx: list = [slice(None)] * m.dim
        a = m.d_mask_dim(face)
        if m.owns_face(face):
            idx[a] = 0 if face[1] == "l" else m.nx[a] - 1
            mask[tuple(idx)] = True
        self[face] = mask
        return mask

    def __iter__(self):
        return iter(self._faces)

    def keys(self):  # type: ignore[override]
        return list(self._faces)

    def __len__(self) -> int:
        return len(self._faces)


class Mesh:
    """``Mesh(Box[0:1, 0:1], None, [128, 128], "cuda", "double")``.

    Args:
        domain: ``Box`` or ``Cylinder`` geometry.
        obstacle: must be ``None`` (obstacles raise downstream in the reference too,
            linalg.py:287-292).
        spacing: node counts (ints) or spacings (floats) per axis (_mesh.py:67-80).
        device: "cpu" | "cuda".
        dtype: "double" | "single".
        slab: optional ``(rank, world_size)``: this mesh is rank's slab of the global box
            along axis 0 (3-D only); ``nx`` is then the LOCAL node count.
    """

    def __init__(self, domain: Geometry, obstacle: Optional[list[Geometry]],
                 spacing: list[int] | list[float] = [], device: str = "cpu",
                 dtype: str | int = "double", slab: tuple[int, int] | None = None):
        assert device in TORCH_DEVICE, "Mesh: device only accept cpu or cuda"
        self.device = TorchDevice(device).device
        if self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        assert dtype in DTYPE_DOUBLE or dtype in DTYPE_SINGLE, "Mesh: dtype only accept double or single"
        self.dtype = DType(dtype)
        self.domain = domain
        if domain.type not in ("box", "cylinder"):
            raise TypeError(f"Mesh: domain type ({domain.type=}) not identifiable")
        if self.coord_sys == "rz":
            assert self.dim == 2, "Mesh: rz coordinate system only accept 2D domain"
        self.obstacle = obstacle
        f = self.dtype.float
        self._lower = torch.tensor(domain.lower, dtype=f, device=self.device)
        self._upper = torch.tensor(domain.upper, dtype=f, device=self.device)
        self._lx = self._upper - self._lower
        lx_host = (torch.tensor(domain.upper, dtype=f) - torch.tensor(domain.lower, dtype=f))
        if int in GeoTypeIdentifier(spacing):
            self._gnx = [int(s) for s in spacing]
            self._dx = [float(l / (n - 1.0)) for l, n in zip(lx_host, self._gnx)]
        elif float in GeoTypeIdentifier(spacing):
            self._dx = [float(s) for s in spacing]
            self._gnx = [int(l / d + 1.0) for l, d in zip(lx_host, self._dx)]
        else:
            raise TypeError("Mesh: spacing only accept int or float")

        # global node coordinates, exactly torch.linspace in the mesh dtype (_mesh.py:84-93)
        gx = [torch.linspace(float(torch.tensor(domain.lower[i], dtype=f)),
                             float(torch.tensor(domain.upper[i], dtype=f)), self._gnx[i], dtype=f)
              for i in range(self.dim)]
        self._gx_host = gx
        self.slab = None
        self.i_off = 0
        self._nx = list(self._gnx)
        if slab is not None:
            rank, world = slab
            if self.dim != 3:
                raise ValueError("Mesh: slab decomposition is for 3-D meshes (1-D/2-D run as replicas)")
            from ..slab import slab_extent
            self.i_off, n_loc = slab_extent(self._gnx[0], rank, world)
            self._nx[0] = n_loc
            self.slab = (rank, world)
        self.x = [g.to(self.device) for g in gx]
        if slab is not None:
            self.x[0] = self.x[0][self.i_off:self.i_off + self._nx[0]]
        self._grid: tuple[Tensor, ...] | None = None
        self.d_mask = _LazyMasks(self)
        self.o_mask: dict = {}
        self._hip = None

    # -- geometry ---------------------------------------------------------
    @property
    def coord_sys(self) -> str:
        """"xyz" (Box) or "rz" (Cylinder: axis 0 = r, axis 1 = z) (_mesh.py:121-131)."""
        return "rz" if self.domain.type == "cylinder" else "xyz"

    @property
    def R(self) -> Tensor:
        if self.coord_sys != "rz":
            raise KeyError("Mesh: R coordinate only available in axisymmetric case.")
        return self.grid[0]

    @property
    def dim(self) -> int:
        return self.domain.dim

    @property
    def grid(self) -> tuple[Tensor, ...]:
        """``torch.meshgrid(self.x, indexing="ij")`` (broadcast views)."""
        if self._grid is None:
            self._grid = torch.meshgrid(self.x, indexing="ij")
        return self._grid

    @property
    def t_mask(self) -> Tensor:
        m = torch.zeros(*self.nx, dtype=torch.bool, device=self.device)
        for f in self.d_mask:
            m = torch.logical_or(m, self.d_mask[f])
        return m

    def owns_face(self, face: str) -> bool:
        """False only on slab ranks that do not hold that global axis-0 boundary plane."""
        if self.slab is None or self.d_mask_dim(face) != 0:
            return True
        if face[1] == "l":
            return self.i_off == 0
        return self.i_off + self._nx[0] == self._gnx[0]

    @property
    def X(self) -> Tensor:
        return self.grid[0]

    @property
    def Y(self) -> Tensor:
        if self.coord_sys == "rz" or self.dim < 2:   # _mesh.py:206-222
            return torch.tensor([], dtype=self.dtype.float, device=self.device)
        return self.grid[1]

    @property
    def Z(self) -> Tensor:
        if self.coord_sys == "rz":                   # _mesh.py:224-238: the second axis of an rz mesh
            return self.grid[1]
        return self.grid[2] if self.dim > 2 else torch.tensor([], dtype=self.dtype.float, device=self.device)

    @property
    def N(self) -> int:
        n = 1
        for v in self._nx:
            n *= v
        return n

    @property
    def size(self) -> float:
        return self.domain.size

    @property
    def lx(self) -> Tensor:
        return self._lx

    @property
    def dx(self) -> Tensor:
        return torch.tensor(self._dx, dtype=self.dtype.float, device=self.device)

    @property
    def dx_list(self) -> list[float]:
        """Spacing as python floats, each exactly representable in the mesh dtype."""
        return list(self._dx)

    @property
    def nx(self) -> torch.Size:
        return torch.Size(self._nx)

    @property
    def global_nx(self) -> torch.Size:
        return torch.Size(self._gnx)

    @property
    def lower(self) -> Tensor:
        return self._lower

    @property
    def upper(self) -> Tensor:
        return self._upper

    @property
    def center(self) -> Tensor:
        return self.lx * 0.5

    @property
    def is_cuda(self) -> bool:
        return self.device.type == "cuda"

    @property
    def dg(self) -> list[Tensor]:
        """Half the sum of the forward and backward node distances, one-sided on the boundary
        (`_mesh.py:262-293`): dx in the interior, dx/2 on the boundary nodes."""
        out = []
        for a, g in enumerate(self.grid):
            fw = torch.roll(g, -1, a) - g
            bw = g - torch.roll(g, 1, a)
            fw = torch.where(fw < 0, torch.zeros_like(fw), fw)
            bw = torch.where(bw < 0, torch.zeros_like(bw), bw)
            out.append((fw + bw) / 2)
        return out

    def face_dxf(self, face: str) -> float:
        """``grid[face] - grid[prev]`` of ``Neumann.apply`` (bcs.py:228-231): the literal
        difference of the two outermost node coordinates, in the mesh dtype."""
        g = self._gx_host[self.d_mask_dim(face)]
        return float(g[0] - g[1]) if face[1] == "l" else float(g[-1] - g[-2])

    def d_mask_dim(self, d_face: str) -> int:
        return d2n_coord(self.coord_sys)[d_face[0]]

    def d_mask_dir(self, d_face: str) -> int:
        """_mesh.py:144-147, literal: ``1 if d_face[1] == "r" else -1`` -- the faces are named ``l`` / ``u``, so this is
        -1 for every face the package ever names (the reference's docstring example speaks of "xr")."""
        return 1 if d_face[1] == "r" else -1

    def d_mask_shift(self, d_face: str, shift: int):
        """_mesh.py:149-175: the face mask rolled by ``-shift * d_mask_dir`` along the face's axis.  With the literal
        direction above that is ``+shift`` for BOTH sides: a lower face's mask moves inwards, an upper face's mask wraps
        round to the low end (kept as is; the BC classes of the reference do their own, correct rolls, bcs.py:84-95)."""
        return torch.roll(self.d_mask[d_face], -shift * self.d_mask_dir(d_face), self.d_mask_dim(d_face))

    def face_index(self, face: str) -> int:
        """0..5 = lower / upper face of mesh axis 0, 1, 2 (the C ABI's face id)"""
        return 2 * self.d_mask_dim(face) + (0 if face[1] == "l" else 1)


    def __repr__(self) -> str:
        return f"{self.domain} with dx={self._dx}"
