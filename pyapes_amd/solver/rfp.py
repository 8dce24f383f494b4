"""Explicit operators of the Rosenbluth-Fokker-Planck equation on axisymmetric meshes (mirrors
``pyapes/solver/rfp.py``): ``RFP().friction(jacH, pdf)``, ``RFP().diffusion(hessG, pdf)`` return tensors
shaped like the mesh; ``minmod`` / ``mc_limiter`` are the flux limiters that file defines.  The
arithmetic is ``k_rfp_friction`` / ``k_rfp_diffusion`` / ``k_limiter`` in ``csrc/pa_rfp.hip``."""
from __future__ import annotations

from torch import Tensor

from ..backend import require_gpu
from ..hip.context import context_for
from ..variables import Field
from ..variables.container import Hess, Jac


class Friction:
    """``div(grad(H) f)`` in conservative form, zero normal flux on the boundary (rfp.py:12-82)."""

    @staticmethod
    def __call__(jacH: Jac, var: Field) -> Tensor:
        if var.mesh.coord_sys != "rz":
            raise NotImplementedError("FP: Friction is only implemented for rz coordinate system.")
        require_gpu(var(), "RFP.friction")
        return context_for(var.mesh).rfp_friction(jacH.r.contiguous(), jacH.z.contiguous(), var()[0])


class Diffusion:
    """``div(D grad f)`` with ``D = hess(G)``, symmetric differences, cell-centre averages for the
    mixed term (rfp.py:85-218)."""

    @staticmethod
    def __call__(hessG: Hess, var: Field) -> Tensor:
        if var.mesh.coord_sys != "rz":
            raise NotImplementedError("FP: Diffusion is only implemented for rz coordinate system.")
        require_gpu(var(), "RFP.diffusion")
        return context_for(var.mesh).rfp_diffusion(hessG.rr.contiguous(), hessG.rz.contiguous(),
                                                   hessG.zz.contiguous(), var()[0])


class RFP:
    """``RFP().friction`` / ``RFP().diffusion`` (rfp.py:254-259)."""

    def __init__(self):
        self.friction = Friction()
        self.diffusion = Diffusion()


_LIMITER_CTX: dict = {}


def _limiter_ctx(t: Tensor):
    """limiters are plain element-wise functions of two tensors: any context of the right dtype / device"""
    require_gpu(t, "limiter")
    key = (t.device, t.dtype)
    if key not in _LIMITER_CTX:
        import torch

        from ..geometry import Box
        from ..mesh import Mesh
        mesh = Mesh(Box[0:1], None, [3], "cuda", "double" if t.dtype == torch.float64 else "single")
        _LIMITER_CTX[key] = context_for(mesh)
    return _LIMITER_CTX[key]


def minmod(a: Tensor, b: Tensor) -> Tensor:
    """rfp.py:268-286"""
    return _limiter_ctx(a).limiter(0, a, b)


def mc_limiter(a: Tensor, b: Tensor) -> Tensor:
    """monotonized-central limiter ``minmod(2 minmod(a, b), (a + b) / 2)`` (rfp.py:262-265)"""
    return _limiter_ctx(a).limiter(1, a, b)
