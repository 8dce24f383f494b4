"""Explicit time marching (new; the reference's ``Ddt`` is a stub).

``euler_step(phi, u, nu, dt, fdm_config)``:  phi <- B( phi + dt * ( nu * lap(phi) - div(u phi) ) )
on the interior set, lap / div being the explicit operators (edge=False) evaluated on the current,
BC-filled phi.  One fused kernel (``k_euler``) + the ordered BC fill.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from ..backend import require_gpu
from ..hip.context import context_for
from ..variables import Field
from .fdc import _adv_of, div_kind


def _march_on_slabs(phi: Field, u: Any, nu: float, dt: float, nsteps: int, kind: int) -> Field:
    """``Mesh(..., slab=(rank, world))``: the same call on every rank of the process group (pyapes_amd/slab.py SlabEuler)."""
    import torch.distributed as dist

    from ..slab import SlabEuler
    if not dist.is_initialized():
        raise RuntimeError("pyapes_amd: a march on a slab mesh needs torch.distributed to be initialised "
                           "(one process per GPU, backend 'nccl' = RCCL)")
    backend = context_for(phi.mesh)
    if not getattr(backend, "is_standin", False):
        require_gpu(phi(), "euler_march")
    if not phi().is_contiguous():
        phi.set_var_tensor(phi().contiguous())
    adv = _adv_of(u, phi)
    final = SlabEuler(phi.mesh, phi, dist, backend=backend).march(kind, adv, nu, dt, nsteps)
    if final.data_ptr() != phi()[0].data_ptr():
        phi.set_var_tensor(final.unsqueeze(0))
    return phi


def euler_step(phi: Field, u: float | Tensor | Field, nu: float, dt: float,
               config: dict | None = None) -> Field:
    """Advance ``phi`` in place by one explicit Euler step; returns ``phi``."""
    if getattr(phi.mesh, "slab", None) is not None:
        cfg = (config or {}).get("div", {"limiter": "upwind"})
        return _march_on_slabs(phi, u, nu, dt, 1, div_kind(cfg.get("limiter", "upwind").lower(), bool(cfg.get("compat", False))))
    require_gpu(phi(), "euler_step")
    if phi.dim != 1:
        raise NotImplementedError("pyapes_amd: euler_step is for scalar fields")
    cfg = (config or {}).get("div", {"limiter": "upwind"})
    kind = div_kind(cfg.get("limiter", "upwind").lower(), bool(cfg.get("compat", False)))
    ctx = context_for(phi.mesh)
    ctx.bind_bcs(phi(), phi.bcs, 0)
    out = torch.empty_like(phi())
    ctx.euler_step(phi()[0], out[0], kind, _adv_of(u, phi), nu, dt)
    phi.set_var_tensor(out)
    return phi


def euler_march(phi: Field, u: float | Tensor | Field, nu: float, dt: float, nsteps: int,
                config: dict | None = None) -> Field:
    """``nsteps`` explicit Euler steps with no host work in between (the whole march is enqueued by one
    C-ABI call: fused step kernel + ordered BC fill per step, ping-pong buffers)."""
    if phi.dim != 1:
        raise NotImplementedError("pyapes_amd: euler_march is for scalar fields")
    cfg = (config or {}).get("div", {"limiter": "upwind"})
    kind = div_kind(cfg.get("limiter", "upwind").lower(), bool(cfg.get("compat", False)))
    if getattr(phi.mesh, "slab", None) is not None:
        phi = _march_on_slabs(phi, u, nu, dt, nsteps, kind)
        if hasattr(phi, "_t"):
            phi.update_time(dt * nsteps)
        return phi
    require_gpu(phi(), "euler_march")
    ctx = context_for(phi.mesh)
    ctx.bind_bcs(phi(), phi.bcs, 0)
    if not phi().is_contiguous():
        phi.set_var_tensor(phi().contiguous())
    tmp = torch.empty_like(phi())
    final = ctx.euler_march(phi()[0], tmp[0], kind, _adv_of(u, phi), nu, dt, nsteps)
    if final.data_ptr() == tmp[0].data_ptr():
        phi.set_var_tensor(tmp)
    if hasattr(phi, "_t"):
        phi.update_time(dt * nsteps)
    return phi
