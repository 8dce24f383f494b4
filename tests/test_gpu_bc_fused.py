"""-m gpu: the fused closed-form BC fill (k_bc_compute / k_bc_scatter, 2 launches) must reproduce the
face-by-face fill (k_bc_face in list order, the literal reference semantics) bit for bit for every
combination of face types -- single GPU and on slab ranks with exchanged far planes."""
import itertools
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.hip.context import HipContext
from pyapes_amd.mesh import Mesh
from pyapes_amd.variables import Field

FACES = ["xl", "xu", "yl", "yu", "zl", "zu"]
AXIS_CHOICES = [
    (("periodic", None), ("periodic", None)),
    (("dirichlet", 0.25), ("neumann", 0.3)),
    (("neumann", -0.2), ("symmetry", None)),
    (("symmetry", None), ("dirichlet", 1.0)),
    (("symmetry", None), ("symmetry", None)),
    (("neumann", 0.0), ("neumann", 0.5)),
]


def _fill(bcs, slab, dtype, fused, monkeypatch):
    from helpers import hip_options
    hip_options(monkeypatch, bc_path=4 if fused else 1)     # closed form at any size / never the closed form
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, [12, 9, 11], "cuda", dtype, slab=slab)
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1, *mesh.nx), generator=g, dtype=torch.float64).to(mesh.dtype.float).cuda()
    far = [torch.randn(tuple(mesh.nx[1:]), generator=g, dtype=torch.float64).to(mesh.dtype.float).cuda()
           for _ in range(3)]
    ctx = HipContext(mesh)
    if slab:
        rank = slab[0]
        ctx.slab_set({"sums": torch.zeros(8, dtype=torch.float64, device="cuda"),
                      "bc_far_lo0": far[0] if rank == 0 else None, "bc_far_lo1": far[1] if rank == 0 else None,
                      "bc_far_hi0": far[2] if rank == 1 else None})
    ctx.bind_bcs(x, var.bcs, 0)
    ctx.apply_bc_bound(x[0])
    torch.cuda.synchronize()
    return x.cpu()


@pytest.mark.parametrize("slab", [None, (0, 2), (1, 2)], ids=["single", "slab0", "slab1"])
@pytest.mark.parametrize("dtype", ["double", "single"])
def test_fused_equals_face_by_face(slab, dtype, monkeypatch):
    bad = []
    for ax, ay, az in itertools.product(AXIS_CHOICES, repeat=3):
        bcs = [ax[0], ax[1], ay[0], ay[1], az[0], az[1]]
        a = _fill(bcs, slab, dtype, True, monkeypatch)
        b = _fill(bcs, slab, dtype, False, monkeypatch)
        if not torch.equal(a, b):
            bad.append(([t for t, _ in bcs], float((a - b).abs().max())))
    assert not bad, bad[:5]
