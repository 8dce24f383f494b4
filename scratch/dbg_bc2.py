import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.variables import Field
from pyapes_amd.hip.context import HipContext
FACES = ["xl","xu","yl","yu","zl","zu"]
def run(bcs, slab):
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i,(t,v) in enumerate(bcs)]
    n = [12, 9, 11]
    outs = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused": os.environ["PYAPES_HIP_BC_UNFUSED"] = "1"
        else: os.environ.pop("PYAPES_HIP_BC_UNFUSED", None)
        mesh = Mesh(Box[0:1,0:1,0:0.5], None, n, "cuda", "double", slab=slab)
        var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
        g = torch.Generator().manual_seed(3)
        x = torch.randn((1, *mesh.nx), generator=g, dtype=torch.float64).cuda()
        far = [torch.randn(tuple(mesh.nx[1:]), generator=g, dtype=torch.float64).cuda() for _ in range(3)]
        ctx = HipContext(mesh)
        if slab:
            rank = slab[0]
            ctx.slab_set({"sums": torch.zeros(8, dtype=torch.float64, device="cuda"),
                "bc_far_lo0": far[0] if rank == 0 else None, "bc_far_lo1": far[1] if rank == 0 else None,
                "bc_far_hi0": far[2] if rank == 1 else None})
        ctx.bind_bcs(x, var.bcs, 0)
        ctx.apply_bc_bound(x[0])
        torch.cuda.synchronize()
        outs[mode] = x.cpu().clone()
    d = (outs["fused"] - outs["unfused"]).abs()[0]
    return float(d.max()), (d > 0).nonzero()[:4].tolist()
PE=("periodic",None); SY=("symmetry",None)
for name, bcs in {"zuSY": [PE,PE,("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),SY],
                  "zuN": [PE,PE,("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),("neumann",0.3)],
                  "zuD": [PE,PE,("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),("dirichlet",0.3)],
                  "zlSY": [PE,PE,("dirichlet",0.0),("dirichlet",1.0),SY,("neumann",0.3)],
                  "yuSY": [PE,PE,("dirichlet",0.0),SY,("neumann",0.0),("dirichlet",0.3)],
                  "xN_zuSY": [("neumann",0.1),("neumann",0.2),("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),SY]}.items():
    for slab in (None, (0,2), (1,2)):
        print(name, slab, run(bcs, slab))
