import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.variables import Field
from pyapes_amd.hip.context import HipContext
FACES = ["xl","xu","yl","yu","zl","zu"]
bcs = [("periodic",None),("periodic",None),("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),("symmetry",None)]
cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i,(t,v) in enumerate(bcs)]
n = [12, 9, 11]
for rank in (0, 1):
    outs = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused": os.environ["PYAPES_HIP_BC_UNFUSED"] = "1"
        else: os.environ.pop("PYAPES_HIP_BC_UNFUSED", None)
        mesh = Mesh(Box[0:1,0:1,0:0.5], None, n, "cuda", "double", slab=(rank, 2))
        var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
        g = torch.Generator().manual_seed(3)
        x = torch.randn((1, *mesh.nx), generator=g, dtype=torch.float64).cuda()
        far = [torch.randn(tuple(mesh.nx[1:]), generator=g, dtype=torch.float64).cuda() for _ in range(3)]
        ctx = HipContext(mesh)
        bufs = {"sums": torch.zeros(8, dtype=torch.float64, device="cuda"),
                "bc_far_lo0": far[0] if rank == 0 else None, "bc_far_lo1": far[1] if rank == 0 else None,
                "bc_far_hi0": far[2] if rank == 1 else None}
        ctx.slab_set(bufs)
        ctx.bind_bcs(x, var.bcs, 0)
        ctx.apply_bc_bound(x[0])
        torch.cuda.synchronize()
        outs[mode] = x.cpu().clone()
    d = (outs["fused"] - outs["unfused"]).abs()[0]
    print("rank", rank, "max diff", float(d.max()), "bad", (d > 0).nonzero()[:10].tolist())
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        x0 = torch.randn((1, *mesh.nx), generator=g, dtype=torch.float64)[0]
        fr = [torch.randn(tuple(mesh.nx[1:]), generator=g, dtype=torch.float64) for _ in range(3)]
        j = 1
        print("fused   x[0,1,10] =", float(outs["fused"][0,0,j,10]), " unfused =", float(outs["unfused"][0,0,j,10]))
        for kk in (8, 9, 10):
            print(" k", kk, "x1-f0+f1 =", float(x0[1,j,kk]-fr[0][j,kk]+fr[1][j,kk]), " x0orig", float(x0[0,j,kk]))
        a9 = x0[1,j,9]-fr[0][j,9]+fr[1][j,9]; a8 = x0[1,j,8]-fr[0][j,8]+fr[1][j,8]
        print(" neumann-like 4/3 a9 - 1/3 a8 =", float(4/3*a9 - 1/3*a8))
        print(" x1[k=9] - x[N-1 local?]..", float(x0[1,j,9]-x0[5,j,9]+x0[4,j,9]))
