import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.variables import Field
from pyapes_amd.hip.context import HipContext
FACES = ["xl","xu","yl","yu","zl","zu"]
PE=("periodic",None); SY=("symmetry",None)
bcs = [PE,PE,("dirichlet",0.0),("dirichlet",1.0),("neumann",0.0),SY]
cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i,(t,v) in enumerate(bcs)]
mesh = Mesh(Box[0:1,0:1,0:0.5], None, [12,9,11], "cuda", "double", slab=(0,2))
var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
g = torch.Generator().manual_seed(3)
x = torch.randn((1, *mesh.nx), generator=g, dtype=torch.float64).cuda()
far = [torch.randn(tuple(mesh.nx[1:]), generator=g, dtype=torch.float64).cuda() for _ in range(3)]
print("expect raw1", float(x[0,1,1,9]), "far0", float(far[0][1,9]), "far1", float(far[1][1,9]))
ctx = HipContext(mesh)
ctx.slab_set({"sums": torch.zeros(8, dtype=torch.float64, device="cuda"), "bc_far_lo0": far[0], "bc_far_lo1": far[1], "bc_far_hi0": None})
ctx.bind_bcs(x, var.bcs, 0)
ctx.apply_bc_bound(x[0])
torch.cuda.synchronize()
print("result", float(x[0,0,1,10]), float(x[0,0,1,9]))
