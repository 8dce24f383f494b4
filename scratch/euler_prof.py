import os, sys, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
warnings.simplefilter("ignore")
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.march import euler_march
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import mixed_bcs
n = 256
mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "single")
bcs = mixed_bcs([0.0, 0.0, None, None, None, None], ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
phi = Field("phi", 1, mesh, {"domain": bcs, "obstacle": None})
phi.set_var_tensor(torch.exp(-((mesh.X - 0.5) ** 2 + (mesh.Y - 0.5) ** 2 + (mesh.Z - 0.5) ** 2) / 0.02).unsqueeze(0).contiguous())
phi.apply_bcs()
euler_march(phi, 1.0, 1e-3, 1e-4, 100, {"div": {"limiter": "upwind"}})
torch.cuda.synchronize()
