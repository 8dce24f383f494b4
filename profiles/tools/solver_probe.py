"""GPU ms per iteration of fixed-iteration Jacobi / CG / BiCGSTAB solves through Solver.solve() on an n^3 fp64 mesh,
Dirichlet and fully periodic:  python profiles/tools/solver_probe.py [n] [name filter, comma separated]
(under rocprofv3 --kernel-trace --stats this gives the per-kernel table of a BiCGSTAB iteration)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs

def run(name, n, bcs, method, K, passes):
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double")
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    rhs = torch.randn_like(var())
    cfg = {"method": method, "tol": -1.0, "max_it": K - 1, "report": False}
    warm = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    sw = Solver({"fdm": dict(cfg, max_it=3)}); sw.set_eq(FDM().laplacian(1.0, warm) == rhs.clone()); sw.solve()
    s = Solver({"fdm": cfg}); s.set_eq(FDM().laplacian(1.0, var) == rhs)
    t0 = time.perf_counter(); rep = s.solve(); wall = (time.perf_counter() - t0) * 1e3
    ms = var.last_gpu_ms / rep["itr"]
    print(json.dumps({"workload": name, "ms": round(ms, 4), "wall": round(wall / rep["itr"], 4), "itr": rep["itr"],
                      "alg_TBs": round(passes * 8 * mesh.N / ms / 1e9, 3)}), flush=True)
    del mesh, var, rhs, warm, s, sw
    torch.cuda.empty_cache()

per = mixed_bcs([None] * 6, ["periodic"] * 6)
dirb = homogeneous_bcs(3, 0.0, "dirichlet")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
_run = run
def run(name, *a):
    if only is None or any(o in name for o in only): _run(name, *a)
run(f"jacobi {n}^3 f64 dirichlet", n, dirb, "jacobi", 60, 3)
run(f"jacobi {n}^3 f64 periodic", n, per, "jacobi", 60, 3)
run(f"cg {n}^3 f64 dirichlet", n, dirb, "cg", 60, 10)
run(f"cg {n}^3 f64 periodic", n, per, "cg", 60, 10)
run(f"bicgstab {n}^3 f64 dirichlet", n, dirb, "bicgstab", 40, 22)
run(f"bicgstab {n}^3 f64 periodic", n, per, "bicgstab", 40, 22)
