"""Diagnostic (GPU box): BiCGSTAB on config 3's mesh (512^3 fp64, fully periodic) -- the stop-test value after K iterations
on one GPU, on one GPU with ONE right-hand-side entry moved by an ulp, and on 2 slabs (hostring stand-in).  Says whether a
slab / whole-mesh difference is the order of the sums (then the ulp run moves as much) or a defect (then it does not).
    python3 profiles/tools/slab_bicg_diag.py [n] [K,K,...] [method] [ranks]
"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
warnings.filterwarnings("ignore")

import torch  # noqa: E402


def whole(n, K, method, ulp=False):
    import bench
    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.solver.fdm import FDM
    from pyapes_amd.solver.ops import Solver
    from pyapes_amd.variables import Field
    mesh = Mesh(Box([0.0] * 3, [1.0] * 3), None, list(n), "cuda", "double")
    var = Field("p", 1, mesh, {"domain": bench.make_bcs("periodic"), "obstacle": None})
    rhs = bench.synth_rhs(n, 0, n[0], os.environ.get("PA_TEST_SLAB_RHS_KIND", "periodic"), mesh.dtype.float, mesh.device)
    if ulp:
        v = rhs[0, n[0] // 3, n[1] // 3, n[2] // 3]
        rhs[0, n[0] // 3, n[1] // 3, n[2] // 3] = torch.nextafter(v, v + 1)
    solver = Solver({"fdm": {"method": method, "tol": 1e-30, "max_it": K, "report": False}})
    solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    rep = solver.solve()
    return var().clone(), float(rep["tol"]), int(rep["itr"])


def main():
    import tempfile
    from test_gpu_slab_ranks import _worker_config3
    from test_slab_gloo import spawn_ranks
    n = (int(sys.argv[1]),) * 3 if len(sys.argv) > 1 else (512, 512, 512)
    Ks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 6]
    method = sys.argv[3] if len(sys.argv) > 3 else "bicgstab"
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    for K in Ks:
        x1, t1, i1 = whole(n, K, method)
        xu, tu, iu = whole(n, K, method, ulp=True)
        d = tempfile.mkdtemp()
        spawn_ranks(_worker_config3, lambda port: (W, port, n, K, method, d), W)
        parts = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(W)]
        xs = torch.cat([p["x"] for p in parts], dim=1).cuda()
        rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
        print(f"{method} n={n[0]} K={K}: tol whole {t1:.16e} ulp {tu:.16e} slabs {parts[0]['tol']:.16e} | itr {i1} {iu} {parts[0]['itr']} | "
              f"rel x: ulp {rel(xu, x1):.3e} slabs {rel(xs, x1):.3e} | dtol: ulp {abs(tu - t1) / abs(t1):.3e} slabs {abs(parts[0]['tol'] - t1) / abs(t1):.3e}",
              flush=True)


if __name__ == "__main__":
    main()
