import os, sys, warnings, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pyapes_oracle as O
from test_slab_gloo import CASES, _free_port
from test_gpu_slab import _worker
if __name__ == "__main__":
    name, n, K, dtype = "xper", (12, 9, 11), int(sys.argv[1]) if len(sys.argv) > 1 else 0, "double"
    out = "/tmp/x.pt"
    mp.spawn(_worker, args=(2, _free_port(), name, n, K, dtype, out), nprocs=2, join=True)
    res = torch.load(out)
    mesh = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(CASES[name])]
    g = torch.Generator().manual_seed(7)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    warnings.simplefilter("ignore")
    xo, ro = O.solve_poisson(mesh, cfg, rhs, method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    d = (res["x"] - xo).abs()[0]
    print("itr", res["itr"], ro["itr"], "tol", res["tol"], ro["tol"], "max diff", float(d.max()))
    idx = (d > 1e-12).nonzero()
    print("num bad", len(idx), "first", idx[:12].tolist())
    X = res["x"][0]
    print("fused  x[0,1:4,9] ", X[0,1:4,9].tolist())
    print("fused  x[0,1:4,10]", X[0,1:4,10].tolist())
    print("oracle x[0,1:4,9] ", xo[0][0,1:4,9].tolist())
    print("oracle x[0,1:4,10]", xo[0][0,1:4,10].tolist())
    print("x[1,1:4,10]-x[11]+x[10] (k=10):", (X[1,1:4,10]-X[11,1:4,10]+X[10,1:4,10]).tolist())
