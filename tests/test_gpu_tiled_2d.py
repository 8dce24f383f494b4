"""-m gpu: 2-D meshes on the tiled kernels (a 2-D mesh is one plane of the 3-D tiling): CG fast path vs
generic vs oracle, and A x / Jacobi / Euler bit-exact against the generic kernels."""
import warnings

import pytest
import torch

import pyapes_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdc import FDC
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.march import euler_step
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

D = lambda v=0.0: ("dirichlet", v)   # noqa: E731
N = lambda v=0.0: ("neumann", v)     # noqa: E731
SY = ("symmetry", None)
PE = ("periodic", None)
BCS = {"mix": [D(0.2), N(0.5), N(-0.3), D(1.0)], "sym": [SY, N(0.1), D(0.0), SY], "per": [PE] * 4,
       "xper": [PE, PE, D(0.0), N(0.2)]}
SHAPES = [((200, 260), "double"), ((37, 132), "double"), ((64, 256), "single"), ((19, 24), "single"),
          ((201, 257), "double"), ((37, 131), "double"), ((64, 255), "single"), ((19, 26), "single")]


def _cfg(bcs):
    return [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]


@pytest.mark.parametrize("bc", list(BCS), ids=list(BCS))
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[0])) + s[1][0])
def test_2d_fast_generic_oracle(shape, bc, monkeypatch):
    n, dtype = shape
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(21)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    if bc == "per":
        rhs0 -= rhs0.mean()
    K = 7
    res = {}
    for fast in (True, False):
        monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
        mesh = Mesh(Box[0:1, 0:0.7], None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
        solver = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": K, "report": False}})
        solver.set_eq(-FDM().laplacian(0.7, var) == rhs0.cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
            out = {"cg": var().cpu(), "rep": rep}
            v2 = Field("q", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
            v2.set_var_tensor(x0.cuda().clone())
            v2.apply_bcs()
            out["lap"] = FDC({"laplacian": {"edge": False}}).laplacian(v2).cpu()
            out["grad"] = FDC({"grad": {"edge": False}}).grad(v2).cpu()
            for _ in range(3):
                euler_step(v2, 0.9, 1e-3, 1e-3, {"div": {"limiter": "upwind"}})
            out["euler"] = v2().cpu()
            v3 = Field("j", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
            s2 = Solver({"fdm": {"method": "jacobi", "tol": 1e-30, "max_it": 10, "report": False, "omega": 0.9}})
            s2.set_eq(FDM().laplacian(1.0, v3) == rhs0.cuda().clone())
            out["jac_rep"] = s2.solve()
            out["jac"] = v3().cpu()
        res[fast] = out
    f, gk = res[True], res[False]
    assert f["rep"]["itr"] == gk["rep"]["itr"] == K + 1
    assert rel_err(f["cg"], gk["cg"]) <= (1e-12 if dtype == "double" else 2e-5)
    for k in ("lap", "grad", "euler", "jac"):
        assert torch.equal(f[k], gk[k]), k
    assert abs(f["jac_rep"]["tol"] - gk["jac_rep"]["tol"]) <= 1e-6 * abs(gk["jac_rep"]["tol"])
    om = O.OMesh([0, 0], [1, 0.7], list(n), dtype)
    orc = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(BCS[bc])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(om, orc, rhs0.clone(), method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    assert ro["itr"] == f["rep"]["itr"]
    assert rel_err(f["cg"], xo) <= (1e-10 if dtype == "double" else 1e-5)
