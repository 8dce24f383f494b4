"""-m gpu: the folded launch sequence (scalar steps in the prologue of the next tiled kernel, DESIGN.md §4 "small meshes") against the launch sequence it replaces
(PYAPES_HIP_FOLD=0): same bits in the iterate, same iteration count, same
tolerance -- for CG, Jacobi and BiCGSTAB, dozens of iterations (so that batches, polls and the flush of a
pending step all happen), random extents / face types / dtypes, and a stop inside a batch."""
import os
import random
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

def _faces():
    import pyapes_oracle as O
    return O.FACES


def _solve(monkeypatch, folded, n, bcs, dtype, method, rhs, x0, tol, max_it, adv):
    monkeypatch.setenv("PYAPES_HIP_FOLD", "1" if folded else "0")
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0 + 0.1 * a for a in range(nd)]), None, list(n), "cuda", dtype)
    cfg = [{"bc_face": _faces()[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    var.set_var_tensor(x0.cuda().clone())
    s = Solver({"fdm": {"method": method, "tol": tol, "max_it": max_it, "report": False}})
    fdm = FDM({"div": {"limiter": "upwind", "edge": False}})
    if adv:
        s.set_eq(fdm.div(0.6, var) - fdm.laplacian(0.05, var) == rhs.cuda().clone())
    else:
        s.set_eq(-fdm.laplacian(0.8, var) == rhs.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep


def _case(rng):
    nd = rng.choice([2, 3, 3])
    if nd == 2:
        n = [rng.choice([9, 16, 33, 64, 100, 129]), rng.choice([8, 17, 64, 65, 128, 130])]
    else:
        n = [rng.choice([5, 8, 12, 17, 24, 33]), rng.choice([6, 9, 16, 20, 40]), rng.choice([8, 17, 32, 33, 64, 66])]
    bcs = []
    for a in range(nd):
        if rng.random() < 0.2:
            bcs += [("periodic", None), ("periodic", None)]
        else:
            for _ in range(2):
                t = rng.choice(["dirichlet", "dirichlet", "neumann", "symmetry"])
                bcs.append((t, None if t == "symmetry" else round(rng.uniform(-1, 1), 3)))
    if not any(t == "dirichlet" for t, _ in bcs):
        bcs[0] = ("dirichlet", 0.25)
        if bcs[1][0] == "periodic":
            bcs[1] = ("dirichlet", -0.5)
    dtype = "double" if rng.random() < 0.7 else "single"
    return n, bcs, dtype


@pytest.mark.parametrize("method", ["cg", "jacobi", "bicgstab"])
def test_folded_sequence_is_bit_identical(monkeypatch, method):
    ncases = int(os.environ.get("PYAPES_FUZZ_CASES", "40"))   # soak runs: PYAPES_FUZZ_CASES=1000
    rng = random.Random(7 + len(method) + ncases)
    checked = 0
    for case in range(ncases):
        n, bcs, dtype = _case(rng)
        tdt = torch.float64 if dtype == "double" else torch.float32
        g = torch.Generator().manual_seed(1000 + case)
        rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
        x0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
        adv = method == "bicgstab" and case % 2 == 0
        # every third case stops on the tolerance somewhere inside a batch of enqueued iterations
        tol = -1.0 if case % 3 else (1e-3 if dtype == "double" else 1e-2)
        max_it = 37 if case % 3 else 400
        try:
            xa, ra = _solve(monkeypatch, True, n, bcs, dtype, method, rhs, x0, tol, max_it, adv)
        except RuntimeError:
            with pytest.raises(RuntimeError):   # non-finite stop test: both sequences must raise
                _solve(monkeypatch, False, n, bcs, dtype, method, rhs, x0, tol, max_it, adv)
            continue
        xb, rb = _solve(monkeypatch, False, n, bcs, dtype, method, rhs, x0, tol, max_it, adv)
        assert ra["itr"] == rb["itr"], (case, n, bcs, dtype, ra, rb)
        assert ra["tol"] == rb["tol"] or (ra["tol"] != ra["tol"] and rb["tol"] != rb["tol"]), (case, n, bcs, ra, rb)
        assert torch.equal(xa, xb), (case, n, bcs, dtype, float((xa - xb).abs().max()))
        checked += 1
    assert checked >= ncases // 2


def test_rhs_adjust_on_the_neumann_layers_only(monkeypatch):
    """The rhs adjustment touches the nodes one step inside a Neumann face; the kernel visits those layers
    (a node on two layers once) instead of the whole mesh (option rhs_full): same bits."""
    from pyapes_amd.hip import lib as L
    from pyapes_amd.hip.context import context_for
    rng = random.Random(99)
    seen = 0
    for case in range(60):
        nd = rng.choice([1, 2, 3])
        n = [rng.choice([3, 4, 5, 8, 17, 33]) for _ in range(nd)]
        bcs = []
        for a in range(nd):
            for _ in range(2):
                t = rng.choice(["dirichlet", "neumann", "neumann", "symmetry"])
                bcs.append((t, None if t == "symmetry" else round(rng.uniform(-1, 1), 3)))
        if not any(t == "neumann" for t, _ in bcs):
            continue
        dtype = rng.choice(["double", "single"])
        tdt = torch.float64 if dtype == "double" else torch.float32
        mesh = Mesh(Box([0.0] * nd, [1.0 + 0.1 * a for a in range(nd)]), None, list(n), "cuda", dtype)
        cfg = [{"bc_face": _faces()[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
        var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        terms = [{"kind": L.OP_LAPLACIAN, "sign": -1.0, "coeff": 0.7}]
        if nd == 1:
            terms.append({"kind": L.OP_GRAD, "sign": 1.0, "coeff": 0.3})
        ctx.set_terms(terms)
        g = torch.Generator().manual_seed(case)
        rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt).cuda()
        a, b = rhs.clone(), rhs.clone()
        ctx.set_option("rhs_full", 0)
        ctx.rhs_adjust(a[0])
        ctx.set_option("rhs_full", 1)
        ctx.rhs_adjust(b[0])
        ctx.set_option("rhs_full", 0)
        assert torch.equal(a, b), (case, n, bcs, dtype)
        assert not torch.equal(a, rhs) or all(v == 0 for t, v in bcs if t == "neumann")
        seen += 1
    assert seen >= 30
