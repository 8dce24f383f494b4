"""-m gpu: BiCGSTAB on the tiled kernels (p/v phase and the fused s/t phase) vs the generic kernels
and the oracle, 3-D and 2-D, for short fixed iteration counts and to convergence."""
import warnings

import pytest
import torch

import pyapes_oracle as O
from helpers import hip_options, rel_err

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

D = lambda v=0.0: ("dirichlet", v)   # noqa: E731
N = lambda v=0.0: ("neumann", v)     # noqa: E731
SY = ("symmetry", None)
CASES = [
    ((12, 18, 20), "double", [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)]),
    ((9, 16, 132), "double", [N(0.3), D(0.0), SY, SY, SY, D(2.0)]),
    ((40, 44), "double", [N(0.0), D(0.0), N(0.0), D(1.0)]),
    ((16, 20, 24), "single", [D(0.0)] * 6),
    ((14, 20, 24), "double", [D(0.0), D(0.5), D(0.0), D(0.0), D(1.0), D(0.0)]),
    # odd row lengths: NARROW kernels
    ((12, 18, 21), "double", [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)]),
    ((9, 16, 131), "double", [N(0.3), D(0.0), SY, SY, SY, D(2.0)]),
    ((40, 45), "double", [N(0.0), D(0.0), N(0.0), D(1.0)]),
    ((16, 20, 27), "double", [D(0.0)] * 6),
]


def _run(n, dtype, bcs, rhs0, K, tol, fast, monkeypatch):
    monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, list(n), "cuda", dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    solver = Solver({"fdm": {"method": "bicgstab", "tol": tol, "max_it": K, "report": False}})
    solver.set_eq(FDM().laplacian(0.9, var) == rhs0.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = solver.solve()
    return var().cpu(), rep


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + c[1][0])
def test_tiled_bicgstab(case, monkeypatch):
    n, dtype, bcs = case
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(13)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    nd = len(n)
    om = O.OMesh([0.0] * nd, [1.0] * nd, list(n), dtype)
    orc = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(bcs)]
    tol_x = 1e-10 if dtype == "double" else 1e-5
    for K in (1, 4, 9):
        xf, rf = _run(n, dtype, bcs, rhs0, K, 1e-30, True, monkeypatch)
        xg, rg = _run(n, dtype, bcs, rhs0, K, 1e-30, False, monkeypatch)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xo, ro = O.solve_poisson(om, orc, rhs0.clone(), method="bicgstab", tol=1e-30, max_it=K, coeff=0.9)
        assert rf["itr"] == rg["itr"] == ro["itr"] == K
        assert rel_err(xf, xg) <= tol_x
        assert rel_err(xf, xo) <= tol_x
        assert abs(rf["tol"] - ro["tol"]) <= (1e-8 if dtype == "double" else 1e-3) * abs(ro["tol"])
    if dtype == "double" and all(t == "dirichlet" for t, _ in bcs):
        # to convergence on a well-conditioned (all-Dirichlet) problem: the stop rule, incl. the early
        # exit on |s|, must end both paths alike.  (With Neumann/symmetry faces BiCGSTAB stagnates
        # erratically at this tolerance in the oracle itself: 958 / 1184 / >2000 iterations over
        # summation orders on the 40x44 case, so long runs are not comparable there.)
        xf, rf = _run(n, dtype, bcs, rhs0, 2000, 1e-9, True, monkeypatch)
        xg, rg = _run(n, dtype, bcs, rhs0, 2000, 1e-9, False, monkeypatch)
        assert rf["converge"] and rg["converge"], (rf, rg)
        assert abs(rf["itr"] - rg["itr"]) <= max(3, rg["itr"] // 5), (rf, rg)
        assert rel_err(xf, xg) <= 1e-7


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + c[1][0])
def test_next_direction_formed_by_the_x_update_changes_no_bit(case, monkeypatch):
    """Round 3: from the second iteration on the p / v phase is v' = A p' alone (k_cg3d phase 8) -- p' was formed by the
    previous iteration's x / r update (k_bicg_x: beta is complete as soon as omega and rho_next are, linalg.py:212,
    246-247).  The same operations on the same operands, the same partial sums: every bit of the iterate, the
    iteration count and the stop-test value must equal the sequence with the full p / v phase in every iteration."""
    n, dtype, bcs = case
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(17)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    for K, tol in ((1, 1e-30), (2, 1e-30), (3, 1e-30), (12, 1e-30), (400, 1e-7 if dtype == "double" else 1e-3)):
        hip_options(monkeypatch, bicg_pfold=1)
        xa, ra = _run(n, dtype, bcs, rhs0, K, tol, True, monkeypatch)
        hip_options(monkeypatch, bicg_pfold=0)
        xb, rb = _run(n, dtype, bcs, rhs0, K, tol, True, monkeypatch)
        assert ra["itr"] == rb["itr"] and ra["tol"] == rb["tol"], (K, ra, rb)
        assert torch.equal(xa, xb), (K, float((xa - xb).abs().max()))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + c[1][0])
def test_s_reformed_by_the_x_update_changes_no_bit(case, monkeypatch):
    """Round 4: the tiled s / t phase stores t alone; the x / r update re-forms s = r - alpha v' from r and v' (which it reads
    anyway) with the combine of phase 6, operation for operation, and updates r in place -- 15 array passes per iteration for
    16.  Every bit of the iterate, the iteration count and the stop-test value must equal the sequence that stores s
    (option bicg_srv 0), with and without the folded next direction, early exit on |s| included (the long run)."""
    n, dtype, bcs = case
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(23)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    for pfold in (1, 0):
        for K, tol in ((1, 1e-30), (2, 1e-30), (7, 1e-30), (400, 1e-7 if dtype == "double" else 1e-3)):
            hip_options(monkeypatch, bicg_pfold=pfold, bicg_srv=1)
            xa, ra = _run(n, dtype, bcs, rhs0, K, tol, True, monkeypatch)
            hip_options(monkeypatch, bicg_pfold=pfold, bicg_srv=0)
            xb, rb = _run(n, dtype, bcs, rhs0, K, tol, True, monkeypatch)
            assert ra["itr"] == rb["itr"] and ra["tol"] == rb["tol"], (pfold, K, ra, rb)
            assert torch.equal(xa, xb), (pfold, K, float((xa - xb).abs().max()))
    hip_options(monkeypatch, bicg_pfold=None, bicg_srv=None)
