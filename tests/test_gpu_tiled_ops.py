"""-m gpu: the tiled single-field kernels (A x, explicit Euler, Jacobi: phases 2-4 of k_cg3d) against
the generic kernels -- bit-exact, since both evaluate the same literal arithmetic -- on shapes with
partial tiles, all BC types, both dtypes; and Euler / Jacobi against the oracle at 3-D sizes the
fast path covers."""
import os
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
BCS = {
    "mix": [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)],
    "sym": [N(0.3), N(0.0), SY, SY, SY, D(2.0)],
    "per": [PE] * 6,
    "xper": [PE, PE, D(0.0), D(1.0), N(0.0), N(0.2)],
    "dirper": [D(0.0), D(0.2), PE, PE, D(0.1), D(0.0)],
}
SHAPES = [((20, 37, 50), "double"), ((9, 16, 128), "double"), ((17, 70, 260), "double"), ((12, 18, 132), "single"),
          ((20, 37, 51), "double"), ((9, 17, 129), "double"), ((12, 18, 133), "single"), ((11, 13, 130), "single")]


def _cfg(bcs):
    return [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]


def _field(n, dtype, bcs, x0, fast, monkeypatch):
    monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
    mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype)
    var = Field("p", 1, mesh, {"domain": _cfg(bcs), "obstacle": None})
    var.set_var_tensor(x0.cuda().clone())
    var.apply_bcs()
    return mesh, var


@pytest.mark.parametrize("bc", list(BCS), ids=list(BCS))
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[0])) + s[1][0])
def test_tiled_equals_generic_bitwise(shape, bc, monkeypatch):
    n, dtype = shape
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    ut = (0.5 + torch.randn((1, *n), generator=g, dtype=torch.float64)).to(tdt)
    out = {}
    for fast in (True, False):
        mesh, var = _field(n, dtype, BCS[bc], x0, fast, monkeypatch)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = {"lap": FDC({"laplacian": {"edge": False}}).laplacian(var).cpu()}
            res["grad"] = FDC({"grad": {"edge": False}}).grad(var).cpu()
            solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 1, "report": False}})
            solver.set_eq(-FDM().laplacian(0.7, var) == torch.zeros_like(var()))
            res["aop"] = solver.Aop(var).cpu()
            lims = ["upwind", "compat"] + (["none"] if bc in ("per", "dirper") else [])
            for lim in lims:
                # (scalar speeds of both signs and zero: k_sf takes the sign of a scalar upwind speed as a launch-time fact and
                # skips the half of every axis term that is a signed zero -- the generic kernels form both halves)
                for uname, u in (("s", 1.3), ("sn", -0.9), ("s0", 0.0), ("t", ut.cuda())):
                    if lim == "none" and uname == "t":
                        continue
                    if uname in ("sn", "s0") and lim != "upwind":
                        continue
                    v2 = var.copy()
                    cfg = {"div": {"limiter": "upwind" if lim == "compat" else lim, "compat": lim == "compat"}}
                    for _ in range(3):
                        euler_step(v2, u, 1e-3, 2e-3, cfg)
                    res[f"euler_{lim}_{uname}"] = v2().cpu()
                    # the explicit operator itself (the Div term alone in the tiled A x kernel)
                    dcfg = {"div": dict(cfg["div"], edge=False)}
                    res[f"div_{lim}_{uname}"] = FDC(dcfg).div(u, var).cpu()
        out[fast] = res
    for k in out[True]:
        assert torch.equal(out[True][k], out[False][k]), (k, float((out[True][k] - out[False][k]).abs().max()))


@pytest.mark.parametrize("bc", ["mix", "sym"], ids=["mix", "sym"])
def test_tiled_jacobi_and_euler_vs_oracle(bc, monkeypatch):
    n, dtype = (14, 20, 36), "double"
    g = torch.Generator().manual_seed(9)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype)
    om = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), dtype)
    orc = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(BCS[bc])]
    # Jacobi, 30 sweeps
    var = Field("p", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
    solver = Solver({"fdm": {"method": "jacobi", "tol": 1e-30, "max_it": 29, "report": False, "omega": 0.9}})
    solver.set_eq(FDM().laplacian(0.8, var) == rhs0.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = solver.solve()
        xo, ro = O.solve_poisson(om, orc, rhs0.clone(), method="jacobi", tol=1e-30, max_it=29, coeff=0.8, omega=0.9)
    assert rep["itr"] == ro["itr"] == 30
    assert rel_err(var().cpu(), xo) < 1e-13
    assert abs(rep["tol"] - ro["tol"]) <= 1e-10 * abs(ro["tol"])
    # Euler, upwind, 5 steps
    phi0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    v2 = Field("phi", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
    v2.set_var_tensor(phi0.cuda().clone())
    v2.apply_bcs()
    po = phi0.clone()
    bo = O.make_bcs(om, orc)
    O.bc_fill(po, bo)
    for _ in range(5):
        euler_step(v2, 0.8, 1e-3, 1e-3, {"div": {"limiter": "upwind"}})
        po = O.euler_step(po, 0.8, 1e-3, 1e-3, om, bo, "upwind")
    assert rel_err(v2().cpu(), po) < 1e-13


def test_euler_march_equals_repeated_steps():
    from pyapes_amd.solver.march import euler_march
    n = (14, 20, 36)
    g = torch.Generator().manual_seed(4)
    phi0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", "double")
    for nsteps in (1, 4, 7):
        a = Field("a", 1, mesh, {"domain": _cfg(BCS["sym"]), "obstacle": None})
        b = Field("b", 1, mesh, {"domain": _cfg(BCS["sym"]), "obstacle": None})
        for f in (a, b):
            f.set_var_tensor(phi0.cuda().clone())
            f.apply_bcs()
        for _ in range(nsteps):
            euler_step(a, 0.8, 1e-3, 1e-3, {"div": {"limiter": "upwind"}})
        euler_march(b, 0.8, 1e-3, 1e-3, nsteps, {"div": {"limiter": "upwind"}})
        assert torch.equal(a(), b()), nsteps


@pytest.mark.parametrize("shape", [((12, 18, 20), "double"), ((40, 44), "double"), ((10, 12, 136), "single")],
                         ids=["3d", "2d", "3d_f32"])
def test_tensor_coefficient_laplacian_tiled(shape, monkeypatch):
    """laplacian(Gamma(x), phi): A x, CG and Jacobi with a tensor coefficient on the
    tiled kernels vs the generic kernels (bit-exact A x) and the oracle."""
    n, dtype = shape
    nd = len(n)
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    gamma = (1.0 + 0.2 * torch.rand((1, *n), generator=g, dtype=torch.float64)).to(tdt)
    bcs = [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)][:2 * nd]
    out = {}
    for fast in (True, False):
        monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
        mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": _cfg(bcs), "obstacle": None})
        var.set_var_tensor(x0.cuda().clone())
        var.apply_bcs()
        solver = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": 5, "report": False}})
        solver.set_eq(FDM().laplacian(gamma.cuda(), var) == rhs0.cuda().clone())
        res = {"aop": solver.Aop(var).cpu()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
        res["cg"], res["rep"] = var().cpu(), rep
        out[fast] = res
    assert torch.equal(out[True]["aop"], out[False]["aop"])
    assert out[True]["rep"]["itr"] == out[False]["rep"]["itr"] == 6
    assert rel_err(out[True]["cg"], out[False]["cg"]) <= (1e-12 if dtype == "double" else 2e-5)
    om = O.OMesh([0.0] * nd, [1.0] * nd, list(n), dtype)
    orc = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(bcs)]
    bo = O.make_bcs(om, orc)
    xo = x0.clone()
    tabs = O.laplacian_tables(xo, om, bo)
    rhs = rhs0.clone() + O.laplacian_rhs_adjust(xo, om, bo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xs, ro = O.cg(xo, rhs, [O.OTerm("laplacian", tabs, gamma, 1.0)], om, bo, 1e-30, 5)
    assert ro["itr"] == 6
    assert rel_err(out[True]["cg"], xs) <= (1e-10 if dtype == "double" else 1e-5)


@pytest.mark.parametrize("bc", ["mix", "per", "xper"], ids=["mix", "per", "xper"])
@pytest.mark.parametrize("shape", [((20, 37, 50), "double"), ((17, 70, 260), "double"), ((12, 18, 133), "single")],
                         ids=lambda s: "x".join(map(str, s[0])) + s[1][0])
def test_jacobi_sweeps_marching_in_alternating_directions(shape, bc, monkeypatch):
    """Round 4: consecutive Jacobi sweeps of a 3-D mesh march their chunks in opposite directions (k_cg3d PHASE 9: what a
    sweep wrote last is what the next one reads first).  No global sum feeds back into a Jacobi iterate, so for a fixed
    number of sweeps -- odd and even: the last sweep forwards / backwards -- every bit of the iterate must equal the
    all-forwards sequence (option jac_alt 0) and the generic kernels; the stop-test value differs only by the order its
    partial sums are added in."""
    from helpers import hip_options
    n, dtype = shape
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(31)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)

    def run(K, alt, fast=True):
        hip_options(monkeypatch, jac_alt=alt)
        monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
        mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": _cfg(BCS[bc]), "obstacle": None})
        solver = Solver({"fdm": {"method": "jacobi", "tol": 1e-30, "max_it": K - 1, "report": False, "omega": 0.9}})
        solver.set_eq(FDM().laplacian(0.8, var) == rhs0.cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
        return var().cpu(), rep

    for K in (1, 2, 7, 12):
        xa, ra = run(K, 1)
        xb, rb = run(K, 0)
        assert ra["itr"] == rb["itr"] == K
        assert torch.equal(xa, xb), (K, float((xa - xb).abs().max()))
        assert abs(ra["tol"] - rb["tol"]) <= (1e-12 if dtype == "double" else 1e-5) * abs(rb["tol"])
    xg, rg = run(7, 1, fast=False)
    xa, ra = run(7, 1)
    assert torch.equal(xa, xg)
    hip_options(monkeypatch, jac_alt=None)


@pytest.mark.parametrize("shape", [((20, 37, 50), "double"), ((12, 18, 133), "single")],
                         ids=lambda s: "x".join(map(str, s[0])) + s[1][0])
def test_jacobi_with_a_tensor_coefficient_alternating_sweeps(shape, monkeypatch):
    """The Jacobi sweep with laplacian(Gamma(x), phi) on the tiled kernels (k_cg3d<..., CF>, phases 4 and 9: sweeps in
    alternating directions): every bit of the iterate equal to the generic kernels and to the all-forwards sequence."""
    from helpers import hip_options
    n, dtype = shape
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(41)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    gamma = (1.0 + 0.2 * torch.rand((1, *n), generator=g, dtype=torch.float64)).to(tdt)

    def run(K, alt, fast):
        hip_options(monkeypatch, jac_alt=alt)
        monkeypatch.setenv("PYAPES_HIP_FASTPATH", "1" if fast else "0")
        mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": _cfg(BCS["mix"]), "obstacle": None})
        solver = Solver({"fdm": {"method": "jacobi", "tol": 1e-30, "max_it": K - 1, "report": False, "omega": 0.8}})
        solver.set_eq(FDM().laplacian(gamma.cuda(), var) == rhs0.cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
        return var().cpu(), rep

    for K in (2, 5):
        xa, ra = run(K, 1, True)
        xb, _ = run(K, 0, True)
        xg, rg = run(K, 1, False)
        assert ra["itr"] == rg["itr"] == K
        assert torch.equal(xa, xb) and torch.equal(xa, xg), (K, float((xa - xg).abs().max()))
    hip_options(monkeypatch, jac_alt=None)
