"""-m gpu: the entry points the golden vectors do not reach -- Jacobi, explicit Euler, intended
upwind, tensor coefficients, callable / per-node BC values, error behaviour -- against the oracle."""
import warnings
from math import pi

import pytest
import torch

import pyapes_oracle as O
from helpers import bit_equal, rel_err

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdc import FDC
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.march import euler_step
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_bcs, poisson_exact_nd, poisson_rhs_nd
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs


def _cfgs(bcs):
    prod = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    orc = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(bcs)]
    return prod, orc


@pytest.mark.parametrize("nd,n", [(1, [33]), (2, [17, 20]), (3, [12, 10, 14])])
def test_jacobi_matches_oracle(nd, n):
    bcs = [("dirichlet", 0.2), ("neumann", 0.1), ("dirichlet", 0.0), ("symmetry", None), ("dirichlet", 1.0),
           ("neumann", 0.0)][:2 * nd]
    prod, orc = _cfgs(bcs)
    lo, up = [0.0] * nd, [1.0] * nd
    g = torch.Generator().manual_seed(3)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    for K, omega in ((0, 1.0), (25, 0.8)):
        mesh = Mesh(Box(lo, up), None, n, "cuda", "double")
        var = Field("p", 1, mesh, {"domain": prod, "obstacle": None})
        rhs = rhs0.cuda().clone()
        solver = Solver({"fdm": {"method": "jacobi", "tol": 1e-30, "max_it": K, "report": False, "omega": omega}})
        solver.set_eq(FDM().laplacian(0.9, var) == rhs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
            om = O.OMesh(lo, up, n, "double")
            xo, ro = O.solve_poisson(om, orc, rhs0.clone(), method="jacobi", tol=1e-30, max_it=K, coeff=0.9,
                                     omega=omega)
        assert rep["itr"] == ro["itr"] == K + 1
        assert rel_err(var().cpu(), xo) < 1e-13
        assert abs(rep["tol"] - ro["tol"]) <= 1e-10 * abs(ro["tol"])


def test_jacobi_converges_to_cg_solution_config1():
    """BASELINE config 1: 2-D Poisson 128x128 fp64, Dirichlet, Jacobi -- no reference Jacobi exists
    (SURVEY Q1); the converged iterate must equal the reference CG solution to solver tolerance."""
    mesh = Mesh(Box[0:1, 0:1], None, [128, 128], "cuda", "double")
    out = {}
    for method, tol, mx in (("cg", 1e-10, 2000), ("jacobi", 1e-9, 100000)):
        var = Field("p", 1, mesh, {"domain": poisson_bcs(2), "obstacle": None})
        rhs = poisson_rhs_nd(mesh, var)
        solver = Solver({"fdm": {"method": method, "tol": tol, "max_it": mx, "report": False}})
        solver.set_eq(FDM().laplacian(1.0, var) == rhs)
        rep = solver.solve()
        assert rep["converge"], (method, rep)
        out[method] = var().clone()
    assert float((out["cg"] - out["jacobi"]).abs().max()) < 5e-5
    assert float((out["cg"][0] - poisson_exact_nd(mesh)).abs().max()) < 1e-3


def test_callable_bcs_128_known_answer():
    """BASELINE config 1 inputs with the reference solver: CG 271 iterations."""
    mesh = Mesh(Box[0:1, 0:1], None, [128, 128], "cuda", "double")
    var = Field("p", 1, mesh, {"domain": poisson_bcs(2), "obstacle": None})
    rhs = poisson_rhs_nd(mesh, var)
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 1000, "report": False}})
    solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    rep = solver.solve()
    assert rep["itr"] == 271 and abs(rep["tol"] - 9.195241388276045e-07) < 1e-15


@pytest.mark.parametrize("limiter", ["upwind", "none"])
@pytest.mark.parametrize("dtype", ["double", "single"])
def test_euler_step_matches_oracle(limiter, dtype):
    """BASELINE config 4 family at small size: explicit adv-diff march, Neumann/Symmetry BCs (upwind)
    or Dirichlet/periodic (central: the reference's central Div cannot take neumann faces)."""
    n = [14, 12, 16]
    if limiter == "upwind":
        bcs = [("neumann", 0.0), ("neumann", 0.0), ("symmetry", None), ("symmetry", None), ("symmetry", None),
               ("symmetry", None)]
    else:
        bcs = [("dirichlet", 0.0), ("dirichlet", 0.0), ("periodic", None), ("periodic", None), ("dirichlet", 0.1),
               ("dirichlet", 0.0)]
    prod, orc = _cfgs(bcs)
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, n, "cuda", dtype)
    om = O.OMesh([0, 0, 0], [1, 1, 1], n, dtype)
    gx = om.grid
    phi0 = torch.exp(-((gx[0] - 0.5) ** 2 + (gx[1] - 0.5) ** 2 + (gx[2] - 0.5) ** 2) / 0.02).unsqueeze(0)
    g = torch.Generator().manual_seed(5)
    ut = (1.0 + 0.3 * torch.randn((1, *n), generator=g, dtype=torch.float64)).to(om.dtype)
    nu, dt = 1e-3, 2e-3
    for u in (1.0, ut):
        var = Field("phi", 1, mesh, {"domain": prod, "obstacle": None})
        var.set_var_tensor(phi0.cuda().clone())
        var.apply_bcs()
        po = phi0.clone()
        bo = O.make_bcs(om, orc)
        O.bc_fill(po, bo)
        for _ in range(5):
            euler_step(var, u if isinstance(u, float) else u.cuda(), nu, dt, {"div": {"limiter": limiter}})
            po = O.euler_step(po, u, nu, dt, om, bo, limiter)
        assert rel_err(var().cpu(), po) < (1e-13 if dtype == "double" else 1e-5)


def test_upwind_intended_and_compat():
    mesh = Mesh(Box[0:1], None, [11], "cuda", "double")
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(1, 0.0, "dirichlet"), "obstacle": None})
    var.set_var_tensor((mesh.X ** 2).unsqueeze(0).clone())
    out = FDC({"div": {"limiter": "upwind", "edge": False}}).div(2.0, var)
    phi = var()[0]
    expect = 2.0 * (phi[1:-1] - phi[:-2]) / mesh.dx[0]      # reference tests/test_fdm.py:239
    assert torch.allclose(out[0][1:-1], expect, rtol=0, atol=1e-13)
    om = O.OMesh([0], [1], [11], "double")
    assert bit_equal(out, O.div_upwind_intended(2.0, var().cpu(), om))


def test_central_div_with_neumann_raises_like_reference():
    mesh = Mesh(Box[0:1, 0:1], None, [8, 8], "cuda", "double")
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(2, 0.0, "neumann"), "obstacle": None})
    from pyapes_amd.hip.lib import PaError
    with pytest.raises(PaError, match="IndexError"):
        FDC({"div": {"limiter": "none", "edge": False}}).div(1.0, var)


def test_tensor_coefficient_laplacian_and_per_node_bc_values():
    n = [9, 11, 10]
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, n, "cuda", "double")
    om = O.OMesh([0, 0, 0], [1, 1, 1], n, "double")
    g = torch.Generator().manual_seed(11)
    face_vals = torch.randn(n[1] * n[2], generator=g, dtype=torch.float64)   # xl: gather order = C order
    grad_vals = torch.randn(n[0] * n[1], generator=g, dtype=torch.float64)   # zu
    # the reference accepts a Tensor for the fill (bcs.py:210-211, 246-247) but only callables /
    # numbers / lists in the rhs adjustment (fdc.py:803-817): the per-node gradient is a callable
    def grad_fn(grid, mask, *_):
        return grad_vals.to(mask.device)

    bcs = [("dirichlet", face_vals), ("dirichlet", 0.0), ("dirichlet", 0.0), ("dirichlet", 0.0),
           ("dirichlet", 0.0), ("neumann", grad_fn)]
    prod, orc = _cfgs(bcs)
    prod[0]["bc_val"] = face_vals.cuda()
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    gamma = 1.0 + 0.1 * torch.randn((1, *n), generator=g, dtype=torch.float64)
    var = Field("p", 1, mesh, {"domain": prod, "obstacle": None})
    var.set_var_tensor(x0.cuda().clone())
    var.apply_bcs()
    bo = O.make_bcs(om, orc)
    xo = x0.clone()
    O.bc_fill(xo, bo)
    assert bit_equal(var(), xo)
    rhs = torch.zeros_like(var())
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 3, "report": False}})
    solver.set_eq(FDM().laplacian(gamma.cuda(), var) == rhs)
    tabs = O.laplacian_tables(xo, om, bo)
    assert bit_equal(solver.Aop(var), O.Aop(xo, [O.OTerm("laplacian", tabs, gamma, 1.0)], 3))
    assert bit_equal(rhs, O.laplacian_rhs_adjust(xo, om, bo))


def test_unknown_method_and_vector_field_errors():
    mesh = Mesh(Box[0:1], None, [11], "cuda", "double")
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(1, 0.0, "dirichlet"), "obstacle": None})
    solver = Solver({"fdm": {"method": "gmres", "tol": 1e-6, "max_it": 3, "report": False}})
    solver.set_eq(FDM().laplacian(1.0, var) == 1.0)
    with pytest.raises(RuntimeError, match="only supports"):
        solver.solve()


def test_nonfinite_tolerance_raises_runtime_error():
    mesh = Mesh(Box[0:1, 0:1], None, [9, 9], "cuda", "double")
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(2, 0.0, "neumann"), "obstacle": None})
    rhs = torch.full((1, 9, 9), float("inf"), dtype=torch.float64, device="cuda")
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 5, "report": False}})
    solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    # reference: r = inf, A d = nan, alpha = nan_to_num(nan) = 0, x = x + 0 * inf = nan -> tol nan ->
    # RuntimeError("Invalid tolerance detected!") (linalg.py:334-336)
    with pytest.raises(RuntimeError, match="Invalid tolerance"):
        solver.solve()


@pytest.mark.parametrize("nd", [1, 2, 3])
def test_box_field_bcs_like_reference(nd):
    """What reference tests/test_variables.py::test_box_field_bcs asserts (its Neumann part with the
    formula the reference code actually implements, +2/3 V dx on both sides, bcs.py:251-253)."""
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, [0.1] * nd, "cuda", "double")

    def filled(val, typ):
        var = Field(typ[0], 1, mesh, {"domain": homogeneous_bcs(nd, val, typ), "obstacle": None}, init_val="random")
        for bc in var.bcs:
            bc.apply(var(), mesh.grid, 0)           # face by face through BC.apply, like the reference test
        return var

    v = filled(0.44, "dirichlet")()[0]
    assert abs(float(v[0].mean()) - 0.44) < 1e-15 and abs(float(v[-1].mean()) - 0.44) < 1e-15
    var = filled(1.0, "neumann")
    v = var()[0]
    inner = (slice(1, -1),) * (nd - 1)               # off the edges the later faces overwrite
    dx0 = mesh.dx_list[0]
    assert torch.allclose(v[0][inner], (4 / 3 * v[1] - 1 / 3 * v[2] + 2 / 3 * 1.0 * dx0)[inner], rtol=0, atol=1e-14)
    assert torch.allclose(v[-1][inner], (4 / 3 * v[-2] - 1 / 3 * v[-3] + 2 / 3 * 1.0 * dx0)[inner], rtol=0, atol=1e-14)
    v = filled(None, "periodic")()[0]
    assert torch.equal(v[0], v[-1])
    var = filled(None, "symmetry")
    v = var()[0]
    assert var.get_bc("d-xl").type == "symmetry" and var.get_bc("d-xl").bc_id == "d-xl" and var.get_bc("nope") is None
    assert torch.equal(v[0][inner], v[1][inner]) and torch.equal(v[-1][inner], v[-2][inner])


def test_callable_bc_with_bc_val_opt():
    """reference tests/test_variables.py::test_cylinder_field_bcs on a Box: callable values see
    (grid, mask, var, bc_val_opt) and return values in boolean-mask gather order."""
    from pyapes_amd.variables.bcs import BoxBoundary
    mesh = Mesh(Box[0:1, 0:2], None, [5, 5], "cuda", "double")

    def xu_bc(grid, mask, *_):
        return grid[1][mask] * 4.4

    def yu_bc(grid, mask, _, opt):
        return grid[0][mask] * torch.sum(opt["T"])

    f_bc = BoxBoundary(xl={"bc_type": "neumann", "bc_val": 0}, xu={"bc_type": "dirichlet", "bc_val": xu_bc},
                       yl={"bc_type": "neumann", "bc_val": 1.3},
                       yu={"bc_type": "dirichlet", "bc_val": yu_bc, "bc_val_opt": {"T": torch.ones(5, 5)}})
    var = Field("d", 1, mesh, {"domain": f_bc(), "obstacle": None}, init_val="random")
    var.apply_bcs()
    v = var()[0]
    assert torch.allclose(v[-1, 1:-1], 4.4 * mesh.grid[1][0][1:-1])
    assert torch.allclose(v[1:-1, -1], mesh.grid[0][1:-1, -1] * 25.0)
    assert torch.allclose(v[0, 1:-1], 4 / 3 * v[1, 1:-1] - 1 / 3 * v[2, 1:-1])
    assert torch.allclose(v[1:-1, 0], 4 / 3 * v[1:-1, 1] - 1 / 3 * v[1:-1, 2] + 2 / 3 * 1.3 * mesh.dx[1])


# ---- Field.VARo and BC callables that read the field (round 2: the two API deviations of round 1) ----------
def _solve_k(method, n, bcs, rhs0, K, save_old):
    prod, _ = _cfgs(bcs)
    mesh = Mesh(Box([0.0] * len(n), [1.0] * len(n)), None, list(n), "cuda", "double")
    var = Field("p", 1, mesh, {"domain": prod, "obstacle": None})
    s = Solver({"fdm": {"method": method, "tol": 1e-30, "max_it": K, "report": False, "save_old": save_old}})
    s.set_eq(FDM().laplacian(1.0, var) == rhs0.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var, rep


@pytest.mark.parametrize("n", [(12, 10, 132), (9, 11, 13), (33, 40)], ids=["tiled", "generic", "2d"])
@pytest.mark.parametrize("method", ["cg", "bicgstab", "jacobi"])
def test_varo_is_the_iterate_before_the_last_iteration(method, n):
    """var.save_old() opens every iteration of the reference's loops (linalg.py:110, 210), so after
    solve() Field.VARo holds the iterate before the last executed iteration.  With {"save_old": True} the
    device loops keep it: it must be, bit for bit, what a solve that stops one iteration earlier returns
    (and that one is pinned to the oracle elsewhere)."""
    bcs = [("dirichlet", 0.2), ("neumann", 0.1), ("dirichlet", 0.0), ("symmetry", None), ("dirichlet", 1.0),
           ("neumann", 0.0)][:2 * len(n)]
    g = torch.Generator().manual_seed(5)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    K = 7
    var, rep = _solve_k(method, n, bcs, rhs0, K, True)
    prev, rep_prev = _solve_k(method, n, bcs, rhs0, K - 1, False)
    assert rep["itr"] == rep_prev["itr"] + 1
    assert bit_equal(var.VARo, prev()), method
    assert not bit_equal(var.VARo, var())
    # the solve without the flag gives the same answer and a VARo that refuses to be read
    plain, rep_plain = _solve_k(method, n, bcs, rhs0, K, False)
    assert bit_equal(plain(), var()) and rep_plain == rep
    with pytest.raises(RuntimeError, match="VARo"):
        plain.VARo
    plain.save_old()                      # the user's own save_old() makes it readable again
    assert bit_equal(plain.VARo, plain())


def test_bc_callable_that_reads_the_field_takes_the_host_stepped_loop():
    """The reference calls a callable bc_val inside every BC fill with the current iterate (bcs.py:203,
    245); the device solvers evaluate it once per solve.  A callable of (grid, mask) alone -- all the
    reference's own tests use -- is the same thing either way; one that reads ``var`` is detected and the
    solve goes through solver/host_stepped.py, which returns to Python for every face of every fill like the
    reference (results against the REFERENCE: the *_robin_* goldens of test_gpu_parity_golden.py; here the
    literal oracle on another mesh, CG and BiCGSTAB).  Jacobi has no reference behaviour and stays refused."""
    n = [12, 14]
    mesh = Mesh(Box[0:1, 0:1], None, n, "cuda", "double")
    robin = lambda grid, mask, var, opt: 0.5 * var[0][torch.roll(mask, 1, 0)]   # noqa: E731  reads the field
    faces = ("xl", "xu", "yl", "yu")
    bcs = [{"bc_face": f, "bc_type": "dirichlet", "bc_val": (robin if f == "xl" else 0.0), "bc_val_opt": None} for f in faces]
    orc = [{"bc_face": f, "bc_type": "dirichlet", "bc_val": (robin if f == "xl" else 0.0)} for f in faces]
    om = O.OMesh([0, 0], [1, 1], n, "double")
    g = torch.Generator().manual_seed(3)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    for method, K in (("cg", 9), ("bicgstab", 7)):
        var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
        assert var.bcs[0].depends_on_var(var()) and not var.bcs[1].depends_on_var(var())
        s = Solver({"fdm": {"method": method, "tol": 1e-30, "max_it": K, "report": False}})
        s.set_eq(FDM().laplacian(1.0, var) == rhs0.cuda().clone())
        with pytest.warns(RuntimeWarning):
            rep = s.solve()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xo, ro = O.solve_poisson(om, orc, rhs0.clone(), method=method, tol=1e-30, max_it=K)
        assert rep["itr"] == ro["itr"]
        assert rel_err(var().cpu(), xo) < 1e-10, (method, rel_err(var().cpu(), xo))
        assert abs(rep["tol"] - ro["tol"]) <= 1e-9 * abs(ro["tol"])
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    s = Solver({"fdm": {"method": "jacobi", "tol": 1e-8, "max_it": 10, "report": False}})
    s.set_eq(FDM().laplacian(1.0, var) == torch.ones_like(var()))
    with pytest.raises(NotImplementedError, match="Jacobi"):
        s.solve()
    # the explicit BC fill evaluates it with the current field (oracle: literal sequential fill)
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None}, init_val="random")
    x0 = var().clone()
    var.apply_bcs()
    xo = x0.cpu().clone()
    O.bc_fill(xo, O.make_bcs(om, orc))
    assert bit_equal(var(), xo)
    # a callable of (grid, mask) only runs on the device loops
    ok = Field("q", 1, mesh, {"domain": poisson_bcs(2), "obstacle": None})
    s2 = Solver({"fdm": {"method": "cg", "tol": 1e-8, "max_it": 50, "report": False}})
    s2.set_eq(FDM().laplacian(1.0, ok) == poisson_rhs_nd(mesh, ok))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert s2.solve()["itr"] > 1


def test_dirichlet_and_neumann_need_a_value_like_the_reference():
    mesh = Mesh(Box[0:1, 0:1], None, [8, 8], "cuda", "double")
    bcs = [{"bc_face": f, "bc_type": "dirichlet", "bc_val": (None if f == "yu" else 0.0), "bc_val_opt": None}
           for f in ("xl", "xu", "yl", "yu")]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    with pytest.raises(AssertionError, match="bc_val is not specified"):   # bcs.py:200
        var.apply_bcs()


def test_switching_streams_orders_the_new_stream_after_the_old():
    """One ctx per mesh; its scratch (BC shell buffers, partial rows, solver scalars) is shared by everything that
    runs on the mesh.  A caller that enqueues a long asynchronous march on one torch stream and then solves on
    ANOTHER (``with torch.cuda.stream(s)``) moves the ctx to that stream: pa_ctx_set_stream must order the new
    stream after what the old one still holds, or the two race on the scratch.  Results must equal the serial ones."""
    from pyapes_amd.solver.march import euler_march
    n = [96, 64, 72]
    nsym = mixed_bcs([0.0, 0.0, None, None, None, None], ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
    mixd = mixed_bcs([0.0, 0.5, 0.0, 0.0, 1.0, -0.25], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])
    g = torch.Generator().manual_seed(11)
    phi0 = torch.rand((1, *n), generator=g, dtype=torch.float64)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)

    def run(two_streams):
        mesh = Mesh(Box[0:1, 0:1, 0:1], None, n, "cuda", "double")
        phi = Field("phi", 1, mesh, {"domain": nsym, "obstacle": None})
        phi.set_var_tensor(phi0.cuda().clone())
        phi.apply_bcs()
        var = Field("p", 1, mesh, {"domain": mixd, "obstacle": None})
        rhs = rhs0.cuda().clone()
        solver = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": 30, "report": False}})
        solver.set_eq(FDM().laplacian(1.0, var) == rhs)
        torch.cuda.synchronize()
        dt = 0.1 * float(mesh.dx_list[0]) ** 2 / 6e-3
        euler_march(phi, 0.7, 1e-3, dt, 400)                 # asynchronous: ~400 x (step + BC fill) queued
        side = torch.cuda.Stream() if two_streams else torch.cuda.current_stream()
        with torch.cuda.stream(side), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
        torch.cuda.synchronize()
        return phi().clone(), var().clone(), rep

    phi_a, x_a, rep_a = run(False)
    phi_b, x_b, rep_b = run(True)
    assert rep_a == rep_b
    assert bit_equal(phi_a, phi_b) and bit_equal(x_a, x_b)


def test_roctx_ranges_do_not_change_anything(tmp_path):
    """PYAPES_HIP_ROCTX=1 wraps the solver phases in roctx ranges (libroctx64 resolved at run time; visible with
    rocprofv3 --marker-trace).  Here: the switch loads the library, the solve runs and gives the same bits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, warnings; warnings.simplefilter('ignore')\n"
            "from helpers import product_solve\n"
            "case = {'lower': [0, 0, 0], 'upper': [1, 1, 1], 'spacing': [40, 36, 72], 'dtype': 'double', 'method': 'cg', 'tol': 1e-30,\n"
            "        'bcs': [['dirichlet', 0.0], ['neumann', 0.5], ['dirichlet', 0.0], ['neumann', 0.0], ['dirichlet', 1.0], ['neumann', -0.25]]}\n"
            "g = torch.Generator().manual_seed(1); rhs = torch.randn((1, 40, 36, 72), generator=g, dtype=torch.float64)\n"
            "x, rep, _ = product_solve(case, rhs, 12)\n"
            "torch.save({'x': x, 'tol': rep['tol']}, sys.argv[1])\n" % (root, os.path.join(root, "tests")))
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"r{flag}.pt")
        env = dict(os.environ, PYAPES_HIP_ROCTX=flag, PYAPES_HIP_RESIDENT="0",
                   PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle"), os.environ.get("PYTHONPATH", "")]))
        r = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(out))
    assert torch.equal(outs[0]["x"], outs[1]["x"]) and outs[0]["tol"] == outs[1]["tol"]
