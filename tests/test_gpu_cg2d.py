"""-m gpu: k_cg2d (csrc/pa_cg2d_kernel.h) -- the CG phases of 2-D meshes marching along the slow axis -- against the
one-plane tiling of k_cg3d it replaces on large 2-D meshes (same arithmetic per node, other grouping of the partial
sums: 1e-12 fp64, identical iteration counts and scalars to 1e-10) and against the literal oracle.  The kernel is
forced onto small meshes here (option cg2d_mincells = 0); by default it takes 2-D meshes of >= 1.5 M cells."""
import warnings

import pytest
import torch

import pyapes_oracle as O
from helpers import rel_err
from pyapes_amd.geometry import Box
from pyapes_amd.hip.context import context_for
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

pytestmark = pytest.mark.gpu

D = lambda v=0.0: ("dirichlet", v)   # noqa: E731
N = lambda v=0.0: ("neumann", v)     # noqa: E731
SY = ("symmetry", None)
PE = ("periodic", None)
CASES = [
    ("dirichlet_f64", [96, 640], "double", [D(0.0), D(1.0), D(0.0), D(0.5)], 9),
    ("mixed_f64", [70, 516], "double", [N(0.3), D(0.0), SY, N(-0.25)], 9),         # rows not a multiple of anything
    ("slow_axis_periodic_f64", [64, 512], "double", [PE, PE, D(0.0), N(0.2)], 9),  # the march axis wraps around
    ("fast_axis_periodic_f64", [48, 256], "double", [D(0.0), N(0.1), PE, PE], 9),  # edge cells wrap around
    ("all_periodic_f64", [40, 128], "double", [PE] * 4, 7),
    ("mixed_f32", [80, 1028], "single", [D(0.0), N(0.0), D(1.0), N(0.0)], 8),
    ("odd_rows_f64", [65, 1025], "double", [D(0.0), D(1.0), N(0.0), D(0.5)], 9),   # PITCH layout
    ("odd_rows_f32", [33, 515], "single", [SY, D(1.0), D(0.0), N(0.5)], 8),
    ("narrow_strip_f64", [200, 12], "double", [D(0.0)] * 4, 8),                     # one strip, mostly empty lanes
]


def _solve(n, dtype, faces, K, mincells, rhs0):
    mesh = Mesh(Box([0.0, 0.0], [1.0, 1.0]), None, n, "cuda", dtype)
    ctx = context_for(mesh)
    ctx.set_option("cg2d_mincells", mincells)
    ctx.set_option("resident", False)
    bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(faces)]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    s = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": K, "report": False}})
    s.set_eq(-FDM().laplacian(0.7, var) == rhs0.to(mesh.dtype.float).cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep, ctx.scalars()


@pytest.mark.parametrize("name,n,dtype,faces,K", CASES, ids=[c[0] for c in CASES])
def test_cg2d_vs_one_plane_tiling_and_oracle(name, n, dtype, faces, K):
    g = torch.Generator().manual_seed(5)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if all(t == "periodic" for t, _ in faces):
        rhs0 -= rhs0.mean()
    x_m, rep_m, sc_m = _solve(n, dtype, faces, K, 0, rhs0)       # marching kernel
    x_t, rep_t, sc_t = _solve(n, dtype, faces, K, -1, rhs0)      # one plane of k_cg3d's tiling
    f64 = dtype == "double"
    assert rep_m["itr"] == rep_t["itr"] == K + 1
    assert rel_err(x_m, x_t) < (1e-12 if f64 else 2e-6), rel_err(x_m, x_t)
    for key in ("alpha", "beta", "tol"):
        assert abs(sc_m[key] - sc_t[key]) <= (1e-10 if f64 else 1e-4) * abs(sc_t[key]), key
    om = O.OMesh([0.0, 0.0], [1.0, 1.0], n, dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(faces)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(om, cfg, rhs0.to(om.dtype).clone(), method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    assert ro["itr"] == rep_m["itr"]
    assert rel_err(x_m, xo) < (1e-10 if f64 else 1e-5), rel_err(x_m, xo)


def test_cg2d_is_the_default_on_large_meshes_and_deterministic():
    """2048 x 1024 (2 M cells) takes the marching kernel by default; two runs give the same bits, and the converged
    solution of the reference's 2-D Poisson problem (poisson_bcs(2), tests/test_solver.py:34) is the exact one"""
    from pyapes_amd.testing.poisson import poisson_bcs, poisson_exact_nd, poisson_rhs_nd
    out = []
    for _ in range(2):
        mesh = Mesh(Box([0.0, 0.0], [1.0, 1.0]), None, [2048, 1024], "cuda", "double")
        var = Field("p", 1, mesh, {"domain": poisson_bcs(2), "obstacle": None})
        s = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": 60, "report": False}})
        s.set_eq(FDM().laplacian(1.0, var) == poisson_rhs_nd(mesh, var))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        out.append((var().clone(), rep["tol"]))
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert bool(torch.isfinite(out[0][0]).all())


# ---- second session of round 3: the Jacobi sweep (phase 4) and the BiCGSTAB phases 6 / 8 on the marching kernel -------
def _solve_m(n, dtype, faces, K, mincells, rhs0, method, omega=0.9):
    mesh = Mesh(Box([0.0, 0.0], [1.0, 1.0]), None, n, "cuda", dtype)
    ctx = context_for(mesh)
    ctx.set_option("cg2d_mincells", mincells)
    ctx.set_option("resident", False)
    bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(faces)]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    cfg = {"method": method, "tol": 1e-30, "max_it": K, "report": False}
    if method == "jacobi":
        cfg["omega"] = omega
    s = Solver({"fdm": cfg})
    s.set_eq(-FDM().laplacian(0.7, var) == rhs0.to(mesh.dtype.float).cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep, ctx.scalars()


@pytest.mark.parametrize("name,n,dtype,faces,K", CASES, ids=[c[0] for c in CASES])
def test_cg2d_jacobi_is_bit_identical_to_the_one_plane_tiling(name, n, dtype, faces, K):
    """no global sum feeds back into a Jacobi iterate: the same arithmetic per node must give the same bits"""
    g = torch.Generator().manual_seed(7)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    x_m, rep_m, _ = _solve_m(n, dtype, faces, 2 * K, 0, rhs0, "jacobi")
    x_t, rep_t, _ = _solve_m(n, dtype, faces, 2 * K, -1, rhs0, "jacobi")
    assert rep_m["itr"] == rep_t["itr"] == 2 * K + 1
    assert torch.equal(x_m, x_t), float((x_m - x_t).abs().max())
    assert abs(rep_m["tol"] - rep_t["tol"]) <= 1e-10 * abs(rep_t["tol"])      # (the stop-test sum is grouped differently)
    om = O.OMesh([0.0, 0.0], [1.0, 1.0], n, dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(faces)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(om, cfg, rhs0.to(om.dtype).clone(), method="jacobi", tol=1e-30, max_it=2 * K, coeff=0.7,
                                 sign=-1.0, omega=0.9)
    assert ro["itr"] == rep_m["itr"]
    assert rel_err(x_m, xo) < (1e-10 if dtype == "double" else 1e-5), rel_err(x_m, xo)


@pytest.mark.parametrize("name,n,dtype,faces,K", CASES, ids=[c[0] for c in CASES])
def test_cg2d_bicgstab_vs_one_plane_tiling_and_oracle(name, n, dtype, faces, K):
    g = torch.Generator().manual_seed(8)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if all(t == "periodic" for t, _ in faces):
        rhs0 -= rhs0.mean()
    f64 = dtype == "double"
    om = O.OMesh([0.0, 0.0], [1.0, 1.0], n, dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(faces)]
    for k in (1, 2, 5):
        x_m, rep_m, sc_m = _solve_m(n, dtype, faces, k, 0, rhs0, "bicgstab")
        x_t, rep_t, sc_t = _solve_m(n, dtype, faces, k, -1, rhs0, "bicgstab")
        assert rep_m["itr"] == rep_t["itr"] == k
        assert rel_err(x_m, x_t) < (1e-11 if f64 else 5e-6), (k, rel_err(x_m, x_t))
        for key in ("alpha", "omega", "tol"):
            assert abs(sc_m[key] - sc_t[key]) <= (1e-9 if f64 else 1e-3) * abs(sc_t[key]), (k, key)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xo, ro = O.solve_poisson(om, cfg, rhs0.to(om.dtype).clone(), method="bicgstab", tol=1e-30, max_it=k, coeff=0.7,
                                     sign=-1.0)
        assert ro["itr"] == rep_m["itr"]
        assert rel_err(x_m, xo) < (1e-10 if f64 else 1e-5), (k, rel_err(x_m, xo))
