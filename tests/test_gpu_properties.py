"""-m gpu: BASELINE.json's FULL sizes, through size-independent properties (the oracle cannot run
there in seconds): exact discrete eigen-solution, constants in the null space, linearity,
conservation on the periodic ring, fast path == generic path, run-to-run bitwise determinism."""
import os
import warnings
from math import cos, pi

import pytest
import torch

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.hip import lib as L
from pyapes_amd.hip.context import HipContext
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs


def _fixed_cg(mesh, bcs, rhs, K, fast=True, method="cg"):
    from pyapes_amd.hip.context import context_for
    ctx = context_for(mesh)
    ctx.set_option("fastpath", fast)                   # tiled kernels, or the generic ones
    try:
        var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
        # K iterations: CG and Jacobi run max_it + 1 (linalg.py:144-157), BiCGSTAB max_it
        solver = Solver({"fdm": {"method": method, "tol": -1.0, "max_it": K if method == "bicgstab" else K - 1,
                                 "report": False}})
        solver.set_eq(FDM().laplacian(1.0, var) == rhs.clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = solver.solve()
    finally:
        ctx.set_option("fastpath", True)
    return var(), rep


def test_c2_256_dirichlet_discrete_eigen_solution():
    """3-D Poisson 256^3 fp64, Dirichlet 0, rhs = sin(pi x) sin(pi y) sin(pi z): an eigenvector of the
    discrete operator, so CG must land on rhs / lambda_h (known in closed form) in 1-2 iterations."""
    n = 256
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double")
    var = Field("p", 1, mesh, {"domain": homogeneous_bcs(3, 0.0, "dirichlet"), "obstacle": None})
    rhs = (torch.sin(pi * mesh.X) * torch.sin(pi * mesh.Y) * torch.sin(pi * mesh.Z)).unsqueeze(0).contiguous()
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-8, "max_it": 200, "report": False}})
    b = rhs.clone()
    solver.set_eq(FDM().laplacian(1.0, var) == b)
    rep = solver.solve()
    h = 1.0 / (n - 1)
    lam = 3 * (2 * cos(pi * h) - 2) / h ** 2
    exact = rhs / lam
    S = (slice(1, -1),) * 3
    err = float(torch.linalg.norm((var()[0] - exact[0])[S]) / torch.linalg.norm(exact[0][S]))
    assert rep["converge"] and rep["itr"] <= 3, rep
    assert err < 1e-10, err
    assert float(var()[0][0].abs().max()) == 0.0 and float(var()[0][:, :, -1].abs().max()) == 0.0


def test_c3_512_periodic_operator_properties():
    n = 512
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double")
    var = Field("p", 1, mesh, {"domain": homogeneous_bcs(3, None, "periodic"), "obstacle": None})
    ctx = mesh._hip = HipContext(mesh)
    ctx.bind_bcs(var(), var.bcs, 0)
    ctx.set_terms([{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": None}])
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, n, n), generator=g, dtype=torch.float64, device="cuda")
    y = torch.randn((n, n, n), generator=g, dtype=torch.float64, device="cuda")
    ax, ay = ctx.aop(x), ctx.aop(y)
    # constants are in the null space, exactly
    assert float(ctx.aop(torch.full_like(x, 3.25)).abs().max()) == 0.0
    # conservation on the ring: sum A x = 0 up to rounding
    assert abs(float(ax.sum())) <= 1e-9 * float(ax.abs().sum())
    # linearity
    lin = ctx.aop(2.0 * x - 0.5 * y)
    assert float((lin - (2.0 * ax - 0.5 * ay)).abs().max()) <= 1e-9 * float(ax.abs().max())
    # symmetry of the ring operator: <y, A x> = <x, A y>
    a, b = float((y * ax).sum()), float((x * ay).sum())
    assert abs(a - b) <= 1e-10 * abs(a)
    del ax, ay, lin, x, y
    mesh._hip = None


@pytest.mark.parametrize("workload", ["c3_512_f64_periodic", "c2_256_f64_dirichlet", "c5_1024x1024x512_f32_mixed"])
def test_full_size_fast_equals_generic_and_is_deterministic(workload):
    if workload.startswith("c3"):
        n, dtype, up = [512, 512, 512], "double", [1, 1, 1]
        bcs = homogeneous_bcs(3, None, "periodic")
    elif workload.startswith("c2"):
        n, dtype, up = [256, 256, 256], "double", [1, 1, 1]
        bcs = homogeneous_bcs(3, 0.0, "dirichlet")
    else:
        n, dtype, up = [1024, 1024, 512], "single", [1, 1, 0.5]
        bcs = mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])
    mesh = Mesh(Box([0, 0, 0], up), None, n, "cuda", dtype)
    g = torch.Generator(device="cuda").manual_seed(0)
    rhs = torch.randn((1, *n), generator=g, dtype=mesh.dtype.float, device="cuda")
    if workload.startswith("c3"):
        rhs -= rhs.mean()
    K = 12
    xf, rf = _fixed_cg(mesh, bcs, rhs, K, True)
    xf2, rf2 = _fixed_cg(mesh, bcs, rhs, K, True)
    assert torch.equal(xf, xf2) and rf["tol"] == rf2["tol"], "fused CG is not run-to-run deterministic"
    del xf2
    xg, rg = _fixed_cg(mesh, bcs, rhs, K, False)
    assert rf["itr"] == rg["itr"] == K
    rel = float(torch.linalg.norm((xf - xg).double()) / torch.linalg.norm(xg.double()))
    assert rel <= (1e-12 if dtype == "double" else 1e-5), rel
    assert abs(rf["tol"] - rg["tol"]) <= (1e-10 if dtype == "double" else 1e-4) * abs(rg["tol"])


@pytest.mark.parametrize("method", ["bicgstab", "jacobi"])
@pytest.mark.parametrize("workload", ["c3_512_f64_periodic", "c2_256_f64_dirichlet"])
def test_full_size_bicgstab_and_jacobi(workload, method):
    """Round 4: the other two solver loops at BASELINE's sizes -- BiCGSTAB with the s-less sequence (the tiled s / t phase
    stores t alone, k_bicg_x re-forms s and walks contiguous ranges backwards) and Jacobi with sweeps marching in
    alternating directions: run-to-run every bit equal, equal to the generic kernels to rounding (Jacobi: every bit), and
    for BiCGSTAB every bit equal to the sequence that stores s (option bicg_srv 0)."""
    from pyapes_amd.hip.context import context_for
    if workload.startswith("c3"):
        n, bcs = [512, 512, 512], homogeneous_bcs(3, None, "periodic")
    else:
        n, bcs = [256, 256, 256], homogeneous_bcs(3, 0.0, "dirichlet")
    mesh = Mesh(Box([0, 0, 0], [1, 1, 1]), None, n, "cuda", "double")
    g = torch.Generator(device="cuda").manual_seed(2)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64, device="cuda")
    if workload.startswith("c3"):
        rhs -= rhs.mean()
    K = 6 if method == "bicgstab" else 7     # (Jacobi: an odd count, the last sweep marches forwards again)
    xf, rf = _fixed_cg(mesh, bcs, rhs, K, True, method)
    xf2, rf2 = _fixed_cg(mesh, bcs, rhs, K, True, method)
    assert torch.equal(xf, xf2) and rf["tol"] == rf2["tol"], "not run-to-run deterministic"
    del xf2
    if method == "bicgstab":
        ctx = context_for(mesh)
        ctx.set_option("bicg_srv", 0)
        try:
            xs, rs = _fixed_cg(mesh, bcs, rhs, K, True, method)
        finally:
            ctx.set_option("bicg_srv", 1)
        assert torch.equal(xf, xs) and rf["tol"] == rs["tol"], "re-forming s changed a bit"
        del xs
    xg, rg = _fixed_cg(mesh, bcs, rhs, K, False, method)
    assert rf["itr"] == rg["itr"] == K
    assert bool(torch.isfinite(xf).all())
    if method == "jacobi":
        assert torch.equal(xf, xg)
    else:
        rel = float(torch.linalg.norm(xf - xg) / torch.linalg.norm(xg))
        assert rel <= 1e-11, rel
    assert abs(rf["tol"] - rg["tol"]) <= 1e-9 * abs(rg["tol"])


SIZE_SWITCHED = {
    # what switches on BY SIZE, at the default thresholds (round 3 only ever forced these paths onto small meshes, and the
    # one data-corruption bug of that round surfaced in bench_ops.py, not in a test -- VERDICT r03 weak #5):
    "2d_4096_f64": ([4096, 4096], "double", "dirichlet"),    # k_cg2d (>= 1.5 M cells), 128 MiB arrays: the placement search is on
    "2d_4097_f64": ([4097, 4097], "double", "dirichlet"),    # ... with odd rows: PITCH layout of k_cg2d
    "3d_257_f64": ([257, 257, 257], "double", "mixed"),      # odd rows in 3-D: PITCH layout of k_cg3d (CG and BiCGSTAB)
    "3d_513_f32": ([513, 513, 513], "single", "mixed"),      # fp32 odd rows: PITCH for CG, one cell per lane (NARROW) for BiCGSTAB / Jacobi
}


@pytest.mark.parametrize("method", ["cg", "bicgstab", "jacobi"])
@pytest.mark.parametrize("case", list(SIZE_SWITCHED), ids=list(SIZE_SWITCHED))
def test_size_switched_paths_at_their_default_thresholds(case, method):
    """K = 8 iterations on the meshes where k_cg2d, the PITCH layout, the one-cell-per-lane kernels and the online
    placement search switch on by themselves: the tiled path twice (every bit equal, run to run) and the generic
    kernels once (equal to rounding, identical counts)."""
    n, dtype, kind = SIZE_SWITCHED[case]
    nd = len(n)
    if kind == "dirichlet":
        bcs = homogeneous_bcs(nd, 0.0, "dirichlet")
    else:
        bcs = mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, n, "cuda", dtype)
    g = torch.Generator(device="cuda").manual_seed(4)
    rhs = torch.randn((1, *n), generator=g, dtype=mesh.dtype.float, device="cuda")
    K = 8
    xf, rf = _fixed_cg(mesh, bcs, rhs, K, True, method)
    xf2, rf2 = _fixed_cg(mesh, bcs, rhs, K, True, method)
    assert torch.equal(xf, xf2) and rf["tol"] == rf2["tol"], "not run-to-run deterministic"
    del xf2
    xg, rg = _fixed_cg(mesh, bcs, rhs, K, False, method)
    assert rf["itr"] == rg["itr"] == K
    assert bool(torch.isfinite(xf).all())
    rel = float(torch.linalg.norm((xf - xg).double()) / torch.linalg.norm(xg.double()))
    if method == "jacobi":
        assert torch.equal(xf, xg)       # no global sum feeds back into a Jacobi iterate: the same bits on every path
    else:
        assert rel <= (1e-11 if dtype == "double" else 1e-5), rel
    assert abs(rf["tol"] - rg["tol"]) <= (1e-9 if dtype == "double" else 1e-4) * abs(rg["tol"])
    if method == "cg" and case == "2d_4096_f64":
        from pyapes_amd.hip.context import context_for
        assert context_for(mesh).place_stats()["state"] in ("searching", "done")     # on by default from 128 MiB arrays


@pytest.mark.parametrize("speed", ["scalar", "tensor"])
def test_c4_256_f32_euler_march_full_size(speed):
    """BASELINE config 4 at its full size: 256^3 fp32, Div(upwind) + Laplacian explicit Euler march,
    Neumann (x) / Symmetry (y, z).  The three kernel paths -- k_sf, k_cg3d's Euler phase, the generic kernel --
    must give the same bits after 30 steps; and the scheme's own properties hold: with u >= 0, nu dt /dx^2
    and u dt / dx inside the monotone range the update is a convex combination of neighbours (discrete
    maximum principle: no new extrema), a constant state stays constant."""
    from pyapes_amd.hip.context import context_for
    from pyapes_amd.solver.march import euler_march
    n = 256
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "single")
    bcs = mixed_bcs([0.0, 0.0, None, None, None, None], ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
    phi0 = torch.exp(-((mesh.X - 0.5) ** 2 + (mesh.Y - 0.5) ** 2 + (mesh.Z - 0.5) ** 2) / 0.02).unsqueeze(0).contiguous()
    u = 1.0 if speed == "scalar" else (0.25 + 0.75 * torch.rand((1, n, n, n), device="cuda", dtype=torch.float32,
                                                                generator=torch.Generator(device="cuda").manual_seed(1)))
    nu, dx = 1e-3, mesh.dx_list[0]
    dt = 0.2 * min(dx * dx / (6 * nu), dx / 1.0)
    cfg = {"div": {"limiter": "upwind"}}
    ctx = context_for(mesh)
    out = {}
    for path, (fast, sf) in {"k_sf": (True, True), "k_cg3d": (True, False), "generic": (False, False)}.items():
        ctx.set_option("fastpath", fast)
        ctx.set_option("sf", sf)
        try:
            phi = Field("phi", 1, mesh, {"domain": bcs, "obstacle": None})
            phi.set_var_tensor(phi0.clone())
            phi.apply_bcs()
            euler_march(phi, u, nu, dt, 30, cfg)
            out[path] = phi().clone()
        finally:
            ctx.set_option("fastpath", True)
            ctx.set_option("sf", True)
    assert torch.equal(out["k_sf"], out["k_cg3d"]) and torch.equal(out["k_sf"], out["generic"])
    x = out["k_sf"]
    assert bool(torch.isfinite(x).all())
    assert float(x.max()) <= float(phi0.max()) * (1 + 1e-6) and float(x.min()) >= -1e-6, (float(x.max()), float(x.min()))
    assert float(x.max()) < float(phi0.max())          # it did diffuse / move
    const = Field("c", 1, mesh, {"domain": bcs, "obstacle": None})
    const.set_var_tensor(torch.full_like(phi0, 0.7))
    const.apply_bcs()
    euler_march(const, u, nu, dt, 10, cfg)
    # (not to the last bit: the Neumann fill 4/3 p - 1/3 p rounds, bcs.py:246-253)
    assert float((const() - 0.7).abs().max()) <= 2e-6
