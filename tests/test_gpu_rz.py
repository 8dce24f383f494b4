"""-m gpu: axisymmetric (Cylinder, rz) meshes -- SURVEY 8f rank 4.  The golden rz cases run through the
generic parity tests (test_gpu_parity_golden / test_gpu_spatial); here: the reference's own
test_poisson_rz at full size, the API surface, Jacobi / Euler on rz against the oracle, the
periodic-face errors."""
import warnings

import pytest
import torch

import pyapes_oracle as O
from conftest import golden_cases, golden_load
from helpers import rel_err

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Cylinder
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdc import FDC
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.march import euler_step
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_rz_bcs, poisson_rz_exact, poisson_rz_rhs
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import CylinderBoundary


def test_poisson_rz_reference_test():
    """tests/test_solver.py:309-358 verbatim in structure: 101 x 101, BiCGSTAB tol 1e-5 -> exp(-z) cos(r)"""
    mesh = Mesh(Cylinder[0:1, 0:1], None, [101, 101], "cuda", "double")
    assert mesh.coord_sys == "rz" and mesh.Y.numel() == 0 and mesh.Z is mesh.grid[1] and mesh.R is mesh.grid[0]
    cfg = poisson_rz_bcs()
    f_bc = CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0.0},
                            ru={"bc_type": "dirichlet", "bc_val": cfg[1]["bc_val"]},
                            zl={"bc_type": "dirichlet", "bc_val": cfg[2]["bc_val"]},
                            zu={"bc_type": "dirichlet", "bc_val": cfg[3]["bc_val"]})
    var = Field("U", 1, mesh, {"domain": f_bc(), "obstacle": None}, init_val=0.0)
    assert [bc.bc_face for bc in var.bcs] == ["rl", "ru", "zl", "zu"]
    assert [bc.bc_face_dim for bc in var.bcs] == [0, 0, 1, 1]
    solver = Solver({"fdm": {"method": "bicgstab", "tol": 1e-5, "max_it": 1000, "report": False}})
    rhs = poisson_rz_rhs(mesh, var)
    torch.testing.assert_close(rhs.cpu(), torch.as_tensor(golden_load("rz_bicg_poisson101_f64")["rhs0"]),
                               rtol=1e-14, atol=1e-15)   # sin / exp evaluated on the GPU: last-ulp differences
    solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    rep = solver.solve()
    assert rep["converge"]
    torch.testing.assert_close(var()[0], poisson_rz_exact(mesh), atol=1e-3, rtol=1e-3)
    # the reference needs 321 iterations; BiCGSTAB is summation-order sensitive, so a band, not equality
    assert abs(rep["itr"] - 321) <= 40, rep


@pytest.mark.parametrize("dtype", ["double", "single"])
def test_rz_jacobi_and_euler_vs_oracle(dtype):
    n = (19, 23)
    mo = O.OMesh([0.0, 0.0], [1.0, 1.5], list(n), dtype, "rz")
    cfg = O.mixed_cfg([0.0, 1.0, 0.0, 0.5], ["neumann", "dirichlet", "dirichlet", "neumann"], O.FACES_RZ)
    pcfg = [dict(c, bc_val_opt=None) for c in cfg]
    mesh = Mesh(Cylinder([0.0, 0.0], [1.0, 1.5]), None, list(n), "cuda", dtype)
    g = torch.Generator().manual_seed(2)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(mo.dtype)
    tol = 1e-10 if dtype == "double" else 2e-5
    # Jacobi, fixed iteration count
    xo, ro = O.solve_poisson(mo, cfg, rhs.clone(), method="jacobi", tol=-1.0, max_it=30)
    var = Field("p", 1, mesh, {"domain": pcfg, "obstacle": None})
    s = Solver({"fdm": {"method": "jacobi", "tol": -1.0, "max_it": 30, "report": False}})
    s.set_eq(FDM().laplacian(1.0, var) == rhs.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    assert rep["itr"] == ro["itr"]
    assert rel_err(var().cpu(), xo) < tol
    # explicit Euler step, upwind (intended) and central-free config: lap + div with the u phi / r term
    bcs = O.make_bcs(mo, cfg)
    phi0 = torch.exp(-((mo.grid[0] - 0.4) ** 2 + (mo.grid[1] - 0.7) ** 2) / 0.05).unsqueeze(0).to(mo.dtype)
    O.bc_fill(phi0, bcs)
    ut = (0.3 + 0.2 * torch.randn((1, *n), generator=g, dtype=torch.float64)).to(mo.dtype)
    for u in (0.8, ut):
        po = O.euler_step(phi0.clone(), u, 1e-2, 1e-3, mo, bcs, "upwind")
        v = Field("phi", 1, mesh, {"domain": pcfg, "obstacle": None})
        v.set_var_tensor(phi0.cuda().clone())
        euler_step(v, u.cuda() if isinstance(u, torch.Tensor) else u, 1e-2, 1e-3, {"div": {"limiter": "upwind"}})
        assert rel_err(v().cpu(), po) < (1e-13 if dtype == "double" else 1e-6)


def test_rz_periodic_faces_raise_like_the_reference():
    mesh = Mesh(Cylinder[0:1, 0:1], None, [8, 8], "cuda", "double")
    names = ["rl", "ru", "zl", "zu"]
    for tv, exc in (([("neumann", 0.0), ("dirichlet", 0.0), ("periodic", None), ("periodic", None)], IndexError),
                    ([("periodic", None), ("periodic", None), ("dirichlet", 0.0), ("dirichlet", 0.0)], KeyError)):
        cfg = [{"bc_face": f, "bc_type": t, "bc_val": v, "bc_val_opt": None} for f, (t, v) in zip(names, tv)]
        var = Field("U", 1, mesh, {"domain": cfg, "obstacle": None})
        s = Solver({"fdm": {"method": "bicgstab", "tol": 1e-5, "max_it": 10, "report": False}})
        s.set_eq(FDM().laplacian(1.0, var) == torch.zeros_like(var()))
        with pytest.raises(exc):
            s.solve()


def test_rz_mesh_rules():
    with pytest.raises(AssertionError):
        Cylinder([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    with pytest.raises(AssertionError):
        Cylinder([-0.1, 0.0], [1.0, 1.0])
    c = Cylinder[0:2, 0:3]
    assert c.type == "cylinder" and c.dim == 2 and abs(c.size - 3.141592653589793 * 4 * 3) < 1e-12
    assert [f["face"] for f in c.config.values()] == ["zl", "zu", "rl", "ru"]
    mesh = Mesh(c, None, [5, 7], "cuda", "single")
    with pytest.raises(IndexError):   # edge=True Div of a scalar field: 1-D only, like the reference
        v = Field("q", 1, mesh, {"domain": None, "obstacle": None})
        FDC({"div": {"limiter": "none", "edge": True}}).div(1.0, v)


@pytest.mark.parametrize("method", ["jacobi", "cg", "bicgstab"])
@pytest.mark.parametrize("dtype", ["double", "single"])
def test_rz_resident_lean_path_matches_the_generic_term_evaluation(method, dtype, monkeypatch):
    """Round 3: the resident solver's LEAN stencil on axisymmetric meshes (the r rows of pa_coord_set's table staged in
    LDS) against pa_apply_terms on the box (option res_rzlean 0) and the launch-per-phase kernels: the same
    arithmetic per node, so Jacobi is bit-identical and CG / BiCGSTAB agree to rounding with identical counts."""
    from pyapes_amd.hip.context import context_for
    n = (37, 45)
    cfg = O.mixed_cfg([0.0, 1.0, 0.3, 0.5], ["neumann", "dirichlet", "dirichlet", "neumann"], O.FACES_RZ)
    pcfg = [dict(c, bc_val_opt=None) for c in cfg]
    g = torch.Generator().manual_seed(5)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    # BiCGSTAB amplifies the other grouping of the partial sums (512 against 256 threads per box) by an order of
    # magnitude every few iterations: a short run for it
    K = 6 if method == "bicgstab" else 24
    out = {}
    from helpers import hip_options
    for name, env, lean in (("lean", {"PYAPES_HIP_RESIDENT": "1"}, 1), ("terms", {"PYAPES_HIP_RESIDENT": "1"}, 0),
                            ("launch", {"PYAPES_HIP_RESIDENT": "0"}, 1)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hip_options(monkeypatch, res_rzlean=lean)
        mesh = Mesh(Cylinder([0.0, 0.0], [1.0, 1.5]), None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": pcfg, "obstacle": None})
        s = Solver({"fdm": {"method": method, "tol": -1.0, "max_it": K, "report": False}})
        s.set_eq(FDM().laplacian(0.9, var) == rhs.to(mesh.dtype.float).cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        out[name] = (var().cpu().double(), rep["itr"], context_for(mesh).resident_used())
    assert out["lean"][2] > 0 and out["terms"][2] > 0 and out["launch"][2] == 0
    # max_it = K: K + 1 iterations for CG / Jacobi (`itr > max_it`, Q6), K for BiCGSTAB (`itr >= max_it`, linalg.py:264)
    assert out["lean"][1] == out["terms"][1] == out["launch"][1] == (K if method == "bicgstab" else K + 1)
    tol = 0.0 if method == "jacobi" else ((1e-12 if method == "cg" else 1e-10) if dtype == "double" else 2e-5)
    for other in ("terms", "launch"):
        err = rel_err(out["lean"][0], out[other][0])
        assert err <= tol, (other, err)


RZ_LARGE = {
    # beyond what the resident solver holds (128 boxes): the launch-per-phase loops, where round 4 put the axisymmetric
    # rows on the 2-D marching kernel (k_cg2d<..., RZ>: r is its march axis, a coefficient triple per row)
    "even_rows_f64": ((520, 1024), "double"),
    "odd_rows_pitched_f64": ((513, 1025), "double"),     # PITCH layout: r / d (BiCGSTAB: every solver array) with padded rows
    "even_rows_f32": ((384, 1536), "single"),
}


@pytest.mark.parametrize("method", ["cg", "bicgstab", "jacobi"])
@pytest.mark.parametrize("case", list(RZ_LARGE), ids=list(RZ_LARGE))
def test_rz_marching_kernel_matches_generic_kernels_and_oracle(case, method):
    """Axisymmetric meshes on k_cg2d<..., RZ> against the generic kernels (pa_apply_terms' rz branch: the same arithmetic
    per node, so the Jacobi sweep is bit-identical and CG / BiCGSTAB agree to rounding with identical counts) and against
    the oracle."""
    from pyapes_amd.hip.context import context_for
    n, dtype = RZ_LARGE[case]
    cfg = O.mixed_cfg([0.0, 1.0, 0.3, 0.5], ["neumann", "dirichlet", "dirichlet", "neumann"], O.FACES_RZ)
    pcfg = [dict(c, bc_val_opt=None) for c in cfg]
    g = torch.Generator().manual_seed(9)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    K = 6
    out = {}
    for fast in (True, False):
        mesh = Mesh(Cylinder([0.0, 0.0], [1.0, 1.5]), None, list(n), "cuda", dtype)
        ctx = context_for(mesh)
        ctx.set_option("fastpath", fast)
        var = Field("p", 1, mesh, {"domain": pcfg, "obstacle": None})
        s = Solver({"fdm": {"method": method, "tol": -1.0, "max_it": K if method == "bicgstab" else K - 1, "report": False}})
        s.set_eq(FDM().laplacian(0.9, var) == rhs.to(mesh.dtype.float).cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        assert ctx.resident_used() == 0
        out[fast] = (var().cpu(), rep)
    (xf, rf), (xg, rg) = out[True], out[False]
    assert rf["itr"] == rg["itr"] == K
    assert bool(torch.isfinite(xf).all())
    if method == "jacobi" and n[1] % (2 if dtype == "double" else 4) == 0:
        assert torch.equal(xf, xg)
    rtol = 1e-11 if dtype == "double" else 1e-5
    assert rel_err(xf, xg) <= rtol, rel_err(xf, xg)
    mo = O.OMesh([0.0, 0.0], [1.0, 1.5], list(n), dtype, "rz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(mo, cfg, rhs.to(mo.dtype), method=method, tol=-1.0,
                                 max_it=K if method == "bicgstab" else K - 1, coeff=0.9)
    assert ro["itr"] == K
    assert rel_err(xf, xo) <= (1e-10 if dtype == "double" else 1e-5), rel_err(xf, xo)


@pytest.mark.parametrize("n,pitched", [((520, 1024), False), ((513, 1025), True)], ids=["even_rows", "odd_rows_pitched"])
def test_rz_marching_kernel_is_what_runs(n, pitched):
    """the launch geometry log names the kernel (PYAPES_HIP_DEBUG): CG phases of a large axisymmetric mesh on k_cg2d"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, warnings\nwarnings.simplefilter('ignore')\n"
            "from pyapes_amd.geometry import Cylinder\nfrom pyapes_amd.mesh import Mesh\nfrom pyapes_amd.variables import Field\n"
            "from pyapes_amd.solver.fdm import FDM\nfrom pyapes_amd.solver.ops import Solver\nimport pyapes_oracle as O\n"
            "cfg = [dict(c, bc_val_opt=None) for c in O.mixed_cfg([0.0, 1.0, 0.3, 0.5], ['neumann', 'dirichlet', 'dirichlet', 'neumann'], O.FACES_RZ)]\n"
            "mesh = Mesh(Cylinder([0.0, 0.0], [1.0, 1.5]), None, %r, 'cuda', 'double')\n"
            "var = Field('p', 1, mesh, {'domain': cfg, 'obstacle': None})\n"
            "s = Solver({'fdm': {'method': 'cg', 'tol': -1.0, 'max_it': 3, 'report': False}})\n"
            "s.set_eq(FDM().laplacian(1.0, var) == torch.ones_like(var()))\n"
            "s.solve()\n" % (root, os.path.join(root, "oracle"), list(n)))
    env = dict(os.environ, PYAPES_HIP_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if "k_cg2d phase" in ln]
    tag = " (pitched)" if pitched else ":"
    assert any(("phase A" + tag) in ln for ln in lines) and any(("phase B" + tag) in ln for ln in lines), r.stderr[-1500:]
