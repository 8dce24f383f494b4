"""-m gpu: CG on meshes whose row length is NOT a multiple of the 16-byte vector -- the normal case of a
node-based mesh (11, 101, 2^k + 1 nodes; reference: pyapes/mesh/_mesh.py:67-93) -- through the PITCH layout of the
tiled phases (csrc/pa_cg3d_kernel.h: r and the direction buffers, which the ctx owns, get a padded row pitch; only
the caller's x is touched cell by cell).  Checked against the one-cell-per-lane (NARROW) kernels it replaces there
(same arithmetic per node, other grouping of the partial sums: 1e-12, identical iteration counts) and against the
literal oracle; the reference goldens with odd rows (cg2d_poisson_n100 is even; cg3d_mix33, cg2d_xper101,
cg2d_poisson100 = 101 nodes) run through it in tests/test_gpu_parity_golden.py."""
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
    # name, n, dtype, faces (factory order), K
    ("3d_dirichlet_f64", [20, 18, 131], "double", [D(0.0), D(0.5), D(0.0), D(0.0), D(1.0), D(0.0)], 9),
    ("3d_mixed_f64", [17, 21, 65], "double", [N(0.3), D(0.0), D(0.3), N(0.0), SY, N(-0.25)], 9),
    ("3d_xyper_f64", [16, 12, 33], "double", [PE, PE, PE, PE, D(0.0), N(0.1)], 7),       # periodic slow axes, odd rows
    ("3d_mixed_f32", [12, 16, 257], "single", [D(0.0), N(0.0), D(0.0), N(0.0), D(1.0), N(0.0)], 8),
    ("3d_odd_not_mult4_f32", [10, 9, 130], "single", [D(0.0)] * 6, 8),                       # 130 = 2 mod 4
    ("2d_dirichlet_f64", [65, 1025], "double", [D(0.0), D(1.0), D(0.0), D(0.5)], 9),
    ("2d_xper_f64", [40, 101], "double", [PE, PE, D(0.0), N(0.2)], 9),
    ("3d_short_rows_f64", [9, 8, 5], "double", [D(0.0)] * 6, 6),                             # n2 = 5: three vectors, one padded
]


def _solve(n, dtype, faces, K, pitch, rhs0):
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, n, "cuda", dtype)
    ctx = context_for(mesh)
    ctx.set_option("pitch", pitch)
    ctx.set_option("resident", False)      # the launch-per-phase loop is what is under test
    bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(faces)]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    s = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": K, "report": False}})
    s.set_eq(-FDM().laplacian(0.7, var) == rhs0.to(mesh.dtype.float).cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep, ctx.scalars()


@pytest.mark.parametrize("name,n,dtype,faces,K", CASES, ids=[c[0] for c in CASES])
def test_pitched_cg_vs_narrow_and_oracle(name, n, dtype, faces, K):
    g = torch.Generator().manual_seed(3)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if all(t == "periodic" for t, _ in faces[:2 * len(n)]):
        rhs0 -= rhs0.mean()
    x_p, rep_p, sc_p = _solve(n, dtype, faces, K, True, rhs0)
    x_n, rep_n, sc_n = _solve(n, dtype, faces, K, False, rhs0)
    tight = 1e-12 if dtype == "double" else 2e-6
    assert rep_p["itr"] == rep_n["itr"] == K + 1
    assert rel_err(x_p, x_n) < tight, rel_err(x_p, x_n)
    for key in ("alpha", "beta", "tol"):
        assert abs(sc_p[key] - sc_n[key]) <= (1e-10 if dtype == "double" else 1e-4) * abs(sc_n[key]), key
    nd = len(n)
    om = O.OMesh([0.0] * nd, [1.0] * nd, n, dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(faces)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(om, cfg, rhs0.to(om.dtype).clone(), method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    assert ro["itr"] == rep_p["itr"]
    assert rel_err(x_p, xo) < (1e-10 if dtype == "double" else 1e-5), rel_err(x_p, xo)


def test_pitched_layout_is_really_taken(monkeypatch, capfd):
    """PYAPES_HIP_DEBUG prints the launch shape of the first tiled launches: odd rows on one GPU must say (pitched)"""
    monkeypatch.setenv("PYAPES_HIP_DEBUG", "1")
    import subprocess, sys, os
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, warnings; warnings.simplefilter('ignore')\n"
            "from test_gpu_pitched import _solve\n"
            "_solve([20, 18, 131], 'double', [('dirichlet', 0.0)] * 6, 3, True, torch.ones(1, 20, 18, 131, dtype=torch.float64))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=os.pathsep.join(
        [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), os.environ.get("PYTHONPATH", "")])))
    assert p.returncode == 0, p.stderr[-2000:]
    assert "phase A (pitched)" in p.stderr and "phase B (pitched)" in p.stderr, p.stderr[-2000:]


# ---- BiCGSTAB (round 3, second session): all eight solver arrays pitched, phases 5 / 6 / 8 on 16-byte lanes ----------
BICG_CASES = [
    ("3d_dirichlet_f64", [20, 18, 131], "double", [D(0.0), D(0.5), D(0.0), D(0.0), D(1.0), D(0.0)]),
    ("3d_mixed_f64", [17, 21, 65], "double", [N(0.3), D(0.0), D(0.3), N(0.0), SY, N(-0.25)]),
    ("3d_xyper_f64", [16, 12, 33], "double", [PE, PE, PE, PE, D(0.0), N(0.1)]),
    ("3d_mixed_f32", [12, 16, 257], "single", [D(0.0), N(0.0), D(0.0), N(0.0), D(1.0), N(0.0)]),
    ("3d_odd_not_mult4_f32", [10, 9, 130], "single", [D(0.0)] * 6),
    ("2d_dirichlet_f64", [65, 1025], "double", [D(0.0), D(1.0), D(0.0), D(0.5)]),
    ("3d_short_rows_f64", [9, 8, 5], "double", [D(0.0)] * 6),
]


def _solve_bicg(n, dtype, faces, K, pitch, rhs0, tol=1e-30, pfold=True, monkeypatch=None):
    if monkeypatch is not None:
        from helpers import hip_options
        hip_options(monkeypatch, bicg_pfold=pfold)
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, n, "cuda", dtype)
    ctx = context_for(mesh)
    ctx.set_option("pitch", pitch)
    ctx.set_option("resident", False)
    bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(faces)]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    s = Solver({"fdm": {"method": "bicgstab", "tol": tol, "max_it": K, "report": False}})
    s.set_eq(-FDM().laplacian(0.7, var) == rhs0.to(mesh.dtype.float).cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep, ctx.scalars()


@pytest.mark.parametrize("name,n,dtype,faces", BICG_CASES, ids=[c[0] for c in BICG_CASES])
def test_pitched_bicgstab_vs_narrow_and_oracle(name, n, dtype, faces, monkeypatch):
    g = torch.Generator().manual_seed(4)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    nd = len(n)
    om = O.OMesh([0.0] * nd, [1.0] * nd, n, dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(faces)]
    for K in (1, 2, 5):
        x_p, rep_p, sc_p = _solve_bicg(n, dtype, faces, K, True, rhs0, monkeypatch=monkeypatch)
        x_n, rep_n, sc_n = _solve_bicg(n, dtype, faces, K, False, rhs0, monkeypatch=monkeypatch)
        x_q, rep_q, _ = _solve_bicg(n, dtype, faces, K, True, rhs0, pfold=False, monkeypatch=monkeypatch)
        assert rep_p["itr"] == rep_n["itr"] == rep_q["itr"] == K
        assert torch.equal(x_p, x_q) and rep_p["tol"] == rep_q["tol"]      # the folded direction update changes no bit
        tight = 1e-11 if dtype == "double" else 5e-6
        assert rel_err(x_p, x_n) < tight, (K, rel_err(x_p, x_n))
        for key in ("alpha", "omega", "tol"):
            assert abs(sc_p[key] - sc_n[key]) <= (1e-9 if dtype == "double" else 1e-3) * abs(sc_n[key]), (K, key)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xo, ro = O.solve_poisson(om, cfg, rhs0.to(om.dtype).clone(), method="bicgstab", tol=1e-30, max_it=K, coeff=0.7,
                                     sign=-1.0)
        assert ro["itr"] == rep_p["itr"]
        assert rel_err(x_p, xo) < (1e-10 if dtype == "double" else 1e-5), (K, rel_err(x_p, xo))


def test_pitched_bicgstab_is_really_taken():
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, warnings; warnings.simplefilter('ignore')\n"
            "from test_gpu_pitched import _solve_bicg\n"
            "_solve_bicg([20, 18, 131], 'double', [('dirichlet', 0.0)] * 6, 3, True, torch.ones(1, 20, 18, 131, dtype=torch.float64))\n"
            % (root, os.path.join(root, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, PYAPES_HIP_DEBUG="1", PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle"), os.environ.get("PYTHONPATH", "")])))
    assert p.returncode == 0, p.stderr[-2000:]
    # phases 5 (first iteration), 6 and 8 print as 'F', 'G', 'I' ('A' + PHASE)
    assert "phase F (pitched)" in p.stderr and "phase G (pitched)" in p.stderr and "phase I (pitched)" in p.stderr, p.stderr[-2000:]
