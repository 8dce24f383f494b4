"""-m gpu: the placement probe of the CG set-up (csrc/pa_solver.hip, cg_place_t).  On large solves the set-up times the
two phase kernels of the solve itself with r / d in a few alternative allocations and at a handful of offsets inside them -- with an EMPTY
interior set, so that phase A writes zeros into a buffer it would write anyway and phase B stores x back exactly as
loaded -- and keeps the fastest.  It must not change a single bit of any solve: here it is forced onto small meshes
(PYAPES_HIP_PLACE_MINBYTES=0) and compared with the probe switched off."""
import os
import subprocess
import sys
import warnings

import pytest
import torch

import pyapes_oracle as O
from pyapes_amd.geometry import Box
from pyapes_amd.hip.context import context_for
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

pytestmark = pytest.mark.gpu

D = lambda v=0.0: ("dirichlet", v)   # noqa: E731
N = lambda v=0.0: ("neumann", v)     # noqa: E731
PE = ("periodic", None)
CASES = [
    ("mixed_f64", [40, 36, 72], "double", [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)]),
    ("periodic_f64", [24, 20, 64], "double", [PE] * 6),
    ("odd_rows_pitched_f64", [20, 18, 131], "double", [D(0.0), D(0.5), D(0.0), D(0.0), D(1.0), D(0.0)]),
    ("mixed_f32", [18, 22, 132], "single", [D(0.0), N(0.0), D(0.0), N(0.0), D(1.0), N(0.0)]),
    ("2d_f64", [96, 640], "double", [D(0.0), D(1.0), N(0.0), D(0.5)]),
    ("2d_odd_rows_f64", [65, 1025], "double", [D(0.0), D(1.0), D(0.0), D(0.5)]),
]


def _solve(n, dtype, faces, place, rhs0, x0, K=9):
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, n, "cuda", dtype)
    ctx = context_for(mesh)
    ctx.set_option("place", place)
    if nd == 2:
        ctx.set_option("cg2d_mincells", 0)     # the marching kernel (what a 2-D mesh large enough for the probe runs)
    ctx.set_option("resident", False)
    bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(faces)]
    var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
    var.set_var_tensor(x0.to(mesh.dtype.float).cuda())
    out = []
    for k in (K, 3):      # the second solve on the same context re-uses the remembered placement
        s = Solver({"fdm": {"method": "cg", "tol": 1e-30, "max_it": k, "report": False}})
        s.set_eq(-FDM().laplacian(0.7, var) == rhs0.to(mesh.dtype.float).cuda())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        out.append((var().clone().cpu(), rep["itr"], rep["tol"]))
    return out


@pytest.mark.parametrize("name,n,dtype,faces", CASES, ids=[c[0] for c in CASES])
def test_placement_probe_changes_no_bit(name, n, dtype, faces, monkeypatch):
    monkeypatch.setenv("PYAPES_HIP_PLACE_MINBYTES", "0")
    g = torch.Generator().manual_seed(21)
    rhs0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if all(t == "periodic" for t, _ in faces):
        rhs0 -= rhs0.mean()
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64)      # a non-trivial start: the probe must hand it back intact
    a = _solve(n, dtype, faces, True, rhs0, x0)
    b = _solve(n, dtype, faces, False, rhs0, x0)
    for (xa, ia, ta), (xb, ib, tb) in zip(a, b):
        assert ia == ib and ta == tb
        assert torch.equal(xa, xb), float((xa - xb).abs().max())


def test_placement_probe_runs_and_reports():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch\nfrom test_gpu_place import _solve, CASES\n"
            "n = CASES[0][1]; g = torch.Generator().manual_seed(1)\n"
            "_solve(n, 'double', CASES[0][3], True, torch.randn((1, *n), generator=g, dtype=torch.float64), torch.zeros((1, *n), dtype=torch.float64))\n"
            % (root, os.path.join(root, "tests")))
    env = dict(os.environ, PYAPES_HIP_DEBUG="1", PYAPES_HIP_PLACE_MINBYTES="0",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle"), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    kept = [ln for ln in r.stderr.splitlines() if "placement probe: kept blocks" in ln]
    assert len(kept) == 1, r.stderr[-2000:]        # once: the second solve re-uses the choice
    assert sum("placement probe:" in ln for ln in r.stderr.splitlines()) >= 8      # blocks and offsets were tried
