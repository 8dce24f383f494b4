"""-m gpu: the last-block epilogues (pa_epilogue.h: the final reduction of d.Ad / r.r / the stop-test sum
and the scalar logic run in the last workgroup of phase A, phase B or the last BC pair kernel) against
the separate k_cg_post_* launches (the default; the epilogues are opt-in, PYAPES_HIP_EPILOGUE=1, because
the device-scope fences they need cost more than the launches they save on an 8-XCD part): same
summation tree, so iterates, iteration counts and the reported tolerance must agree bit for bit."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

FACES = ["xl", "xu", "yl", "yu", "zl", "zu"]
D = lambda v=0.0: ("dirichlet", v)   # noqa: E731
N = lambda v=0.0: ("neumann", v)     # noqa: E731
PE = ("periodic", None)
BCS = {
    "static": [D(0.0), D(0.5), D(0.0), D(0.0), D(1.0), D(0.0)],       # tail = phase B
    "mixed": [D(0.0), N(0.5), D(0.3), N(0.0), D(1.0), N(-0.25)],       # tail = last BC kernel
    "periodic": [PE] * 6,
}


def _solve(n, bcs, env, monkeypatch, tol, max_it, dtype="double"):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nd = len(n)
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs[:2 * nd])]
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, list(n), "cuda", dtype)
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    g = torch.Generator().manual_seed(4)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(mesh.dtype.float)
    if bcs[0][0] == "periodic":
        rhs -= rhs.mean()
    s = Solver({"fdm": {"method": "cg", "tol": tol, "max_it": max_it, "report": False}})
    s.set_eq(FDM().laplacian(1.0, var) == rhs.cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep


@pytest.mark.parametrize("fast", ["1", "0"], ids=["tiled", "generic"])
@pytest.mark.parametrize("bc_path", ["fused", "pair"])
@pytest.mark.parametrize("bc", list(BCS))
@pytest.mark.parametrize("n", [(20, 18, 132), (40, 132)], ids=["3d", "2d"])
def test_epilogue_equals_post_kernels(n, bc, bc_path, fast, monkeypatch):
    base = {"PYAPES_HIP_FASTPATH": fast}
    if bc_path == "pair":
        base["PYAPES_HIP_BC_UNFUSED"] = "1"
    for tol, max_it in ((-1.0, 7), (1e-9, 400)):
        a, ra = _solve(n, BCS[bc], dict(base, PYAPES_HIP_EPILOGUE="1"), monkeypatch, tol, max_it)
        b, rb = _solve(n, BCS[bc], dict(base, PYAPES_HIP_EPILOGUE="0"), monkeypatch, tol, max_it)
        assert ra["itr"] == rb["itr"] and ra["converge"] == rb["converge"], (ra, rb)
        assert ra["tol"] == rb["tol"], (ra, rb)
        assert torch.equal(a, b)


def test_epilogue_single_precision_and_large_grid(monkeypatch):
    for n, dtype in (((64, 64, 256), "single"), ((96, 128, 256), "double")):
        a, ra = _solve(n, BCS["static"], {"PYAPES_HIP_EPILOGUE": "1"}, monkeypatch, -1.0, 12, dtype)
        b, rb = _solve(n, BCS["static"], {"PYAPES_HIP_EPILOGUE": "0"}, monkeypatch, -1.0, 12, dtype)
        assert torch.equal(a, b) and ra["tol"] == rb["tol"] and ra["itr"] == rb["itr"] == 13


def test_graph_replay_equals_plain_enqueue(monkeypatch):
    """PYAPES_HIP_GRAPH=1 (opt-in: the loop is bound by the GPU-side dependent-dispatch latency, not by
    the host's launch rate, so the replay measures no gain): pa_cg_iterate replays a captured pair of
    iterations as a hipGraph -- same kernels, same order, so the iterate is bit-identical."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, warnings, torch
sys.path.insert(0, %r)
warnings.simplefilter("ignore")
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.hip import lib as L
from pyapes_amd.hip.context import context_for
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import mixed_bcs
mesh = Mesh(Box[0:1, 0:1, 0:1], None, [24, 20, 132], "cuda", "double")
var = Field("p", 1, mesh, {"domain": mixed_bcs([0.0, 0.5, 0.3, 0.0, 1.0, -0.25], ["dirichlet", "neumann"] * 3), "obstacle": None})
rhs = torch.randn((1, 24, 20, 132), generator=torch.Generator().manual_seed(3), dtype=torch.float64).cuda()
ctx = context_for(mesh)
ctx.bind_bcs(var(), var.bcs, 0)
ctx.set_terms([{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}])
ctx.rhs_adjust(rhs[0])
ctx.cg_begin(var()[0], rhs[0], -1.0, 100)
ctx.cg_iterate(11)          # odd count: 5 replayed pairs + 1 plain iteration
rep = ctx.cg_end()
torch.save({"x": var().cpu(), "itr": int(rep.itr), "tol": float(rep.tol)}, sys.argv[1])
""" % ROOT
    import os
    res = {}
    for g in ("1", "0"):
        out = f"/tmp/pa_graph_{g}_{os.getpid()}.pt"
        env = dict(os.environ, PYAPES_HIP_GRAPH=g)
        p = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        res[g] = torch.load(out)
        os.remove(out)
    assert res["1"]["itr"] == res["0"]["itr"] == 11 and res["1"]["tol"] == res["0"]["tol"]
    assert torch.equal(res["1"]["x"], res["0"]["x"])
