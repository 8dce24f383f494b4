"""-m gpu: the per-axis BC pair kernels of the CG loop (k_bc_pair: lower + upper face of one axis and
their share of the stop-test sum in one launch) against the face-by-face fill + k_shell path
(option bc_path bit 1, the literal reference order): the iterates must agree bit for bit for
every combination of face types, the stop-test value to summation order."""
import itertools
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
AXIS_CHOICES = [
    (("periodic", None), ("periodic", None)),
    (("dirichlet", 0.25), ("neumann", 0.3)),
    (("neumann", -0.2), ("symmetry", None)),
    (("symmetry", None), ("dirichlet", 1.0)),
    (("neumann", 0.0), ("neumann", 0.5)),
]


def _solve(bcs, n, dtype, K, paired, monkeypatch, ndim=3):
    from helpers import hip_options
    hip_options(monkeypatch, bc_path=1 if paired else 3)   # never the closed form: the pair path is what runs where it does not
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    box = Box[0:1, 0:1, 0:0.5] if ndim == 3 else Box[0:1, 0:0.7]
    mesh = Mesh(box, None, list(n), "cuda", dtype)
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    g = torch.Generator().manual_seed(11)
    var.set_var_tensor(torch.randn((1, *n), generator=g, dtype=torch.float64).to(mesh.dtype.float).cuda())
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(mesh.dtype.float).cuda()
    s = Solver({"fdm": {"method": "cg", "tol": -1.0, "max_it": K - 1, "report": False}})
    s.set_eq(FDM().laplacian(1.0, var) == rhs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    return var().cpu(), rep


@pytest.mark.parametrize("dtype", ["double", "single"])
def test_pair_equals_face_by_face_3d(dtype, monkeypatch):
    bad = []
    for ax, ay, az in itertools.product(AXIS_CHOICES, repeat=3):
        bcs = [ax[0], ax[1], ay[0], ay[1], az[0], az[1]]
        a, ra = _solve(bcs, (9, 7, 12), dtype, 5, True, monkeypatch)
        b, rb = _solve(bcs, (9, 7, 12), dtype, 5, False, monkeypatch)
        tol_ok = abs(ra["tol"] - rb["tol"]) <= 1e-5 * abs(rb["tol"]) if dtype == "single" else \
            abs(ra["tol"] - rb["tol"]) <= 1e-12 * abs(rb["tol"])
        if not (torch.equal(a, b) and tol_ok and ra["itr"] == rb["itr"]):
            bad.append(([t for t, _ in bcs], float((a - b).abs().max()), ra["tol"], rb["tol"]))
    assert not bad, bad[:5]


@pytest.mark.parametrize("n", [(40, 70, 260), (17, 33, 129)], ids=["40x70x260", "17x33x129"])
def test_pair_equals_face_by_face_tiled_sizes(n, monkeypatch):
    for bcs in ([("periodic", None)] * 6,
                [("dirichlet", 0.0), ("neumann", 0.5), ("symmetry", None), ("neumann", 0.0), ("dirichlet", 1.0),
                 ("neumann", -0.25)]):
        a, ra = _solve(bcs, n, "double", 6, True, monkeypatch)
        b, rb = _solve(bcs, n, "double", 6, False, monkeypatch)
        assert torch.equal(a, b)
        assert abs(ra["tol"] - rb["tol"]) <= 1e-12 * abs(rb["tol"])


def test_pair_equals_face_by_face_2d(monkeypatch):
    for ax, ay in itertools.product(AXIS_CHOICES, repeat=2):
        bcs = [ax[0], ax[1], ay[0], ay[1]]
        a, ra = _solve(bcs, (14, 19), "double", 5, True, monkeypatch, ndim=2)
        b, rb = _solve(bcs, (14, 19), "double", 5, False, monkeypatch, ndim=2)
        assert torch.equal(a, b), [t for t, _ in bcs]
        assert abs(ra["tol"] - rb["tol"]) <= 1e-12 * abs(rb["tol"])


def test_stop_iteration_count_unchanged(monkeypatch):
    """to convergence: same iteration count and iterate with and without the pair kernels"""
    bcs = [("dirichlet", 0.0), ("dirichlet", 0.5), ("neumann", 0.0), ("dirichlet", 0.0), ("symmetry", None),
           ("dirichlet", 1.0)]
    out = []
    from helpers import hip_options
    for paired in (True, False):
        hip_options(monkeypatch, bc_path=1 if paired else 3)
        cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
        mesh = Mesh(Box[0:1, 0:1, 0:1], None, [21, 23, 25], "cuda", "double")
        var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
        rhs = torch.sin(3 * mesh.X) * torch.cos(2 * mesh.Y) + mesh.Z
        s = Solver({"fdm": {"method": "cg", "tol": 1e-8, "max_it": 2000, "report": False}})
        s.set_eq(FDM().laplacian(1.0, var) == rhs.unsqueeze(0).contiguous())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        out.append((var().cpu(), rep))
    assert out[0][1]["itr"] == out[1][1]["itr"] and out[0][1]["converge"]
    assert torch.equal(out[0][0], out[1][0])
