"""-m gpu: the computational cells of the reference's three demo notebooks (demos/poisson_equations/
pure_dirichlet.ipynb, axisymmetric.ipynb, demos/advection_diffusion/ss_advection_diffusion.ipynb),
CONDENSED: their own import lines (``pyapes.core.*``) and call sequences, plotting dropped, the BC
lambdas inlined, asserts on the numbers the notebooks print added.  The imports resolve to this package
through ``pyapes_amd.install_as_pyapes()``; the one edit a user makes is the device string.  Runs in a
child process so the aliases do not leak.  (Not the .ipynb files themselves: those cannot travel.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DEMOS = r'''
import sys, warnings
sys.path.insert(0, %r)
warnings.filterwarnings("ignore")
import pyapes_amd
pyapes_amd.install_as_pyapes()

# ---- pure_dirichlet.ipynb -------------------------------------------------------------------------
from pyapes.core.geometry import Box
from pyapes.core.mesh import Mesh
from pyapes.core.solver.fdm import FDM
from pyapes.core.solver.ops import Solver
from pyapes.core.variables import Field
from pyapes.testing.poisson import poisson_bcs, poisson_rhs_nd, poisson_exact_nd
import torch

mesh = Mesh(Box[0:1, 0:1], None, [100, 100], "cuda")
var = Field("p", 1, mesh, {"domain": poisson_bcs(2), "obstacle": None})
rhs = poisson_rhs_nd(mesh, var)
sol_ex = poisson_exact_nd(mesh)
solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 1000, "report": True}})
fdm = FDM()
solver.set_eq(fdm.laplacian(1.0, var) == rhs)
rep = solver.solve()
assert rep["itr"] == 210 and rep["converge"], rep            # the notebook's recorded output: 210 iterations
assert abs(rep["tol"] - 9.661285603057063e-07) < 1e-12, rep
assert float((var()[0] - sol_ex).abs().max()) < 1e-3

# ---- axisymmetric.ipynb ------------------------------------------------------------------------------
from pyapes.core.geometry import Cylinder
from pyapes.core.variables.bcs import CylinderBoundary

mesh = Mesh(Cylinder[0:1, 0:2], None, [64, 64], "cuda")
f_bc = CylinderBoundary(
    rl={"bc_type": "neumann", "bc_val": 0.0},
    ru={"bc_type": "dirichlet", "bc_val": lambda grid, mask, *_: torch.zeros_like(grid[0][mask])},
    zl={"bc_type": "dirichlet", "bc_val": lambda grid, mask, *_: 1 - grid[0][mask] ** 2},
    zu={"bc_type": "dirichlet", "bc_val": lambda grid, mask, *_: torch.exp(-2.0 * grid[1][mask]) * (1 - grid[0][mask])},
)
var = Field("p", 1, mesh, {"domain": f_bc(), "obstacle": None})
solver = Solver({"fdm": {"method": "bicgstab", "tol": 1e-7, "max_it": 1000, "report": True}})
rhs = torch.zeros_like(var())
rhs[0] = -4.0 * mesh.R ** 2 * torch.exp(-2.0 * mesh.Z)
solver.set_eq(FDM().laplacian(1.0, var) == rhs)
rep = solver.solve()
assert rep["converge"] and bool(torch.isfinite(var()).all()), rep
print("axisymmetric: itr", rep["itr"], "tol", rep["tol"], "acc", float(torch.linalg.norm(torch.exp(-2.0 * mesh.Z) * (1 - mesh.R ** 2) - var()[0])))
# the notebook's recorded output: 195 iterations, tol 8.149016007661279e-08.  BiCGSTAB on this problem is summation-order
# sensitive: the REFERENCE ALGORITHM itself (the oracle, tests/tools/axisymmetric_demo_band.py) needs 201 iterations
# with torch.sum as it is and 200 ... 244 under 40 random orders of its dot products -- the band, with a margin
assert 175 <= rep["itr"] <= 260 and rep["tol"] <= 1e-7, rep

# ---- ss_advection_diffusion.ipynb ----------------------------------------------------------------------
from math import exp
from pyapes.core.variables.bcs import homogeneous_bcs

mesh = Mesh(Box[0:1], None, [0.02], "cuda")
recorded = {1: 51, 0.5: 49, 0.2: 50, 0.1: 54, 0.02: 52}     # the notebook's iteration counts
for eps in [1, 0.5, 0.2, 0.1, 0.02]:
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(1, 0.0, "dirichlet"), "obstacle": None})
    solver = Solver({"fdm": {"method": "bicgstab", "tol": 1e-5, "max_it": 1000, "report": True}})
    fdm = FDM()
    solver.set_eq(fdm.grad(var) - fdm.laplacian(eps, var) == 1.0)
    rep = solver.solve()
    exact = mesh.X - (torch.exp(-(1 - mesh.X) / eps) - exp(-1 / eps)) / (1 - exp(-1 / eps))
    err = float((var()[0] - exact).abs().max())
    print("adv-diff eps", eps, "itr", rep["itr"], "err", err)
    assert rep["converge"] and err < (0.05 if eps >= 0.1 else 0.2), (eps, rep, err)
    assert abs(rep["itr"] - recorded[eps]) <= 3, (eps, rep)
print("demos ok")
'''


def test_demo_notebooks_run_unmodified_but_for_the_device():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", DEMOS % ROOT], capture_output=True, text=True, timeout=600, env=env)
    print(out.stdout)
    assert out.returncode == 0 and "demos ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
