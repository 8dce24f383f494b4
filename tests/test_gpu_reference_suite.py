"""-m gpu: the reference's own passing tests, restated against pyapes_amd with the same problems,
assertions and tolerances (tests/test_solver.py, tests/test_variables.py::test_cylinder_field_bcs,
tests/test_spatial.py) -- what a pyapes user sees when switching the import and the device string.
Bit-level parity is pinned elsewhere (goldens); here the point is the user-visible contract."""
import warnings
from math import cos, cosh, exp, pi, sin

import pytest
import torch
from torch.testing import assert_close

from conftest import golden_load

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box, Cylinder
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdc import DiffFlux, hessian, jacobian
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_bcs, poisson_exact_nd, poisson_rhs_nd
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import CylinderBoundary, homogeneous_bcs, mixed_bcs

DEV = "cuda"


def _solver(method, tol, max_it=1000):
    return Solver({"fdm": {"method": method, "tol": tol, "max_it": max_it, "report": False}})


@pytest.mark.parametrize("domain, spacing, dim", [(Box[0:1], [11], 1), (Box[0:1, 0:1], [0.01, 0.01], 2),
                                                   (Box[0:1, 0:1, 0:1], [0.1, 0.1, 0.1], 3)],
                         ids=["1d", "2d", "3d"])
def test_poisson_nd_pure_dirichlet(domain, spacing, dim):
    """test_solver.py:30-88: CG then BiCGSTAB from zero, both converge to the analytic solution"""
    mesh = Mesh(domain, None, spacing, DEV)
    var = Field("p", 1, mesh, {"domain": poisson_bcs(dim), "obstacle": None})
    rhs = poisson_rhs_nd(mesh, var)
    exact = poisson_exact_nd(mesh)
    fdm = FDM()
    for method in ("cg", "bicgstab"):
        solver = _solver(method, 1e-6)
        solver.set_eq(fdm.laplacian(1.0, var) == rhs)
        solver.solve()
        assert solver.report["converge"] == True   # noqa: E712
        assert_close(var()[0], exact, rtol=0.1, atol=0.01)
        var = var.zeros_like()


def test_heat_conduction_2d_mixed():
    """test_solver.py:91-151: Laplace equation, Neumann / Dirichlet mix, against the reference's CSV"""
    mesh = Mesh(Box[0:1, 0:1], None, [11, 11], DEV)
    f_bc = mixed_bcs([0.0, 0.0, 0.0, 1.0], ["neumann", "dirichlet", "neumann", "dirichlet"])
    var = Field("p", 1, mesh, {"domain": f_bc, "obstacle": None}, init_val=0.0)
    solver = _solver("bicgstab", 1e-8)
    solver.set_eq(FDM().laplacian(var) == 0.0)
    solver.solve()
    ref = torch.from_numpy(golden_load("ref_heat_10x10")["sol"])
    assert_close(var()[0][:-1, :-1].cpu(), ref, atol=0.01, rtol=0.01)
    series = torch.zeros_like(mesh.X)
    for i in range(1, 201):
        lam = (2 * i - 1) * pi / 2
        series += 2 * (-1) ** (i - 1) / (lam * cosh(lam)) * torch.cosh(lam * mesh.Y) * torch.cos(lam * mesh.X)
    assert float((var()[0] - series)[:-1, :-1].abs().max()) < 0.01   # analytic series, corner node aside


def test_poisson_2d_mixed_periodic_and_neumann_run():
    """test_solver.py:164-207, 271-306: no assertion in the reference beyond running to completion"""
    mesh = Mesh(Box[0:1, 0:1], None, [41, 41], DEV)
    f_bc = mixed_bcs([None, None, 0, 0], ["periodic", "periodic", "dirichlet", "dirichlet"])
    var = Field("p", 1, mesh, {"domain": f_bc, "obstacle": None}, init_val=0.0)
    rhs = torch.zeros_like(var())
    rhs[0] = mesh.X * torch.sin(5.0 * pi * mesh.Y) + torch.exp(-((mesh.X - 0.5) ** 2 + (mesh.Y - 0.5) ** 2) / 0.02)
    solver = _solver("bicgstab", 1e-8)
    solver.set_eq(-FDM().laplacian(1.0, var) == rhs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = solver.solve()
    assert rep["itr"] > 0 and bool(torch.isfinite(var()).all())
    mesh = Mesh(Box[0:0.5, 0:0.5], None, [33, 33], DEV)
    f_bc = mixed_bcs([0, 0, 0, 0], ["dirichlet", "neumann", "dirichlet", "neumann"])
    var = Field("p", 1, mesh, {"domain": f_bc, "obstacle": None}, init_val=0.0)
    solver = _solver("cg", 1e-8)
    solver.set_eq(FDM().laplacian(1.0, var) == torch.ones_like(var()))
    rep = solver.solve()
    assert rep["converge"] and bool(torch.isfinite(var()).all())


def test_poisson_1d_mixed_neumann():
    """test_solver.py:210-268: phi'' = cos(pi x / 2 + pi / 4), phi'(-pi/2) = 1/4, phi(pi/4) = -1/2"""
    mesh = Mesh(Box[-pi / 2: pi / 4], None, [101], DEV)
    f_bc = mixed_bcs([-1 / 4, -1 / 2], ["neumann", "dirichlet"])
    var = Field("phi", 1, mesh, {"domain": f_bc, "obstacle": None}, init_val=0.0)
    rhs = torch.zeros_like(var())
    rhs[0] = torch.cos(pi / 2 * mesh.X + pi / 4)
    exact = ((1 / 4 - 2 / pi * sin(-(pi ** 2) / 4 + pi / 4)) * (mesh.X - pi / 4)
             - (4 / pi ** 2) * (torch.cos(pi / 2 * mesh.X + pi / 4) - cos(pi ** 2 / 8 + pi / 4)) - 1 / 2)
    solver = _solver("bicgstab", 1e-6)
    solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    solver.solve()
    one_sided = lambda t: (-3 / 2 * t[0] + 2 * t[1] - 1 / 2 * t[2]) / mesh.dx[0]   # noqa: E731
    assert_close(one_sided(var()[0]), one_sided(exact), atol=1e-1, rtol=1e-1)
    assert_close(var()[0], exact, atol=1e-3, rtol=1e-3)


def test_advection_diffusion_1d():
    """test_solver.py:361-390: u' - eps u'' = 1 on [0, 1], u(0) = u(1) = 0"""
    mesh = Mesh(Box[0:1], None, [0.05], DEV)
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(1, 0.0, "dirichlet"), "obstacle": None}, init_val=0.5)
    eps = 0.5
    exact = mesh.X - (torch.exp(-(1 - mesh.X) / eps) - exp(-1 / eps)) / (1 - exp(-1 / eps))
    solver = _solver("bicgstab", 1e-5)
    fdm = FDM()
    solver.set_eq(fdm.grad(var) - fdm.laplacian(eps, var) == 1.0)
    solver.solve()
    assert_close(var()[0], exact, rtol=0.1, atol=0.01)


def test_cylinder_field_bcs():
    """test_variables.py:132-188: BC fill on an rz mesh incl. a callable value and bc_val_opt"""
    mesh = Mesh(Cylinder[0:1, 0:2], None, [5, 5], DEV)
    ru_bc = lambda grid, mask, *_: grid[1][mask] * 4.4   # noqa: E731
    f_bc = CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0}, ru={"bc_type": "dirichlet", "bc_val": ru_bc},
                            zl={"bc_type": "neumann", "bc_val": 1.3}, zu={"bc_type": "dirichlet", "bc_val": 0.44})
    var = Field("d", 1, mesh, {"domain": f_bc(), "obstacle": None}, init_val="random")
    for bc in var.bcs:
        bc.apply(var(), mesh.grid, 0)
    v = var()[0]
    assert_close(v[-1, 1:-1], 4.4 * mesh.grid[1][0][1:-1])
    assert_close(v[1:-1, -1], 0.44 * torch.ones_like(v[1:-1, -1]))
    assert_close(v[0, 1:-1], 4 / 3 * v[1, 1:-1] - 1 / 3 * v[2, 1:-1])
    assert_close(v[1:-1, 0], 4 / 3 * v[1:-1, 1] - 1 / 3 * v[1:-1, 2] + 2 / 3 * 1.3 * mesh.dx[1])

    def zu_bc(grid, mask, _, opt):
        return grid[0][mask] * torch.sum(opt["T"])

    f_bc = CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0}, ru={"bc_type": "dirichlet", "bc_val": ru_bc},
                            zl={"bc_type": "neumann", "bc_val": 1.3},
                            zu={"bc_type": "dirichlet", "bc_val": zu_bc, "bc_val_opt": {"T": torch.ones_like(v)}})
    var = Field("d", 1, mesh, {"domain": f_bc(), "obstacle": None}, init_val="random")
    for bc in var.bcs:
        bc.apply(var(), mesh.grid, 0)
    assert_close(var()[0][1:-1, -1], mesh.grid[0][1:-1, -1] * var()[0].numel())


def test_diff_flux():
    """test_spatial.py:16-48"""
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [3, 3, 3], DEV)
    var = Field("test", 1, mesh, {"domain": None, "obstacle": None})
    var.set_var_tensor(mesh.grid[0] ** 2 + 2 * mesh.grid[2] ** 2)
    grad = torch.gradient(var()[0].cpu(), spacing=mesh.dx.tolist(), edge_order=2)
    hess = hessian(var)
    flux = DiffFlux()(hess, var)
    assert_close(flux[0].cpu(), hess.xx.cpu() * grad[0] + hess.xy.cpu() * grad[1] + hess.xz.cpu() * grad[2])
    mesh = Mesh(Cylinder[0:1, 0:1], None, [3, 3], DEV)
    var = Field("test", 1, mesh, {"domain": None, "obstacle": None})
    var.set_var_tensor(mesh.grid[0] ** 2)
    grad = [t.to(DEV) for t in torch.gradient(var()[0].cpu(), spacing=mesh.dx.tolist(), edge_order=2)]
    hess = hessian(var)
    flux = DiffFlux()(hess, var)
    assert_close(flux[0], mesh.grid[0] * hess.rr * grad[0] + mesh.grid[0] * hess.rz * grad[1])
    assert_close(flux[1], hess.rz * grad[0] + hess.zz * grad[1])


def test_jac_and_hess_and_containers():
    """test_spatial.py:51-109"""
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [3, 3, 3], DEV)
    var = Field("test", 1, mesh, {"domain": None, "obstacle": None})
    var.set_var_tensor(mesh.grid[0] ** 2 + 2 * mesh.grid[2] ** 2)
    jac = jacobian(var)
    assert_close(jac.x, 2 * mesh.grid[0])
    assert_close(jac.y, torch.zeros_like(var()[0]))
    assert_close(jac.z, 4 * mesh.grid[2])
    var.set_var_tensor((mesh.grid[0] ** 2) * (mesh.grid[2] ** 2))
    hess = hessian(var)
    assert_close(hess.xx, 2 * mesh.grid[2] ** 2)
    assert_close(hess.xy, torch.zeros_like(var()[0]))
    assert_close(hess.xz, 4 * mesh.grid[0] * mesh.grid[2])
    mesh = Mesh(Box[0:1, 0:1], None, [3, 3], DEV)
    var = Field("test", 1, mesh, {"domain": None, "obstacle": None})
    var.set_var_tensor(mesh.grid[0] ** 2)
    jac, hess = jacobian(var), hessian(var)
    assert_close(hess.xy, hess["yx"])
    with pytest.raises(KeyError):
        jac["z"]
    with pytest.raises(KeyError):
        hess["zz"]
    assert len(jac) == 2 and len(hess) == 3 and [t.shape for t in jac] == [var()[0].shape] * 2
