"""Shared builders for the parity tests: the same case descriptors drive the oracle
(tests only) and the product path (pyapes_amd on the GPU through the C ABI)."""
from __future__ import annotations

import numpy as np
import torch

import pyapes_oracle as O
from pyapes_amd.geometry import Box
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_bcs
from pyapes_amd.variables import Field

FACES = O.FACES


def oracle_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return O.poisson_cfg(nd)
    return [{"bc_face": FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(case["bcs"])]


def product_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return poisson_bcs(nd)
    return [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
            for i, (t, v) in enumerate(case["bcs"])]


def oracle_mesh(case):
    return O.OMesh(case["lower"], case["upper"], case["spacing"], case["dtype"])


def product_mesh(case, device="cuda"):
    return Mesh(Box(case["lower"], case["upper"]), None, case["spacing"], device, case["dtype"])


def product_field(case, mesh, x0=None):
    var = Field("p", 1, mesh, {"domain": product_cfg(case), "obstacle": None})
    if x0 is not None:
        var.set_var_tensor(torch.as_tensor(x0).to(mesh.device).clone())
    return var


def product_solve(case, rhs0, K, method=None, x0=None):
    """rhs0: numpy/tensor (1,*nx).  Returns (x cpu tensor, report, solver)."""
    mesh = product_mesh(case)
    var = product_field(case, mesh, x0)
    rhs = torch.as_tensor(rhs0).to(mesh.device).clone()
    solver = Solver({"fdm": {"method": method or case["method"], "tol": case["tol"], "max_it": K,
                             "report": False}})
    fdm = FDM()
    coeff, sign = case.get("coeff", 1.0), case.get("sign", 1.0)
    eq = fdm.laplacian(coeff, var) if sign > 0 else -fdm.laplacian(coeff, var)
    solver.set_eq(eq == rhs)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = solver.solve()
    return var().cpu(), rep, solver


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float(torch.linalg.norm(a - b)) / max(float(torch.linalg.norm(b)), 1e-300)


def bit_equal(a, b):
    return torch.equal(torch.as_tensor(a).cpu(), torch.as_tensor(b).cpu())


def oracle_solve(case, rhs0, K, flip_sums=False):
    """The oracle on a golden case; flip_sums=True evaluates every torch.sum over the reversed
    tensor -- same algorithm, same inputs, a different (equally valid) summation order."""
    import warnings
    mesh = oracle_mesh(case)
    orig = torch.sum

    def fsum(t, dim=None, **kw):
        if dim is None:
            return orig(t.flip(tuple(range(t.dim()))).contiguous())
        return orig(t.flip(tuple(dim)).contiguous(), dim=dim)

    if flip_sums:
        torch.sum = fsum
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return O.solve_poisson(mesh, oracle_cfg(case), torch.as_tensor(rhs0).clone(), method=case["method"],
                                   tol=case["tol"], max_it=K, coeff=case.get("coeff", 1.0),
                                   sign=case.get("sign", 1.0))
    finally:
        torch.sum = orig


def summation_sensitivity(case, rhs0, K):
    """(rel. change of the result, change of the iteration count) of the REFERENCE ALGORITHM when only
    the order of its dot-product summations changes.  BiCGSTAB and CG on the reference's
    non-symmetric periodic operator amplify 1e-16 perturbations; no implementation
    with a different reduction tree can be closer to the reference than this."""
    x0, r0 = oracle_solve(case, rhs0, K, False)
    x1, r1 = oracle_solve(case, rhs0, K, True)
    return rel_err(x1, x0), abs(r1["itr"] - r0["itr"])


def true_residual(case, rhs0, x):
    """|| (b_adj - A x) ||_2 over the interior set, evaluated by the product's own operator."""
    from pyapes_amd.mesh.tools import boundary_slicer
    mesh = product_mesh(case)
    var = product_field(case, mesh, x)
    rhs = torch.as_tensor(rhs0).to(mesh.device).clone()
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 1, "report": False}})
    fdm = FDM()
    coeff, sign = case.get("coeff", 1.0), case.get("sign", 1.0)
    eq = fdm.laplacian(coeff, var) if sign > 0 else -fdm.laplacian(coeff, var)
    solver.set_eq(eq == rhs)
    S = tuple(boundary_slicer(mesh.dim, var.bcs))
    r = (solver.rhs - solver.Aop(var))[0][S]
    return float(torch.linalg.norm(r))
