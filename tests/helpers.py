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
