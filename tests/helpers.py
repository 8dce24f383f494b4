"""Shared builders for the parity tests: the same case descriptors drive the oracle
(tests only) and the product path (pyapes_amd on the GPU through the C ABI)."""
from __future__ import annotations

import numpy as np
import torch

import pyapes_oracle as O
from pyapes_amd.geometry import Box, Cylinder
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_bcs
from pyapes_amd.variables import Field

FACES = O.FACES


def case_faces(case):
    return O.FACES_RZ if case.get("coord", "xyz") == "rz" else FACES


def oracle_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return O.poisson_cfg(nd)
    if case["bcs"] == "poisson_rz":
        return O.poisson_rz_cfg()
    faces = case_faces(case)
    return [{"bc_face": faces[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(case["bcs"])]


def product_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return poisson_bcs(nd)
    if case["bcs"] == "poisson_rz":
        from pyapes_amd.testing.poisson import poisson_rz_bcs
        return poisson_rz_bcs()
    faces = case_faces(case)
    return [{"bc_face": faces[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
            for i, (t, v) in enumerate(case["bcs"])]


def oracle_mesh(case):
    return O.OMesh(case["lower"], case["upper"], case["spacing"], case["dtype"], case.get("coord", "xyz"))


def product_mesh(case, device="cuda"):
    geo = Cylinder if case.get("coord", "xyz") == "rz" else Box
    return Mesh(geo(case["lower"], case["upper"]), None, case["spacing"], device, case["dtype"])


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


def oracle_solve(case, rhs0, K, variant=0):
    """The oracle on a golden case.  variant > 0 evaluates every torch.sum in a different, equally
    valid order (reversed along all / the first / the last summed axis, or as two half sums):
    same algorithm, same inputs, same arithmetic otherwise."""
    import warnings
    mesh = oracle_mesh(case)
    orig = torch.sum

    def fsum(t, dim=None, **kw):
        dims = tuple(range(t.dim())) if dim is None else tuple(dim)
        if variant == 4:
            if dim is None:
                f = t.contiguous().flatten()
                h = f.numel() // 2
                return orig(f[:h]) + orig(f[h:])
            f = t.contiguous().flatten(1)
            h = f.shape[1] // 2
            return orig(f[:, :h], dim=1) + orig(f[:, h:], dim=1)
        fd = {1: dims, 2: dims[:1], 3: dims[-1:]}[variant]
        tt = t.flip(fd).contiguous()
        return orig(tt) if dim is None else orig(tt, dim=dim)

    if variant:
        torch.sum = fsum
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return O.solve_poisson(mesh, oracle_cfg(case), torch.as_tensor(rhs0).clone(), method=case["method"],
                                   tol=case["tol"], max_it=K, coeff=case.get("coeff", 1.0),
                                   sign=case.get("sign", 1.0))
    finally:
        torch.sum = orig


def summation_band(case, rhs0, K):
    """(max rel. deviation of the result, list of iteration counts) of the REFERENCE ALGORITHM over
    five summation orders of its dot products.  BiCGSTAB and CG on the reference's non-symmetric
    periodic operator amplify 1e-16 perturbations; no implementation with a different
    reduction tree can be expected to sit closer to the reference than this band."""
    x0, r0 = oracle_solve(case, rhs0, K, 0)
    dev, its = 0.0, [r0["itr"]]
    for v in (1, 2, 3, 4):
        x, r = oracle_solve(case, rhs0, K, v)
        dev = max(dev, rel_err(x, x0))
        its.append(r["itr"])
    return dev, its


def _oracle_solve_random_order(case, rhs0, K, seed):
    """the oracle with every torch.sum evaluated over a random permutation of its addends, split into a
    random number of partial sums: one more equally valid summation order of the same algorithm"""
    import warnings
    orig = torch.sum
    gen = torch.Generator().manual_seed(1000 + seed)

    def fsum(t, dim=None, **kw):
        f = t.contiguous().flatten() if dim is None else t.contiguous().flatten(1)
        perm = torch.randperm(f.shape[-1], generator=gen)
        nb = int(torch.randint(2, 64, (1,), generator=gen))
        acc = None
        for c in f[..., perm].chunk(nb, dim=-1):
            s = orig(c, dim=-1)
            acc = s if acc is None else acc + s
        return acc

    torch.sum = fsum
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return O.solve_poisson(oracle_mesh(case), oracle_cfg(case), torch.as_tensor(rhs0).clone(),
                                   method=case["method"], tol=case["tol"], max_it=K,
                                   coeff=case.get("coeff", 1.0), sign=case.get("sign", 1.0))
    finally:
        torch.sum = orig


def summation_hull(case, rhs0, K, x_ref, n_random=24):
    """The REFERENCE ALGORITHM evaluated with 5 structured + ``n_random`` random summation orders of its
    dot products -> (band, diam, iteration counts): band = largest rel. distance of a sample from ``x_ref``
    (the reference's own result), diam = largest rel. distance between two samples.
    See test_solve_vs_reference for how they grade."""
    xs, its = [], []
    for v in range(5):
        x, r = oracle_solve(case, rhs0, K, v)
        xs.append(x)
        its.append(r["itr"])
    for s in range(n_random):
        x, r = _oracle_solve_random_order(case, rhs0, K, s)
        xs.append(x)
        its.append(r["itr"])
    band = max(rel_err(x, x_ref) for x in xs)
    n = len(xs)
    diam = max(rel_err(xs[i], xs[j]) for i in range(n) for j in range(i + 1, n))
    return band, diam, its


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
