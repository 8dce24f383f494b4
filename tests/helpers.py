"""Shared builders for the parity tests: the same case descriptors drive the oracle
(tests only) and the product path (pyapes_amd on the GPU through the C ABI)."""
from __future__ import annotations

import contextlib
import os

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

HOSTRING_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpa_hostring.so")


def hip_options(monkeypatch, **kw):
    """Options of every pa_ctx created from now on, in this process and in the processes it starts:
    PYAPES_HIP_OPTIONS="name=value,..." (pa_ctx_set_option names; None takes a name out again)."""
    cur = dict(item.split("=") for item in os.environ.get("PYAPES_HIP_OPTIONS", "").split(",") if item)
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(int(v))
    if cur:
        monkeypatch.setenv("PYAPES_HIP_OPTIONS", ",".join(f"{k}={v}" for k, v in cur.items()))
    else:
        monkeypatch.delenv("PYAPES_HIP_OPTIONS", raising=False)


def use_hostring():
    """Hand the test stand-in for librccl (tests/lib/pa_hostring.hip: ranks as processes that may share one GPU, host
    shared memory as the wire) to libpyapes_hip -- an explicit call in every rank process, before its first communicator."""
    from pyapes_amd.hip.lib import load_library
    assert os.path.exists(HOSTRING_LIB), f"{HOSTRING_LIB} missing: pyapes_amd/csrc/build.sh builds it"
    rc = load_library().pa_comm_use_impl(HOSTRING_LIB.encode())
    assert rc == 0, f"pa_comm_use_impl({HOSTRING_LIB}) -> {rc}"


def case_faces(case):
    return O.FACES_RZ if case.get("coord", "xyz") == "rz" else FACES


def oracle_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return O.poisson_cfg(nd)
    if case["bcs"] == "poisson_rz":
        return O.poisson_rz_cfg()
    if case["bcs"] == "robin":
        return O.robin_cfg(nd)
    faces = case_faces(case)
    return [{"bc_face": faces[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(case["bcs"])]


def product_cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return poisson_bcs(nd)
    if case["bcs"] == "poisson_rz":
        from pyapes_amd.testing.poisson import poisson_rz_bcs
        return poisson_rz_bcs()
    if case["bcs"] == "robin":   # callables that READ THE FIELD (the same functions drive reference, oracle and product)
        return [dict(d, bc_val_opt=None) for d in O.robin_cfg(nd)]
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


class SumTap:
    """Records the value of every ``torch.sum(t, dim=...)`` made while it is active -- the dot products of the
    reference's (and the oracle's) solver loops (linalg.py:119-137, 204-250), which is all they use it for.
    It wraps whatever ``torch.sum`` is at that moment, so it goes on LAST (outside a reordering ``fsum``)."""

    def __init__(self):
        self.vals = []

    def __enter__(self):
        self.prev = torch.sum
        prev, vals = self.prev, self.vals

        def tap(t, dim=None, **kw):
            if dim is None:
                return prev(t, **kw)
            r = prev(t, dim=dim, **kw)
            vals.append(float(r.reshape(-1)[0]))
            return r

        torch.sum = tap
        return self

    def __exit__(self, *exc):
        torch.sum = self.prev


def scalar_history(method, sums):
    """The per-iteration scalars the solver loops compute from their dot products and drop, rebuilt from the
    recorded sums: CG -> rows (alpha, beta) from (r.r, d.Ad, r.r, r'.r') per iteration (linalg.py:118-137);
    BiCGSTAB -> rows (alpha, omega, rho_next) from r0.r0, then (r0.v [, t.s, t.t, r0.t]) per iteration
    (linalg.py:204-250; the iteration that leaves through the first stop test has alpha only)."""
    def nn(v):
        return 0.0 if (v != v or v in (float("inf"), float("-inf"))) else v
    rows = []
    if method == "cg":
        for i in range(len(sums) // 4):
            rr, dad, rr2, rrn = sums[4 * i:4 * i + 4]
            rows.append((nn(rr / dad) if dad != 0 else 0.0, rrn / rr2 if rr2 != 0 else float("nan")))
        return np.array(rows, dtype=np.float64).reshape(-1, 2)
    rho, i = sums[0], 1
    while i < len(sums):
        alpha = nn(rho / sums[i]) if sums[i] != 0 else 0.0
        if i + 3 <= len(sums) - 1:
            ts, tt, r0t = sums[i + 1:i + 4]
            omega = nn(ts / tt) if tt != 0 else 0.0
            rho = -omega * r0t
            rows.append((alpha, omega, rho))
            i += 4
        else:
            rows.append((alpha, float("nan"), float("nan")))
            i += 1
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


def oracle_solve(case, rhs0, K, variant=0, tap=None):
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
            with (tap if tap is not None else contextlib.nullcontext()):
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


def _oracle_solve_random_order(case, rhs0, K, seed, tap=None):
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
            with (tap if tap is not None else contextlib.nullcontext()):
                return O.solve_poisson(oracle_mesh(case), oracle_cfg(case), torch.as_tensor(rhs0).clone(),
                                       method=case["method"], tol=case["tol"], max_it=K,
                                       coeff=case.get("coeff", 1.0), sign=case.get("sign", 1.0))
    finally:
        torch.sum = orig


def summation_hull(case, rhs0, K, x_ref, n_random=24, histories=None):
    """The REFERENCE ALGORITHM evaluated with 5 structured + ``n_random`` random summation orders of its
    dot products -> (band, diam, iteration counts): band = largest rel. distance of a sample from ``x_ref``
    (the reference's own result), diam = largest rel. distance between two samples.
    See test_solve_vs_reference for how they grade.  ``histories`` (a list) receives every sample's
    per-iteration scalars (``scalar_history``)."""
    xs, its = [], []
    for v in range(5):
        tap = SumTap() if histories is not None else None
        x, r = oracle_solve(case, rhs0, K, v, tap=tap)
        xs.append(x)
        its.append(r["itr"])
        if tap is not None:
            histories.append(scalar_history(case["method"], tap.vals))
    for s in range(n_random):
        tap = SumTap() if histories is not None else None
        x, r = _oracle_solve_random_order(case, rhs0, K, s, tap=tap)
        xs.append(x)
        its.append(r["itr"])
        if tap is not None:
            histories.append(scalar_history(case["method"], tap.vals))
    band = max(rel_err(x, x_ref) for x in xs)
    n = len(xs)
    diam = max(rel_err(xs[i], xs[j]) for i in range(n) for j in range(i + 1, n))
    return band, diam, its


def agreement_horizon(ref_hist, histories, rtol):
    """First iteration (0-based) at which some reordered run of the reference algorithm differs from the
    reference's recorded scalars by more than ``rtol`` (relative, any column); len(ref_hist) if none does."""
    n = len(ref_hist)
    for i in range(n):
        for h in histories:
            if i >= len(h):
                return i
            a, b = ref_hist[i], h[i]
            for u, v in zip(a, b):
                if u != u and v != v:
                    continue
                if not abs(u - v) <= rtol * max(abs(u), 1e-300):
                    return i
    return n


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
