"""-m gpu: the resident small-mesh solver (pa_resident.hip: the whole CG / Jacobi / BiCGSTAB solve in one
cooperative launch, fields in LDS) against the launch-per-phase loops it replaces and against the oracle.

The two paths run the same per-cell arithmetic; only the grouping of the global sums differs (per-box partials
summed in box order, against per-tile partial rows), so Jacobi iterates -- no global sum feeds back into them --
must be bit-identical and CG / BiCGSTAB iterates agree to rounding.  tests/test_gpu_parity_golden.py runs the
reference goldens through BOTH loops; this file pins the path itself: that it is taken, every box layout / face
order / dtype, periodic axes, the stop test, Field.VARo, a wait that gives up, and the fall-backs."""
import os
import random
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pyapes_amd.geometry import Box
from pyapes_amd.hip.context import context_for
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.ops import Solver
from pyapes_amd.variables import Field

FACES = ["xl", "xu", "yl", "yu", "zl", "zu"]


def _solve(monkeypatch, resident, n, bcs, dtype, method, rhs, x0, tol, max_it, order=None, save_old=False, coeff=0.8,
           env=None, adv=False):
    monkeypatch.setenv("PYAPES_HIP_RESIDENT", "1" if resident else "0")
    from helpers import hip_options
    hip_options(monkeypatch, **{k: (None if (env or {}).get(k) is None else int(env[k])) for k in
                                ("res_cells", "res_nt", "res_nt_cells", "res_spin")})
    nd = len(n)
    mesh = Mesh(Box([0.0] * nd, [1.0 + 0.1 * a for a in range(nd)]), None, list(n), "cuda", dtype)
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
    if order is not None:
        cfg = [cfg[i] for i in order]
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    var.set_var_tensor(x0.cuda().clone())
    c = {"method": method, "tol": tol, "max_it": max_it, "report": False}
    if save_old:
        c["save_old"] = True
    s = Solver({"fdm": c})
    if adv:   # a term list the lean (one Laplacian) build does not cover: pa_apply_terms on the box
        fdm = FDM({"div": {"limiter": "upwind", "edge": False}})
        s.set_eq(fdm.div(0.6, var) - fdm.laplacian(0.05, var) == rhs.cuda().clone())
    else:
        s.set_eq(-FDM().laplacian(coeff, var) == rhs.cuda().clone())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = s.solve()
    ctx = context_for(mesh)
    old = var.VARo.cpu() if save_old and rep["itr"] >= 1 else None
    return var().cpu(), rep, ctx.resident_used(), ctx.resident_plan(method), old


def _case(rng):
    nd = rng.choice([2, 3, 3])
    if nd == 2:
        n = [rng.choice([9, 16, 33, 64, 100, 129]), rng.choice([8, 17, 64, 65, 128, 130])]
    else:
        n = [rng.choice([5, 8, 12, 17, 24, 33]), rng.choice([6, 9, 16, 20, 40]), rng.choice([8, 17, 32, 33, 64, 66])]
    bcs = []
    for a in range(nd):
        if rng.random() < 0.25:   # a periodic axis: never cut, the box is its own neighbour there
            bcs += [("periodic", None), ("periodic", None)]
            continue
        for _ in range(2):
            t = rng.choice(["dirichlet", "dirichlet", "neumann", "symmetry"])
            bcs.append((t, None if t == "symmetry" else round(rng.uniform(-1, 1), 3)))
    if not any(t == "dirichlet" for t, _ in bcs):
        for a in range(nd):
            if bcs[2 * a][0] != "periodic":
                bcs[2 * a] = ("dirichlet", 0.25)
                break
        else:
            bcs[0], bcs[1] = ("dirichlet", 0.25), ("dirichlet", -0.5)
    dtype = "double" if rng.random() < 0.7 else "single"
    return n, bcs, dtype


def _fields(n, dtype, seed):
    tdt = torch.float64 if dtype == "double" else torch.float32
    g = torch.Generator().manual_seed(seed)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64).to(tdt)
    return rhs, x0


@pytest.mark.parametrize("method", ["cg", "jacobi", "bicgstab"])
def test_resident_matches_launch_per_phase(monkeypatch, method):
    """random extents / face types / face ORDER / dtypes; a dozen iterations (far from convergence, where a
    change of the summation grouping cannot yet have been amplified) and, every third case, a stop on the
    tolerance"""
    ncases = int(os.environ.get("PYAPES_FUZZ_CASES", "36"))
    rng = random.Random(11 + len(method) + ncases)
    taken = 0
    for case in range(ncases):
        n, bcs, dtype = _case(rng)
        rhs, x0 = _fields(n, dtype, 2000 + case)
        order = list(range(2 * len(n)))
        if case % 4 == 1:
            rng.shuffle(order)          # not the factory order: the resident fill is literal, face after face
        stop = case % 3 == 0
        tol = (1e-1 if dtype == "double" else 3e-1) if stop else -1.0
        max_it = 60 if stop else 11
        adv = method == "bicgstab" and case % 2 == 0
        if method == "bicgstab":
            # BiCGSTAB amplifies a change of the summation grouping within tens of iterations on the nearly singular
            # operators this fuzz draws (one dirichlet face): tests/test_gpu_parity_golden.py grades such runs against
            # the reference's own summation-order hull; here every case is a short fixed run
            tol, max_it = -1.0, 7
        xa, ra, ua, _, _ = _solve(monkeypatch, False, n, bcs, dtype, method, rhs, x0, tol, max_it, order, adv=adv)
        xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bcs, dtype, method, rhs, x0, tol, max_it, order, adv=adv)
        assert ua == 0
        assert ub == plan[0]
        if ub == 0:
            continue   # a mesh the plan does not take (an axis too short to cut ...): nothing to compare
        taken += 1
        assert int(np.prod(plan[1])) == ub
        assert ra["itr"] == rb["itr"], (case, n, bcs, dtype, ra, rb)
        scale = float(xa.abs().max())
        if method == "jacobi":
            assert torch.equal(xa, xb), (case, n, bcs, dtype, float((xa - xb).abs().max()))
        else:
            rtol = (1e-11 if dtype == "double" else 2e-4) * (50.0 if method == "bicgstab" else 1.0)
            assert float((xa - xb).abs().max()) <= rtol * scale, (case, n, bcs, dtype, float((xa - xb).abs().max()), scale)
        assert rb["tol"] == pytest.approx(ra["tol"], rel=(1e-9 if dtype == "double" else 1e-3) * (100.0 if method == "bicgstab" else 1.0))
        assert ra["converge"] == rb["converge"]
    assert taken >= ncases // 2


def test_resident_plan_and_layouts(monkeypatch):
    """box layouts: one workgroup, cuts along one / two / three axes, uneven boxes, the largest mesh the plan takes
    (64^3: boxes of 16 x 16 x 13 with a one-cell halo), and just above it (launch-per-phase loops)"""
    bc3 = [("dirichlet", 0.0), ("neumann", 0.3), ("symmetry", None), ("dirichlet", 1.0), ("neumann", -0.2), ("dirichlet", 0.5)]
    for n, expect_resident in (([9, 9], True), ([128, 128], True), ([7, 300], True), ([33, 33, 33], True), ([23, 5, 130], True),
                               ([64, 64, 64], True), ([96, 96, 96], False)):
        bcs = bc3[: 2 * len(n)]
        rhs, x0 = _fields(n, "double", 7)
        xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bcs, "double", "cg", rhs, x0, -1.0, 6)
        assert (ub > 0) == expect_resident, (n, ub, plan)
        xa, ra, ua, _, _ = _solve(monkeypatch, False, n, bcs, "double", "cg", rhs, x0, -1.0, 6)
        assert ra["itr"] == rb["itr"] == 7
        assert float((xa - xb).abs().max()) <= 1e-11 * float(xa.abs().max()), (n, plan)
    # forced layouts of one mesh: many small boxes / few large ones, 256 .. 1024 threads
    n = [24, 20, 28]
    rhs, x0 = _fields(n, "double", 8)
    xa, ra, _, _, _ = _solve(monkeypatch, False, n, bc3, "double", "cg", rhs, x0, -1.0, 9)
    seen = set()
    for env in ({"res_cells": 128}, {"res_cells": 4096}, {"res_nt": 256}, {"res_nt": 1024, "res_nt_cells": 1}):
        xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bc3, "double", "cg", rhs, x0, -1.0, 9, env=env)
        assert ub > 0
        seen.add(ub)
        assert float((xa - xb).abs().max()) <= 1e-11 * float(xa.abs().max()), (env, plan)
    assert len(seen) >= 2


def test_resident_vs_oracle(monkeypatch):
    """the resident path against the CPU oracle (the literal reference algorithm) on a mixed-BC mesh, per-node
    Dirichlet values and an inhomogeneous Neumann face"""
    import pyapes_oracle as O
    n = [18, 14, 22]
    mesh_o = O.OMesh([0.0, 0.0, 0.0], [1.0, 1.1, 1.2], n, "double", "xyz")
    bcs = [("dirichlet", 0.4), ("neumann", 0.25), ("symmetry", None), ("dirichlet", -0.3), ("neumann", 0.0), ("dirichlet", 1.0)]
    cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(bcs)]
    rhs, x0 = _fields(n, "double", 21)
    for method, K in (("cg", 8), ("jacobi", 8), ("bicgstab", 6)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xo, rep_o = O.solve_poisson(mesh_o, cfg, rhs.clone(), x0=x0.clone(), method=method, tol=-1.0, max_it=K - 1,
                                        coeff=0.8, sign=-1.0)[:2]
        xb, rb, ub, _, _ = _solve(monkeypatch, True, n, bcs, "double", method, rhs, x0, -1.0, K - 1)
        assert ub > 0
        assert rb["itr"] == rep_o["itr"]
        assert float((xb - xo).abs().max()) <= (1e-9 if method == "bicgstab" else 1e-11) * float(xo.abs().max())
        assert rb["tol"] == pytest.approx(rep_o["tol"], rel=1e-9)


def test_resident_edges(monkeypatch):
    """zero iterations (tol >= 1), Field.VARo, a coefficient-free and a negative-coefficient equation, fp32"""
    n = [20, 24]
    bcs = [("dirichlet", 0.1), ("neumann", 0.2), ("dirichlet", -0.4), ("symmetry", None)]
    rhs, x0 = _fields(n, "double", 3)
    xb, rb, ub, _, _ = _solve(monkeypatch, True, n, bcs, "double", "cg", rhs, x0, 1.5, 10)
    xa, ra, _, _, _ = _solve(monkeypatch, False, n, bcs, "double", "cg", rhs, x0, 1.5, 10)
    assert rb["itr"] == ra["itr"] == 0 and torch.equal(xa, xb)
    for method in ("cg", "jacobi", "bicgstab"):
        xb, rb, ub, _, ob = _solve(monkeypatch, True, n, bcs, "double", method, rhs, x0, -1.0, 5, save_old=True)
        xa, ra, _, _, oa = _solve(monkeypatch, False, n, bcs, "double", method, rhs, x0, -1.0, 5, save_old=True)
        assert ub > 0 and rb["itr"] == ra["itr"] == (5 if method == "bicgstab" else 6)
        assert float((oa - ob).abs().max()) <= 1e-10 * float(oa.abs().max())
        xp, rp, _, _, _ = _solve(monkeypatch, True, n, bcs, "double", method, rhs, x0, -1.0, 4)
        assert float((xp - ob).abs().max()) <= 1e-10 * float(xp.abs().max())   # VARo = the iterate one iteration earlier
    rhs32, x032 = _fields(n, "single", 4)
    xb, rb, ub, _, _ = _solve(monkeypatch, True, n, bcs, "single", "cg", rhs32, x032, -1.0, 7, coeff=-1.3)
    xa, ra, _, _, _ = _solve(monkeypatch, False, n, bcs, "single", "cg", rhs32, x032, -1.0, 7, coeff=-1.3)
    assert ub > 0 and float((xa - xb).abs().max()) <= 2e-4 * float(xa.abs().max())


def test_resident_periodic_axes(monkeypatch):
    """a periodic axis is never cut (its fill reads the far end): taken when the uncut boxes still fit (x-periodic
    16 x 20 x 24: boxes 16 x b x b; fully periodic 12^3: one box), launch-per-phase loops otherwise (fully periodic
    32^3) -- same results either way; non-factory order of the two periodic faces included"""
    per = ("periodic", None)
    for n, bcs, order, expect in (
            ([16, 20, 24], [per, per, ("dirichlet", 0.0), ("dirichlet", 0.0), ("neumann", 0.1), ("dirichlet", 0.2)], None, True),
            ([16, 20, 24], [per, per, ("dirichlet", 0.0), ("dirichlet", 0.0), ("neumann", 0.1), ("dirichlet", 0.2)], [1, 0, 2, 3, 4, 5], True),
            ([24, 40], [("dirichlet", 0.3), ("symmetry", None), per, per], None, True),
            ([12, 12, 12], [per] * 6, None, True),
            ([32, 32, 32], [per] * 6, None, False)):
        rhs, x0 = _fields(n, "double", 5)
        rhs = rhs - rhs.mean()
        for method in ("cg", "jacobi", "bicgstab"):
            xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bcs, "double", method, rhs, x0, -1.0, 5, order)
            assert (ub > 0) == expect and ub == plan[0], (n, method, ub, plan)
            xa, ra, _, _, _ = _solve(monkeypatch, False, n, bcs, "double", method, rhs, x0, -1.0, 5, order)
            assert ra["itr"] == rb["itr"]
            if ub == 0 or method == "jacobi":
                assert torch.equal(xa, xb), (n, method)
            else:
                assert float((xa - xb).abs().max()) <= 1e-9 * float(xa.abs().max()), (n, method, float((xa - xb).abs().max()))


def test_resident_falls_back(monkeypatch):
    """a one-sided periodic axis and meshes too large for 128 boxes run the launch-per-phase loops (resident_used == 0)"""
    n = [16, 20, 24]
    rhs, x0 = _fields(n, "double", 5)
    bcs = [("periodic", None), ("dirichlet", 0.3), ("dirichlet", 0.0), ("dirichlet", 0.0), ("neumann", 0.1), ("dirichlet", 0.2)]
    xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bcs, "double", "cg", rhs, x0, -1.0, 5)
    assert ub == 0 and plan[0] == 0
    xa, ra, _, _, _ = _solve(monkeypatch, False, n, bcs, "double", "cg", rhs, x0, -1.0, 5)
    assert torch.equal(xa, xb)


@pytest.mark.parametrize("method", ["cg", "jacobi", "bicgstab"])
def test_resident_wait_timeout_falls_back(monkeypatch, method):
    """a grid-wide wait that gives up (forced: spin bound 0) stores nothing; the host runs the launch-per-phase loop
    from the untouched state: same bits as with the resident path switched off"""
    n = [20, 18, 22]
    bcs = [("dirichlet", 0.0), ("neumann", 0.3), ("symmetry", None), ("dirichlet", 1.0), ("neumann", -0.2), ("dirichlet", 0.5)]
    rhs, x0 = _fields(n, "double", 31)
    xa, ra, ua, _, _ = _solve(monkeypatch, False, n, bcs, "double", method, rhs, x0, -1.0, 6)
    xb, rb, ub, plan, _ = _solve(monkeypatch, True, n, bcs, "double", method, rhs, x0, -1.0, 6, env={"res_spin": 0})
    assert plan[0] > 1 and ub == 0
    assert ra["itr"] == rb["itr"] and ra["tol"] == rb["tol"]
    assert torch.equal(xa, xb)


@pytest.mark.parametrize("method", ["cg", "jacobi", "bicgstab"])
def test_resident_nonfinite_stop_test_raises(monkeypatch, method):
    """an infinite right-hand side makes the stop-test value NaN / inf: RuntimeError("Invalid tolerance detected!")
    like linalg.py:334-336, from inside the one launch as from the launch-per-phase loops"""
    monkeypatch.setenv("PYAPES_HIP_RESIDENT", "1")
    mesh = Mesh(Box([0.0, 0.0], [1.0, 1.0]), None, [40, 36], "cuda", "double")
    cfg = [{"bc_face": FACES[i], "bc_type": "dirichlet", "bc_val": 0.0, "bc_val_opt": None} for i in range(4)]
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
    rhs = torch.full((1, 40, 36), float("inf"), dtype=torch.float64, device="cuda")
    s = Solver({"fdm": {"method": method, "tol": 1e-6, "max_it": 5, "report": False}})
    s.set_eq(FDM().laplacian(1.0, var) == rhs)
    assert context_for(mesh).resident_plan(method)[0] > 0
    with pytest.raises(RuntimeError, match="Invalid tolerance"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s.solve()
    assert context_for(mesh).resident_used() > 0


@pytest.mark.parametrize("method", ["cg", "jacobi", "bicgstab"])
@pytest.mark.parametrize("n", [[40, 36], [12, 16, 18]], ids=["2d", "3d"])
def test_resident_tensor_coefficient(monkeypatch, method, n):
    """Round 4: ``laplacian(Gamma(x), phi)`` with a tensor coefficient runs resident too -- the general-equation build reads
    Gamma from device memory through pa_apply_terms (and, for the Jacobi sweep, in diag(A)) while the solver's own fields
    stay in LDS.  Against the launch-per-phase loops (Jacobi bit for bit, CG / BiCGSTAB to rounding with identical counts)
    and the oracle."""
    import pyapes_oracle as O
    nd = len(n)
    bcs = [("dirichlet", 0.0), ("neumann", 0.3), ("symmetry", None), ("dirichlet", 1.0), ("neumann", -0.2), ("dirichlet", 0.5)][:2 * nd]
    g = torch.Generator().manual_seed(13)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    x0 = torch.randn((1, *n), generator=g, dtype=torch.float64)
    gamma = 1.0 + 0.2 * torch.rand((1, *n), generator=g, dtype=torch.float64)
    K = 6
    out = {}
    for resident in (True, False):
        monkeypatch.setenv("PYAPES_HIP_RESIDENT", "1" if resident else "0")
        mesh = Mesh(Box([0.0] * nd, [1.0 + 0.1 * a for a in range(nd)]), None, list(n), "cuda", "double")
        cfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None} for i, (t, v) in enumerate(bcs)]
        var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None})
        var.set_var_tensor(x0.cuda().clone())
        s = Solver({"fdm": {"method": method, "tol": -1.0, "max_it": K if method == "bicgstab" else K - 1, "report": False}})
        s.set_eq(-FDM().laplacian(gamma.cuda(), var) == rhs.cuda().clone())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rep = s.solve()
        used = context_for(mesh).resident_used()
        assert (used > 0) == resident, (resident, used)
        out[resident] = (var().cpu(), rep)
    (xa, ra), (xb, rb) = out[True], out[False]
    assert ra["itr"] == rb["itr"] == K
    if method == "jacobi":
        assert torch.equal(xa, xb)
    assert float((xa - xb).abs().max()) <= 1e-11 * float(xb.abs().max())
    om = O.OMesh([0.0] * nd, [1.0 + 0.1 * a for a in range(nd)], list(n), "double")
    ocfg = [{"bc_face": FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(bcs)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(om, ocfg, rhs.clone(), x0=x0.clone(), method=method, tol=-1.0,
                                 max_it=K if method == "bicgstab" else K - 1, coeff=gamma, sign=-1.0)
    assert ro["itr"] == K
    assert float(torch.linalg.norm(xa - xo) / torch.linalg.norm(xo)) <= 1e-10
