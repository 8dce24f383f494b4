"""-m gpu: the library-side slab loop (pa_cg_iterate_comm, csrc/pa_comm.hip) with MORE THAN ONE rank.

RCCL refuses two ranks on one device and the test box has one GPU, so every rank process hands the library the stand-in
of tests/lib/pa_hostring.hip (an explicit pa_comm_use_impl() call: helpers.use_hostring): the ranks are PROCESSES that share
cuda:0, each with its own ctx / streams / events; everything in pa_comm.hip runs as it does over RCCL (grouped
send / recv between distinct peers and, on a 2-rank periodic ring, twice to the same peer; out-of-place row
all-reduces with uneven slabs; the cross-stream event pair of the second communicator; k_slab_mid) -- only the wire
(host shared memory, stream-ordered through pinned staging) differs.  Every case is compared with the stepwise
torch.distributed driver (gloo; the path tests/test_gpu_slab.py pins) on the same ranks, and with the
single-domain oracle.

What "equal" can mean.  stepwise-in-library adds the same per-rank sums as the stepwise driver, so with two ranks
(a + b is commutative) the iterates are bit-identical.  The folded sequence all-reduces the per-workgroup partial
ROWS and every rank then adds the rows: sum over rows of sums over ranks instead of the reverse -- the same numbers
added in another order, so those runs are held to 1e-12 against the stepwise driver (and, like everything, to
1e-10 against the oracle) with identical iteration counts.
"""
import os
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyapes_oracle as O
from test_slab_gloo import CASES, _free_port, spawn_ranks

pytestmark = pytest.mark.gpu

TERMS = [{"kind": 0, "sign": -1.0, "coeff": 0.7}]
MODES = {
    "stepwise_driver": {"PYAPES_HIP_COMM": "0"},                       # the reference sequence (torch.distributed, gloo)
    "folded": {},                                                     # default for N > 1: rows all-reduced, exchange on
                                                                      # the second communicator + stream
    "folded_one_stream": {"PYAPES_HIP_COMM_OVERLAP": "0"},            # exchange on the ctx stream
    "stepwise_in_library": {"PYAPES_HIP_SLAB_FOLD": "0"},             # round 1's sequence inside the C loop
}
ENV_KEYS = ("PYAPES_HIP_FASTPATH", "PYAPES_HIP_COMM", "PYAPES_HIP_COMM_OVERLAP", "PYAPES_HIP_SLAB_FOLD", "PYAPES_HIP_HOSTRING_FAIL",
            "PYAPES_HIP_COMM_TIMEOUT", "PYAPES_HIP_HOSTRING_TIMEOUT", "PYAPES_HIP_PLACE")


def _worker(rank, world, port, cases, out):
    """cases: list of (key, BC mix, n, K, dtype, runs); runs: list of (label, env dict, generic_rank or None).  Everything is
    executed one after the other by the SAME rank processes in ONE process group (a process start costs more than a
    solve), each run on a fresh mesh / ctx / communicator."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    warnings.filterwarnings("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from helpers import use_hostring
    use_hostring()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyapes_amd.geometry import Box
        from pyapes_amd.hip.context import context_for
        from pyapes_amd.mesh import Mesh
        from pyapes_amd.slab import SlabCG
        from pyapes_amd.variables import Field
        torch.cuda.set_device(0)
        allres = {}
        for key, name, n, K, dtype, runs in cases:
            bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
                   for i, (t, v) in enumerate(CASES[name])]
            g = torch.Generator().manual_seed(7)
            rhs_g = torch.randn((1, *n), generator=g, dtype=torch.float64)
            if name == "per":
                rhs_g -= rhs_g.mean()
            res = {}
            for label, env, generic_rank in runs:
                for k in ENV_KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype, slab=(rank, world))
                ctx = context_for(mesh)
                if generic_rank == rank:
                    ctx.set_option("fastpath", False)     # this rank runs the generic kernels: it cannot fold
                var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
                rhs = rhs_g.to(mesh.dtype.float)[:, mesh.i_off:mesh.i_off + mesh.nx[0]].contiguous().cuda()
                drv = SlabCG(mesh, var, rhs, TERMS, dist)
                info = {"lib_comm": bool(drv.lib_comm), "impl": ctx.comm_impl(),
                        "overlap": bool(drv.lib_comm and ctx.comm_overlap()), "err": getattr(drv, "lib_comm_error", None)}
                rep = drv.solve(1e-30, K, poll=3)
                info["folded"] = bool(drv.folded)
                if drv.lib_comm:      # release the communicators while every rank is alive
                    torch.cuda.synchronize()
                    ctx.comm_destroy()
                    ctx.comm_ready = None
                parts, infos = [None] * world, [None] * world
                dist.all_gather_object(parts, var().cpu())
                dist.all_gather_object(infos, info)
                res[label] = {"x": torch.cat(parts, dim=1), "itr": int(rep.itr), "tol": float(rep.tol), "ranks": infos}
            allres[key] = res
        if rank == 0:
            torch.save(allres, out)
    finally:
        dist.destroy_process_group()


def _spawn(world, name, n, K, dtype, runs, tmp_path):
    out = str(tmp_path / "res.pt")
    spawn_ranks(_worker, lambda port: (world, port, [("only", name, n, K, dtype, runs)], out), world)
    return torch.load(out)["only"]


N2, N4, K_IT = (24, 20, 132), (26, 20, 132), 6
CANNOT_FOLD = [("stepwise_driver", MODES["stepwise_driver"], 1), ("library", {}, 1)]
THREE = [(m, MODES[m], None) for m in ("stepwise_driver", "folded", "stepwise_in_library")]
PLACE_RUNS = [("folded", {"PYAPES_HIP_PLACE": "0"}, None), ("folded_search", {"PYAPES_HIP_PLACE": "2"}, None),
              ("stepwise", {"PYAPES_HIP_SLAB_FOLD": "0", "PYAPES_HIP_PLACE": "0"}, None),
              ("stepwise_search", {"PYAPES_HIP_SLAB_FOLD": "0", "PYAPES_HIP_PLACE": "2"}, None)]


@pytest.fixture(scope="module")
def two_ranks(tmp_path_factory):
    """ONE pair of rank processes runs every fault-free two-rank case of the SlabCG tests below."""
    cases = [(f"modes-{name}", name, N2, K_IT, "double", [(m, MODES[m], None) for m in MODES])
             for name in ("per", "xper", "mix", "dir")]
    cases += [("fp32", "xper", (16, 12, 136), K_IT, "single", THREE),
              ("cannot_fold", "xper", N2, K_IT, "double", CANNOT_FOLD),
              ("place", "xper", N2, 40, "double", PLACE_RUNS)]
    out = str(tmp_path_factory.mktemp("two_ranks") / "res.pt")
    spawn_ranks(_worker, lambda port: (2, port, cases, out), 2)
    return torch.load(out)


@pytest.fixture(scope="module")
def four_ranks(tmp_path_factory):
    cases = [(f"uneven-{name}", name, N4, K_IT, "double", THREE) for name in ("per", "xper", "mix")]
    cases += [("cannot_fold", "xper", N4, K_IT, "double", CANNOT_FOLD)]
    out = str(tmp_path_factory.mktemp("four_ranks") / "res.pt")
    spawn_ranks(_worker, lambda port: (4, port, cases, out), 4)
    return torch.load(out)


def _oracle(name, n, K, dtype):
    mesh = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(CASES[name])]
    g = torch.Generator().manual_seed(7)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if name == "per":
        rhs -= rhs.mean()
    rhs = rhs.to(mesh.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return O.solve_poisson(mesh, cfg, rhs, method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)


def _rel(a, b):
    return float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))


def _check_against_oracle(res, name, n, K, dtype):
    xo, ro = _oracle(name, n, K, dtype)
    for label, r in res.items():
        assert r["itr"] == ro["itr"] == K + 1, (label, r["itr"])
        err = _rel(r["x"], xo)
        assert err < (1e-10 if dtype == "double" else 1e-5), (label, err)


@pytest.mark.parametrize("name", ["per", "xper", "mix", "dir"])
def test_two_ranks_every_library_mode(name, two_ranks):
    """P = 2.  On the periodic ring ("per", "xper") both neighbours of a rank are the SAME peer: two sends and two
    receives per group to one rank, paired in program order (pa_comm.hip exchange())."""
    if name not in CASES:
        pytest.skip(name)
    n, K = N2, K_IT
    res = two_ranks[f"modes-{name}"]
    ref = res["stepwise_driver"]
    assert all(not r["lib_comm"] for r in ref["ranks"])
    for m in ("folded", "folded_one_stream", "stepwise_in_library"):
        r = res[m]
        assert all(k["lib_comm"] and "hostring" in k["impl"] for k in r["ranks"]), r["ranks"]
        assert all(k["folded"] == (m != "stepwise_in_library") for k in r["ranks"]), r["ranks"]
        assert all(k["overlap"] == (m != "folded_one_stream") for k in r["ranks"]), r["ranks"]
    # same per-rank sums, two ranks: bit for bit
    assert torch.equal(res["stepwise_in_library"]["x"], ref["x"]) and res["stepwise_in_library"]["tol"] == ref["tol"]
    # folded: rows summed after the ranks instead of before; the two stream layouts run the same arithmetic
    assert torch.equal(res["folded"]["x"], res["folded_one_stream"]["x"]) and res["folded"]["tol"] == res["folded_one_stream"]["tol"]
    assert _rel(res["folded"]["x"], ref["x"]) < 1e-12
    assert abs(res["folded"]["tol"] - ref["tol"]) <= 1e-9 * abs(ref["tol"])
    _check_against_oracle(res, name, n, K, "double")


@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_four_ranks_uneven_slabs(name, four_ranks):
    """P = 4, 26 planes = 7 + 7 + 6 + 6: interior ranks with two DISTINCT neighbours and no global x face, a ring
    longer than its end ranks, and per-rank grids of different size -- ranks 2 and 3 write fewer partial rows than
    the agreed counts, so their row all-reduces run out of place (rows they never write stay zero)."""
    n, K = N4, K_IT
    res = four_ranks[f"uneven-{name}"]
    ref = res["stepwise_driver"]
    for m in ("folded", "stepwise_in_library"):
        r = res[m]
        assert all(k["lib_comm"] and k["overlap"] and k["folded"] == (m == "folded") for k in r["ranks"]), r["ranks"]
        # four ranks: gloo's all-reduce adds the ranks in its own order, the stand-in in rank order
        assert _rel(r["x"], ref["x"]) < 1e-12, m
        assert abs(r["tol"] - ref["tol"]) <= 1e-9 * abs(ref["tol"])
    _check_against_oracle(res, name, n, K, "double")


def test_fp32_two_ranks(two_ranks):
    n, K = (16, 12, 136), K_IT
    res = two_ranks["fp32"]
    assert torch.equal(res["stepwise_in_library"]["x"], res["stepwise_driver"]["x"])
    assert all(k["lib_comm"] and k["folded"] for k in res["folded"]["ranks"])
    assert _rel(res["folded"]["x"], res["stepwise_driver"]["x"]) < 1e-5
    _check_against_oracle(res, "xper", n, K, "single")


@pytest.mark.parametrize("world,n", [(2, (24, 20, 132)), (4, (26, 20, 132))])
def test_one_rank_cannot_fold_all_stay_stepwise(world, n, request):
    """Rank 1 runs the generic kernels (no partial rows to fold): the ranks must agree to stay on the stepwise
    sequence INSIDE the library -- a rank folding alone would all-reduce rows against its peers' sums."""
    K = K_IT
    res = request.getfixturevalue("two_ranks" if world == 2 else "four_ranks")["cannot_fold"]
    assert all(k["lib_comm"] and not k["folded"] for k in res["library"]["ranks"]), res["library"]["ranks"]
    if world == 2:
        assert torch.equal(res["library"]["x"], res["stepwise_driver"]["x"])
    else:
        assert _rel(res["library"]["x"], res["stepwise_driver"]["x"]) < 1e-12
    _check_against_oracle(res, "xper", n, K, "double")


@pytest.mark.parametrize("fail,expect", [
    ("1:init:1", "no_second_communicator"),     # rank 1 cannot create the exchange communicator
    ("1:corrupt:0", "stepwise_driver"),          # rank 1 fails the collective self-test (wrong sum)
    ("1:init:0", "stepwise_driver"),             # rank 1 cannot create a communicator at all
])
def test_a_rank_failing_set_up_takes_every_rank_to_the_same_fallback(fail, expect, tmp_path):
    """pa_comm_init / pa_comm_selftest failing on ONE rank: the agreements (MIN all-reduce of an ok flag inside
    pa_comm_init and in SlabCG._setup_lib_comm) must put every rank on the same path -- without the second
    communicator, or on the stepwise torch.distributed driver -- and the solve must still be right."""
    n, K = (24, 20, 132), 6
    env = {"PYAPES_HIP_HOSTRING_FAIL": fail, "PYAPES_HIP_COMM_TIMEOUT": "6", "PYAPES_HIP_HOSTRING_TIMEOUT": "12"}
    res = _spawn(2, "per", n, K, "double", [("stepwise_driver", MODES["stepwise_driver"], None), ("faulty", env, None)], tmp_path)
    ranks = res["faulty"]["ranks"]
    if expect == "no_second_communicator":
        assert all(k["lib_comm"] and k["folded"] and not k["overlap"] for k in ranks), ranks
        assert _rel(res["faulty"]["x"], res["stepwise_driver"]["x"]) < 1e-12
    else:
        assert all(not k["lib_comm"] and not k["folded"] for k in ranks), ranks
        assert ranks[1]["err"], "rank 1 should have recorded why it left the library-side path"
        assert torch.equal(res["faulty"]["x"], res["stepwise_driver"]["x"])
    _check_against_oracle(res, "per", n, K, "double")


# ---- Solver.set_eq() / solve() on slab meshes: the reference's own surface on 2 and 4 ranks ---------------------------
def _worker_solver(rank, world, port, cases, out):
    """cases: list of (key, BC mix, n, jobs, dtype); jobs: list of (label, method, tol, K[, env]) solved one after the other
    through ``Solver`` on ``Mesh(..., slab=...)`` by the same rank processes; env: variables set while that job's mesh /
    context is created (PYAPES_HIP_COMM=0: stepwise torch.distributed driver)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    warnings.filterwarnings("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from helpers import use_hostring
    use_hostring()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyapes_amd.geometry import Box
        from pyapes_amd.hip.context import context_for
        from pyapes_amd.mesh import Mesh
        from pyapes_amd.solver.fdm import FDM
        from pyapes_amd.solver.ops import Solver
        from pyapes_amd.variables import Field
        torch.cuda.set_device(0)
        allres = {}
        for key, name, n, jobs, dtype in cases:
            bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
                   for i, (t, v) in enumerate(CASES[name])]
            g = torch.Generator().manual_seed(7)
            rhs_g = torch.randn((1, *n), generator=g, dtype=torch.float64)
            if name == "per":
                rhs_g -= rhs_g.mean()
            res = {}
            for job in jobs:
                label, method, tol, K = job[:4]
                for k in ENV_KEYS:
                    os.environ.pop(k, None)
                os.environ.update(job[4] if len(job) > 4 else {})
                mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype, slab=(rank, world))
                var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
                if method == "euler":     # the explicit march (solver/march.py -> SlabEuler), K steps
                    from pyapes_amd.solver.march import euler_march
                    from test_slab_gloo import EULER_DT, EULER_NU, EULER_U, euler_start
                    var.set_var_tensor(euler_start(name, n, dtype)[:, mesh.i_off:mesh.i_off + mesh.nx[0]].contiguous().cuda())
                    euler_march(var, EULER_U, EULER_NU, EULER_DT, K)
                    parts = [None] * world
                    dist.all_gather_object(parts, var().cpu())
                    res[label] = {"x": torch.cat(parts, dim=1)}
                    continue
                rhs = rhs_g.to(mesh.dtype.float)[:, mesh.i_off:mesh.i_off + mesh.nx[0]].contiguous().cuda()
                cfg = {"method": method, "tol": tol, "max_it": K, "report": False}
                if method == "jacobi":
                    cfg["omega"] = 0.9
                solver = Solver({"fdm": cfg})
                solver.set_eq(-FDM().laplacian(0.7, var) == rhs)
                rep = solver.solve()
                ctx = context_for(mesh)
                in_lib = bool(getattr(ctx, "comm_ready", None))
                if getattr(ctx, "comm_ready", None):      # release the communicators while every rank is alive
                    torch.cuda.synchronize()
                    ctx.comm_destroy()
                    ctx.comm_ready = None
                parts = [None] * world
                dist.all_gather_object(parts, var().cpu())
                res[label] = {"x": torch.cat(parts, dim=1), "itr": int(rep["itr"]), "tol": float(rep["tol"]),
                              "converge": bool(rep["converge"]), "in_lib": in_lib}
            allres[key] = res
        if rank == 0:
            torch.save(allres, out)
    finally:
        dist.destroy_process_group()


SOLVER_JOBS = [("cg", "cg", 1e-30, 6), ("bicgstab", "bicgstab", 1e-30, 6),
               ("bicgstab_stepwise", "bicgstab", 1e-30, 6, {"PYAPES_HIP_COMM": "0"}),
               # Jacobi (SlabJacobi): 7 sweeps end in the context's field (copied back), 8 in the caller's
               ("jacobi", "jacobi", 1e-30, 6), ("jacobi_even", "jacobi", 1e-30, 7),
               ("jacobi_stepwise", "jacobi", 1e-30, 6, {"PYAPES_HIP_COMM": "0"})]
EULER_JOBS = [("euler", "euler", 0.0, 5), ("euler_generic", "euler", 0.0, 5, {"PYAPES_HIP_FASTPATH": "0"})]
CONV_JOBS = [("bicgstab", "bicgstab", 1e-8, 1500), ("cg", "cg", 1e-8, 30)]
FP32_JOBS = [("cg", "cg", 1e-30, 6), ("bicgstab", "bicgstab", 1e-30, 6), ("jacobi", "jacobi", 1e-30, 6)]


@pytest.fixture(scope="module")
def solver_two_ranks(tmp_path_factory):
    cases = [(f"solve-{name}", name, N2, SOLVER_JOBS, "double") for name in ("per", "xper", "mix")]
    cases += [("converge", "xper", (24, 20, 36), CONV_JOBS, "double"), ("fp32", "mix", (16, 12, 136), FP32_JOBS, "single")]
    cases += [(f"euler-{name}", name, N2, EULER_JOBS, "double") for name in ("per", "xper", "mix", "neu_hi")]
    cases += [("euler-fp32", "mix", (16, 12, 136), EULER_JOBS[:1], "single")]
    out = str(tmp_path_factory.mktemp("solver_two") / "res.pt")
    spawn_ranks(_worker_solver, lambda port: (2, port, cases, out), 2)
    return torch.load(out)


@pytest.fixture(scope="module")
def solver_four_ranks(tmp_path_factory):
    cases = [(f"solve-{name}", name, N4, SOLVER_JOBS, "double") for name in ("per", "xper", "mix")]
    cases += [(f"euler-{name}", name, N4, EULER_JOBS, "double") for name in ("xper", "mix")]
    out = str(tmp_path_factory.mktemp("solver_four") / "res.pt")
    spawn_ranks(_worker_solver, lambda port: (4, port, cases, out), 4)
    return torch.load(out)


def _oracle_any(name, n, method, tol, K, dtype="double"):
    mesh = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(CASES[name])]
    g = torch.Generator().manual_seed(7)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if name == "per":
        rhs -= rhs.mean()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = {"omega": 0.9} if method == "jacobi" else {}
        return O.solve_poisson(mesh, cfg, rhs.to(mesh.dtype), method=method, tol=tol, max_it=K, coeff=0.7, sign=-1.0, **kw)


@pytest.mark.parametrize("world,n", [(2, (24, 20, 132)), (4, (26, 20, 132))], ids=["2ranks", "4ranks_uneven"])
@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_solver_solve_on_slab_meshes(name, world, n, request):
    """``Solver.solve()`` with ``Mesh(..., slab=(rank, world))`` on every rank: linalg.solve hands CG to SlabCG (the
    library-side loop over the stand-in wire), Jacobi to SlabJacobi and BiCGSTAB to SlabBiCGSTAB (planes of v' and r, three small all-reduces
    per iteration: inside the library, pa_bicg_iterate_comm, and with torch.distributed between the step calls).
    Identical iteration counts and <= 1e-10 against the single-domain oracle; fully periodic, x-periodic (the ring
    across the ranks) and mixed faces; even and uneven slabs."""
    jobs = SOLVER_JOBS
    res = request.getfixturevalue("solver_two_ranks" if world == 2 else "solver_four_ranks")[f"solve-{name}"]
    assert res["cg"]["in_lib"] and res["bicgstab"]["in_lib"] and not res["bicgstab_stepwise"]["in_lib"]
    assert res["jacobi"]["in_lib"] and not res["jacobi_stepwise"]["in_lib"]
    # the same step calls, the sums added by the stand-in in rank order / by gloo: two ranks -> the same bits
    for m in ("bicgstab", "jacobi"):
        if world == 2:
            assert torch.equal(res[m]["x"], res[m + "_stepwise"]["x"])
        else:
            assert _rel(res[m]["x"], res[m + "_stepwise"]["x"]) < 1e-12
    for label, method, tol, K in [j[:4] for j in jobs]:
        xo, ro = _oracle_any(name, n, method, tol, K)
        r = res[label]
        assert r["itr"] == ro["itr"] == (K if method == "bicgstab" else K + 1), (label, r["itr"], ro["itr"])
        assert r["converge"] == ro["converge"]
        assert _rel(r["x"], xo) < 1e-10, (label, _rel(r["x"], xo))
        assert abs(r["tol"] - ro["tol"]) <= 1e-7 * abs(ro["tol"]), (label, r["tol"], ro["tol"])


def test_slab_bicgstab_converges_on_the_periodic_problem_cg_cannot_finish(solver_two_ranks):
    """The 3-D form of the reference's tests/test_solver.py:164-207 (x periodic, the other faces Dirichlet, ``-laplacian ==
    rhs``, BiCGSTAB): on two ranks -- the periodic axis is the ring across them -- BiCGSTAB meets its stop test, where CG
    with a periodic face runs to max_it.  The count of such a run is summation-order sensitive; the
    bar is convergence, the stop-test value and the solution."""
    n = (24, 20, 36)
    bc_name = "xper"
    res = solver_two_ranks["converge"]
    xo, ro = _oracle_any(bc_name, n, "bicgstab", 1e-8, 1500)
    b = res["bicgstab"]
    assert b["converge"] and ro["converge"] and b["tol"] <= 1e-8
    assert 0.5 * ro["itr"] <= b["itr"] <= 2 * ro["itr"], (b["itr"], ro["itr"])
    # (how close two converged runs of this problem sit: the reference algorithm itself, with nothing changed but the order
    # of its torch.sum, spreads by 3e-2 on the 2-D golden case of the same kind -- tests/golden/hulls.npz bicg2d_xper_f64;
    # the periodic BC fill keeps editing nodes of the interior set)
    assert _rel(b["x"], xo) < 5e-3, _rel(b["x"], xo)
    assert not res["cg"]["converge"] and res["cg"]["itr"] == 31        # K + 1 iterations, stop test never met


def test_solver_solve_on_a_slab_fp32(solver_two_ranks):
    jobs = FP32_JOBS
    n = (16, 12, 136)
    res = solver_two_ranks["fp32"]
    for label, method, tol, K in jobs:
        xo, ro = _oracle_any("mix", n, method, tol, K, "single")
        assert res[label]["itr"] == ro["itr"]
        assert _rel(res[label]["x"], xo) < 1e-5, (label, _rel(res[label]["x"], xo))


def test_placement_search_inside_the_library_side_slab_loop(two_ranks):
    """The online placement search (csrc/pa_place.hip) ticks inside pa_cg_iterate_comm too -- every rank searches for
    itself while the row all-reduces and the plane exchange go on.  Forced onto these small slabs without a budget
    (PYAPES_HIP_PLACE=2), where its timings are noise and roles move at random, it must change no bit of the folded
    solve on two ranks -- r moves by having phase B write the new residual elsewhere AFTER the mid kernel has formed the
    send planes from the old one -- and the stepwise-in-library sequence likewise."""
    K = 40
    res = two_ranks["place"]
    assert all(k["lib_comm"] and k["folded"] for k in res["folded_search"]["ranks"])
    assert torch.equal(res["folded"]["x"], res["folded_search"]["x"]) and res["folded"]["tol"] == res["folded_search"]["tol"]
    assert torch.equal(res["stepwise"]["x"], res["stepwise_search"]["x"]) and res["stepwise"]["tol"] == res["stepwise_search"]["tol"]
    assert res["folded"]["itr"] == res["folded_search"]["itr"] == K + 1


@pytest.mark.parametrize("world,name,dtype", [(2, "per", "double"), (2, "xper", "double"), (2, "mix", "double"),
                                              (2, "neu_hi", "double"), (2, "fp32", "single"), (4, "xper", "double"),
                                              (4, "mix", "double")])
def test_euler_march_on_slab_meshes(world, name, dtype, request):
    """``euler_march`` on ``Mesh(..., slab=(rank, world))`` (SlabEuler): five steps (the result ends in the second
    ping-pong buffer) on 2 and 4 ranks, with the marching step kernel and with the generic one (the same bits), against
    the single-domain oracle; a periodic ring (the far planes of the NEW field cross it before the fill), uneven slabs."""
    from test_slab_gloo import euler_oracle
    res = request.getfixturevalue("solver_two_ranks" if world == 2 else "solver_four_ranks")[f"euler-{name}"]
    bc, n = ("mix", (16, 12, 136)) if name == "fp32" else (name, N2 if world == 2 else N4)
    xo = euler_oracle(bc, n, 5, dtype)
    assert _rel(res["euler"]["x"], xo) < (1e-12 if dtype == "double" else 1e-5), _rel(res["euler"]["x"], xo)
    if "euler_generic" in res:
        assert torch.equal(res["euler"]["x"], res["euler_generic"]["x"])


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 3 at its full size on slabs ("configs 3 and 5 only in their 1-GPU form")


def _worker_config3(rank, world, port, n, K, method, outdir, wl="c3"):
    """Solver.set_eq() / solve() on Mesh(..., slab=(rank, world)) at 512^3 fp64, fully periodic: the default N > 1 sequence
    (library-side loop, folded iterations, plane exchange on the second communicator) between rank PROCESSES sharing cuda:0."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    warnings.filterwarnings("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    import json
    os.environ.update(json.loads(os.environ.get("PA_TEST_SLAB_ENV", "{}")))     # (profiles/tools/slab_bicg_diag.py: other modes)
    from helpers import use_hostring
    use_hostring()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from pyapes_amd.geometry import Box
        from pyapes_amd.hip.context import context_for
        from pyapes_amd.mesh import Mesh
        from pyapes_amd.solver.fdm import FDM
        from pyapes_amd.solver.ops import Solver
        from pyapes_amd.variables import Field
        torch.cuda.set_device(0)
        _, _, dtype, kind, upper, _ = bench.WORKLOADS[wl]
        mesh = Mesh(Box([0.0] * 3, list(upper)), None, list(n), "cuda", dtype, slab=(rank, world))
        var = Field("p", 1, mesh, {"domain": bench.make_bcs(kind), "obstacle": None})
        rhs = bench.synth_rhs(n, mesh.i_off, mesh.nx[0], os.environ.get("PA_TEST_SLAB_RHS_KIND", kind), mesh.dtype.float,
                              mesh.device)
        if method == "euler":       # config 4's march: the Gaussian of bench.py, K steps in one call
            from pyapes_amd.solver.march import euler_march
            var.set_var_tensor(bench.gaussian(mesh.X, mesh.Y, mesh.Z).to(mesh.dtype.float).unsqueeze(0).contiguous())
            var.apply_bcs()
            nu, dts = bench.euler_params(mesh.dx_list[0])
            euler_march(var, 1.0, nu, dts, K)
            rep = {"itr": K, "tol": 0.0}
        else:
            cfg = {"method": method, "tol": 1e-30, "max_it": K, "report": False}
            if method == "jacobi":
                cfg["omega"] = 0.9
            solver = Solver({"fdm": cfg})
            solver.set_eq(FDM().laplacian(1.0, var) == rhs)
            rep = solver.solve()
        ctx = context_for(mesh)
        in_lib = bool(getattr(ctx, "comm_ready", None))
        impl = ctx.comm_impl() if in_lib else None
        torch.cuda.synchronize()
        if in_lib:      # release the communicators while every rank is alive
            ctx.comm_destroy()
            ctx.comm_ready = None
        torch.save({"x": var().cpu(), "i_off": int(mesh.i_off), "itr": int(rep["itr"]), "tol": float(rep["tol"]),
                    "in_lib": in_lib, "impl": impl}, os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wl,method,K,rhs_kind,bar", [("c3", "cg", 8, "periodic", 1e-10), ("c3", "bicgstab", 6, "mixed", 1e-12),
                                                      ("c5", "cg", 8, "mixed", 1e-5), ("c3", "jacobi", 6, "periodic", 1e-14),
                                                      ("c4", "euler", 10, "neusym", 0.0), ("c2", "cg", 8, "dirichlet", 1e-10)])
def test_baseline_configs_full_size_on_two_and_four_slabs(wl, method, K, rhs_kind, bar, tmp_path, monkeypatch):
    """(BASELINE config 5 likewise: 1024 x 1024 x 512 fp32, Dirichlet / Neumann faces, 2 x 512 and 4 x 256 planes, fp32 bars; Jacobi
    on config 3's mesh; config 4's explicit march, 256^3 fp32, ten steps in one call: no sum crosses the ranks, so the same bits.)
    512^3 fp64, fully periodic (the mesh BASELINE's metric is quoted on), 2 x 256 and 4 x 128 planes, through
    Solver.set_eq() / solve(), against the SAME solve on the whole mesh on one GPU: identical iteration counts, iterate and
    stop-test value within the bar (the slabs add the same products in another order).  The one-GPU solve at this size is
    what test_gpu_properties.py pins (eigen-solution, null space, fast == generic); the CPU oracle needs ~10 s per iteration
    here.  CG runs bench.py's own right-hand side.  BiCGSTAB runs the right-hand side with bench.synth_rhs's pseudo-random
    term: on the periodic one (a sum of Fourier modes) |r| drops seven orders in two iterations, BiCGSTAB's scalars are then
    quotients of nearly cancelled sums and the ORDER of the sums alone moves the iterate by 2e-11 ... 7e-11 and the residual
    norm by a factor with four ranks, while on a generic right-hand side 2, 3 and 4 ranks sit within 6e-16 of the whole mesh
    (profiles/r04_slab_bicg_diag.txt, made by profiles/tools/slab_bicg_diag.py)."""
    import bench
    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.solver.fdm import FDM
    from pyapes_amd.solver.ops import Solver
    from pyapes_amd.variables import Field
    _, n, dtype, kind, upper, _ = bench.WORKLOADS[wl]
    tol_bar = 1e-9 if dtype == "double" else 1e-4
    monkeypatch.setenv("PA_TEST_SLAB_RHS_KIND", rhs_kind)      # (the rank processes inherit it)
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 40 * 2 ** 30:
        pytest.skip("needs ~ 30 GiB of device memory")
    results = {}
    for world in (2, 4):
        d = tmp_path / f"w{world}"
        d.mkdir()
        spawn_ranks(_worker_config3, lambda port: (world, port, n, K, method, str(d), wl), world)
        results[world] = [torch.load(str(d / f"rank{r}.pt")) for r in range(world)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mesh = Mesh(Box([0.0] * 3, list(upper)), None, list(n), "cuda", dtype)
        var = Field("p", 1, mesh, {"domain": bench.make_bcs(kind), "obstacle": None})
        rhs = bench.synth_rhs(n, 0, n[0], rhs_kind, mesh.dtype.float, mesh.device)
        if method == "euler":
            from pyapes_amd.solver.march import euler_march
            var.set_var_tensor(bench.gaussian(mesh.X, mesh.Y, mesh.Z).to(mesh.dtype.float).unsqueeze(0).contiguous())
            var.apply_bcs()
            nu, dts = bench.euler_params(mesh.dx_list[0])
            euler_march(var, 1.0, nu, dts, K)
            rep = {"itr": K, "tol": 0.0}
        else:
            cfg = {"method": method, "tol": 1e-30, "max_it": K, "report": False}
            if method == "jacobi":
                cfg["omega"] = 0.9
            solver = Solver({"fdm": cfg})
            solver.set_eq(FDM().laplacian(1.0, var) == rhs)
            rep = solver.solve()
    x1 = var()
    assert bool(torch.isfinite(x1).all()) and float(x1.abs().max()) > 0
    for world, parts in results.items():
        if method != "euler":
            assert all(p["in_lib"] and "hostring" in p["impl"] for p in parts), [(p["in_lib"], p["impl"]) for p in parts]
        assert sorted(p["i_off"] for p in parts) == [r * (n[0] // world) for r in range(world)]
        num = den = 0.0
        for p in parts:
            xs = p["x"].cuda()
            ref = x1[:, p["i_off"]:p["i_off"] + xs.shape[1]]
            num += float(((xs - ref).double() ** 2).sum())
            den += float((ref.double() ** 2).sum())
            assert p["itr"] == int(rep["itr"]), (world, p["itr"], rep["itr"])
            assert abs(p["tol"] - float(rep["tol"])) <= tol_bar * abs(float(rep["tol"])), (world, p["tol"], rep["tol"])
            del xs
        assert (num / den) ** 0.5 <= bar, (world, (num / den) ** 0.5)     # (the march has no sums across ranks: bar 0 = the same bits)
