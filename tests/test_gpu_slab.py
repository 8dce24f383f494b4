"""-m gpu: the real HIP kernels in slab mode.  Two ranks share cuda:0 (gloo process group, planes
staged through the host by the driver) -- a rehearsal of the P > 1 path on the one-GPU box: slab
extents, ghost planes, direction-ghost recurrence, periodic far planes, split reductions.
Checked against the single-domain oracle."""
import os
import socket
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyapes_oracle as O
from test_slab_gloo import CASES, _free_port, spawn_ranks

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, jobs, out):
    """jobs: list of (key, BC mix, n, K, dtype, options): solved one after the other by the SAME rank processes (one
    process group; a fresh mesh / context per job, created under that job's PYAPES_HIP_OPTIONS)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    warnings.filterwarnings("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyapes_amd.geometry import Box
        from pyapes_amd.mesh import Mesh
        from pyapes_amd.slab import SlabCG
        from pyapes_amd.variables import Field
        torch.cuda.set_device(0)
        res = {}
        for key, name, n, K, dtype, opts in jobs:
            if opts:
                os.environ["PYAPES_HIP_OPTIONS"] = ",".join(f"{k}={int(v)}" for k, v in opts.items())
            else:
                os.environ.pop("PYAPES_HIP_OPTIONS", None)
            bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
                   for i, (t, v) in enumerate(CASES[name])]
            mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", dtype, slab=(rank, world))
            var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
            g = torch.Generator().manual_seed(7)
            rhs_g = torch.randn((1, *n), generator=g, dtype=torch.float64).to(mesh.dtype.float)
            if name == "per":
                rhs_g -= rhs_g.mean()
            rhs = rhs_g[:, mesh.i_off:mesh.i_off + mesh.nx[0]].contiguous().cuda()
            drv = SlabCG(mesh, var, rhs, [{"kind": 0, "sign": -1.0, "coeff": 0.7}], dist)
            rep = drv.solve(1e-30, K, poll=3)
            parts = [None] * world
            dist.all_gather_object(parts, var().cpu())
            res[key] = {"x": torch.cat(parts, dim=1), "itr": int(rep.itr), "tol": float(rep.tol)}
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


SHAPES = {"fast_f64": ((24, 20, 132), "double"), "generic_f64": ((12, 9, 11), "double"), "fast_f32": ((16, 12, 136), "single")}
K_IT = 6


@pytest.fixture(scope="module")
def two_slabs(tmp_path_factory):
    """ONE pair of rank processes solves every two-slab job of this module (a process start costs more than a solve)."""
    jobs = []
    for name in CASES:
        jobs += [(f"{name}-{sid}", name, n, K_IT, dt, {}) for sid, (n, dt) in SHAPES.items()]
        # the per-axis BC pair kernels (what a large slab uses) / one launch per face instead of the fused fill
        jobs += [(f"{name}-pair", name, (24, 20, 132), K_IT, "double", {"bc_path": 1}),
                 (f"{name}-faces", name, (24, 20, 132), K_IT, "double", {"bc_path": 3})]
    out = str(tmp_path_factory.mktemp("two_slabs") / "res.pt")
    spawn_ranks(_worker, lambda port: (2, port, jobs, out), 2)
    return torch.load(out)


@pytest.fixture(scope="module")
def four_slabs(tmp_path_factory):
    jobs = []
    for name in ("per", "xper", "mix"):
        jobs += [(f"{name}-fused", name, (26, 20, 132), K_IT, "double", {}),
                 (f"{name}-pair", name, (26, 20, 132), K_IT, "double", {"bc_path": 1})]
    out = str(tmp_path_factory.mktemp("four_slabs") / "res.pt")
    spawn_ranks(_worker, lambda port: (4, port, jobs, out), 4)
    return torch.load(out)


def _check_against_oracle(res, name, n, dtype, K=K_IT):
    mesh = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), dtype)
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(CASES[name])]
    g = torch.Generator().manual_seed(7)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64).to(mesh.dtype)
    if name == "per":
        rhs -= rhs.mean()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(mesh, cfg, rhs, method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    assert res["itr"] == ro["itr"] == K + 1
    err = float(torch.linalg.norm(res["x"].double() - xo.double()) / torch.linalg.norm(xo.double()))
    assert err < (1e-10 if dtype == "double" else 1e-5), err


@pytest.mark.parametrize("shape", list(SHAPES), ids=list(SHAPES))
@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
def test_two_slabs_on_one_gpu(name, shape, two_slabs):
    n, dtype = SHAPES[shape]
    _check_against_oracle(two_slabs[f"{name}-{shape}"], name, n, dtype)


@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
def test_two_slabs_pair_bc_kernels(name, two_slabs):
    """same, with the per-axis BC pair kernels (what a large slab uses) instead of the fused fill"""
    paired, plain = two_slabs[f"{name}-pair"], two_slabs[f"{name}-faces"]
    _check_against_oracle(paired, name, (24, 20, 132), "double")
    _check_against_oracle(plain, name, (24, 20, 132), "double")
    assert torch.equal(paired["x"], plain["x"])
    assert abs(paired["tol"] - plain["tol"]) <= 1e-12 * abs(plain["tol"])


def _worker_rccl(rank, world, port, names, n, K, out, runs):
    """runs: list of (label, env) solved one after the other, for every BC mix in names, in ONE process (one RCCL process
    group; a fresh mesh / context / library-side communicator per run)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    warnings.filterwarnings("ignore")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from pyapes_amd.geometry import Box
        from pyapes_amd.hip.context import context_for
        from pyapes_amd.mesh import Mesh
        from pyapes_amd.slab import SlabCG
        from pyapes_amd.variables import Field
        res = {}
        for name in names:
            bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
                   for i, (t, v) in enumerate(CASES[name])]
            g = torch.Generator().manual_seed(7)
            rhs_g = torch.randn((1, *n), generator=g, dtype=torch.float64)
            if name == "per":
                rhs_g -= rhs_g.mean()
            for label, env in runs:
                for k in ("PYAPES_HIP_COMM", "PYAPES_HIP_COMM_OVERLAP", "PYAPES_HIP_SLAB_FOLD"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                lib_comm = env.get("PYAPES_HIP_COMM", "1") != "0"
                mesh = Mesh(Box[0:1, 0:1, 0:0.5], None, list(n), "cuda", "double", slab=(rank, world))
                var = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
                drv = SlabCG(mesh, var, rhs_g.cuda(), [{"kind": 0, "sign": -1.0, "coeff": 0.7}], dist)
                assert drv.lib_comm == lib_comm, "library-side RCCL communicator not in use"
                rep = drv.solve(1e-30, K, poll=3)
                want_fold = lib_comm and env.get("PYAPES_HIP_SLAB_FOLD", "1") != "0"
                assert drv.folded == want_fold, f"folded={drv.folded}, expected {want_fold}"
                if lib_comm:   # a second solve on the same mesh reuses the communicator (no silent fallback)
                    first = var().clone()
                    var2 = Field("p", 1, mesh, {"domain": bcs, "obstacle": None})
                    drv2 = SlabCG(mesh, var2, rhs_g.cuda(), [{"kind": 0, "sign": -1.0, "coeff": 0.7}], dist)
                    assert drv2.lib_comm, "second solve fell back to the stepwise driver"
                    rep2 = drv2.solve(1e-30, K, poll=3)
                    assert torch.equal(var2(), first) and rep2.itr == rep.itr
                    torch.cuda.synchronize()          # release the communicators before the next run makes its own
                    context_for(mesh).comm_destroy()
                    context_for(mesh).comm_ready = None
                res[(name, label)] = {"x": var().cpu(), "itr": int(rep.itr), "tol": float(rep.tol)}
        torch.save(res, out)
    finally:
        dist.destroy_process_group()


LIB_MODES = {
    "folded": {},                                                  # default: rows all-reduced, mid kernel; with ONE
                                                                   # rank the exchange stays on the ctx stream
    "folded_two_streams": {"PYAPES_HIP_COMM_OVERLAP": "1"},        # what N > 1 ranks run: second communicator + stream
    "stepwise_in_library": {"PYAPES_HIP_SLAB_FOLD": "0"},                    # the round-1 sequence, still in C
}


@pytest.fixture(scope="module")
def one_rank_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("one_rank") / "x.pt")
    runs = [("stepwise_driver", {"PYAPES_HIP_COMM": "0"})] + [(m, LIB_MODES[m]) for m in LIB_MODES]
    spawn_ranks(_worker_rccl, lambda port: (1, port, ["per", "xper", "mix"], (24, 20, 132), K_IT, out, runs), 1)
    return torch.load(out)


@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_library_side_rccl_one_rank(name, one_rank_rccl):
    """The C-side iteration loop (pa_cg_iterate_comm) with a 1-rank RCCL communicator -- on a periodic
    axis 0 the rank is its own ring neighbour, so the packed plane exchange really runs -- against the
    stepwise torch.distributed driver (bit for bit: with one rank the folded sequence adds the same
    partial rows in the same order) and the single-domain oracle.  Modes: the folded sequence (row
    all-reduces, mid kernel), the same with the exchange on the second communicator / stream (the default
    for N > 1 ranks), and the stepwise sequence inside the library.  (One rank process runs all of it.)"""
    allres = {label: v for (nm, label), v in one_rank_rccl.items() if nm == name}
    ref = allres["stepwise_driver"]
    for m in LIB_MODES:
        assert torch.equal(allres[m]["x"], ref["x"]) and allres[m]["itr"] == ref["itr"] == K_IT + 1, m
        assert allres[m]["tol"] == ref["tol"], m
    _check_against_oracle(allres["folded"], name, (24, 20, 132), "double")


@pytest.mark.parametrize("bc_path", ["fused", "pair"])
@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_four_slabs_on_one_gpu(name, bc_path, four_slabs):
    """P = 4: two interior ranks that own no global x face (no x BC fill, both neighbours real), the
    uneven split 26 = 7 + 7 + 6 + 6, a periodic ring longer than its two end ranks"""
    _check_against_oracle(four_slabs[f"{name}-{bc_path}"], name, (26, 20, 132), "double")
