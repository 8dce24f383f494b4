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


def _worker(rank, world, port, name, n, K, dtype, out):
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
        if rank == 0:
            torch.save({"x": torch.cat(parts, dim=1), "itr": int(rep.itr), "tol": float(rep.tol)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [((24, 20, 132), "double"), ((12, 9, 11), "double"), ((16, 12, 136), "single")],
                         ids=["fast_f64", "generic_f64", "fast_f32"])
@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
def test_two_slabs_on_one_gpu(name, shape, tmp_path, world=2):
    (n, dtype), K = shape, 6
    out = str(tmp_path / "x.pt")
    spawn_ranks(_worker, lambda port: (world, port, name, n, K, dtype, out), world)
    res = torch.load(out)
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


@pytest.mark.parametrize("name", list(CASES), ids=list(CASES))
def test_two_slabs_pair_bc_kernels(name, tmp_path, monkeypatch):
    """same, with the per-axis BC pair kernels (what a large slab uses) instead of the fused fill"""
    from helpers import hip_options
    hip_options(monkeypatch, bc_path=1)      # (PYAPES_HIP_OPTIONS: the rank processes inherit it)
    test_two_slabs_on_one_gpu(name, ((24, 20, 132), "double"), tmp_path)
    paired = torch.load(str(tmp_path / "x.pt"))
    hip_options(monkeypatch, bc_path=3)
    test_two_slabs_on_one_gpu(name, ((24, 20, 132), "double"), tmp_path)
    plain = torch.load(str(tmp_path / "x.pt"))
    assert torch.equal(paired["x"], plain["x"])
    assert abs(paired["tol"] - plain["tol"]) <= 1e-12 * abs(plain["tol"])


def _worker_rccl(rank, world, port, name, n, K, out, runs):
    """runs: list of (label, env) solved one after the other in ONE process (one RCCL process group; a fresh mesh /
    context / library-side communicator per run)."""
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
        bcs = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v, "bc_val_opt": None}
               for i, (t, v) in enumerate(CASES[name])]
        g = torch.Generator().manual_seed(7)
        rhs_g = torch.randn((1, *n), generator=g, dtype=torch.float64)
        if name == "per":
            rhs_g -= rhs_g.mean()
        res = {}
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
            res[label] = {"x": var().cpu(), "itr": int(rep.itr), "tol": float(rep.tol)}
        torch.save(res, out)
    finally:
        dist.destroy_process_group()


LIB_MODES = {
    "folded": {},                                                  # default: rows all-reduced, mid kernel; with ONE
                                                                   # rank the exchange stays on the ctx stream
    "folded_two_streams": {"PYAPES_HIP_COMM_OVERLAP": "1"},        # what N > 1 ranks run: second communicator + stream
    "stepwise_in_library": {"PYAPES_HIP_SLAB_FOLD": "0"},                    # the round-1 sequence, still in C
}


@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_library_side_rccl_one_rank(name, tmp_path):
    """The C-side iteration loop (pa_cg_iterate_comm) with a 1-rank RCCL communicator -- on a periodic
    axis 0 the rank is its own ring neighbour, so the packed plane exchange really runs -- against the
    stepwise torch.distributed driver (bit for bit: with one rank the folded sequence adds the same
    partial rows in the same order) and the single-domain oracle.  Modes: the folded sequence (row
    all-reduces, mid kernel), the same with the exchange on the second communicator / stream (the default
    for N > 1 ranks), and the stepwise sequence inside the library.  (One rank process per BC mix runs all four.)"""
    if name not in CASES:
        pytest.skip(name)
    n, K = (24, 20, 132), 6
    out = str(tmp_path / "x.pt")
    runs = [("stepwise_driver", {"PYAPES_HIP_COMM": "0"})] + [(m, LIB_MODES[m]) for m in LIB_MODES]
    spawn_ranks(_worker_rccl, lambda port: (1, port, name, n, K, out, runs), 1)
    allres = torch.load(out)
    ref = allres["stepwise_driver"]
    for m in LIB_MODES:
        assert torch.equal(allres[m]["x"], ref["x"]) and allres[m]["itr"] == ref["itr"] == K + 1, m
        assert allres[m]["tol"] == ref["tol"], m
    res = {True: allres["folded"], False: ref}
    mesh = O.OMesh([0, 0, 0], [1, 1, 0.5], list(n), "double")
    cfg = [{"bc_face": O.FACES[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(CASES[name])]
    g = torch.Generator().manual_seed(7)
    rhs = torch.randn((1, *n), generator=g, dtype=torch.float64)
    if name == "per":
        rhs -= rhs.mean()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xo, ro = O.solve_poisson(mesh, cfg, rhs, method="cg", tol=1e-30, max_it=K, coeff=0.7, sign=-1.0)
    err = float(torch.linalg.norm(res[True]["x"] - xo) / torch.linalg.norm(xo))
    assert err < 1e-10, err


@pytest.mark.parametrize("bc_path", ["fused", "pair"])
@pytest.mark.parametrize("name", ["per", "xper", "mix"])
def test_four_slabs_on_one_gpu(name, bc_path, tmp_path, monkeypatch):
    """P = 4: two interior ranks that own no global x face (no x BC fill, both neighbours real), the
    uneven split 26 = 7 + 7 + 6 + 6, a periodic ring longer than its two end ranks"""
    if bc_path == "pair":
        from helpers import hip_options
        hip_options(monkeypatch, bc_path=1)
    test_two_slabs_on_one_gpu(name, ((26, 20, 132), "double"), tmp_path, world=4)
