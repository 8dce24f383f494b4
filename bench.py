#!/usr/bin/env python3
"""bench.py -- headline benchmark: cell-updates x iterations / second of the fused CG
iteration on a 3-D Poisson problem (BASELINE.json metric), plus its HBM roofline
fraction and the CPU baseline timed on the host of the same box.

A "step" is ONE CG iteration (stencil apply + AXPYs + two dot reductions + BC fill +
stop test) over the whole grid.  Default workload at every N: BASELINE config 3, the
configuration the metric is quoted on -- 3-D Poisson 512^3 fp64, periodic BCs, CG --
slab-decomposed along axis 0 for N > 1 (strong scaling: the global grid is fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5] [--size n0,n1,n2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable
# algorithmic HBM bytes per cell per CG iteration: 10 array passes
ALG_PASSES_CG = 10
ALG_PASSES_PHASE = {"a": 4, "b": 6}

WORKLOADS = {
    # name: (global nodes, dtype, bc kind, box upper)
    "c3": ((512, 512, 512), "double", "periodic", (1.0, 1.0, 1.0)),
    "c2": ((256, 256, 256), "double", "dirichlet", (1.0, 1.0, 1.0)),
    "c5": ((1024, 1024, 512), "single", "mixed", (1.0, 1.0, 0.5)),
}


def make_bcs(kind):
    from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs
    if kind == "periodic":
        return homogeneous_bcs(3, None, "periodic")
    if kind == "dirichlet":
        return homogeneous_bcs(3, 0.0, "dirichlet")
    return mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])


def synth_rhs(mesh, kind):
    """Deterministic synthetic right-hand side as a function of the GLOBAL node index, so every
    decomposition sees the same problem.  periodic: ring-mean-zero product of sines plus a
    zero-mean deterministic 'noise' (SURVEY 8d C3); otherwise sin(pi x) sin(pi y) sin(pi z) + noise."""
    f = mesh.dtype.float
    dev = mesh.device
    gn = mesh.global_nx
    idx = [torch.arange(mesh.i_off, mesh.i_off + mesh.nx[0], device=dev, dtype=torch.float64),
           torch.arange(gn[1], device=dev, dtype=torch.float64),
           torch.arange(gn[2], device=dev, dtype=torch.float64)]
    I, J, K = torch.meshgrid(idx, indexing="ij")
    if kind == "periodic":
        base = torch.sin(2 * math.pi * I / gn[0]) * torch.sin(2 * math.pi * J / gn[1]) * torch.sin(2 * math.pi * K / gn[2])
        # zero-mean on the ring in every axis: cos of an integer number of periods
        noise = 0.25 * torch.cos(2 * math.pi * (3 * I / gn[0] + 5 * J / gn[1] + 7 * K / gn[2]))
    else:
        base = torch.sin(math.pi * I / (gn[0] - 1)) * torch.sin(math.pi * J / (gn[1] - 1)) * torch.sin(math.pi * K / (gn[2] - 1))
        noise = 0.25 * torch.sin(12.9898 * I + 78.233 * J + 37.719 * K)
    return (base + noise).to(f).unsqueeze(0).contiguous()


def host_cores() -> int:
    """CPUs this process may actually use: min(os.cpu_count(), affinity, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(kind, dtype, target_s=12.0):
    """The oracle (literal torch-CPU restatement of the reference algorithm) timed on this box's
    host cores on a bounded sample of the same workload family.  Checker code used ONLY as the
    reported baseline, never on the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyapes_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle CG on {cores} host threads ...")
    n = 256
    its = 24
    mesh = O.OMesh([0, 0, 0], [1, 1, 1], [n, n, n], dtype)
    if kind == "periodic":
        cfg = O.homogeneous_cfg(3, None, "periodic")
    elif kind == "dirichlet":
        cfg = O.homogeneous_cfg(3, 0.0, "dirichlet")
    else:
        cfg = O.mixed_cfg([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])
    g = torch.Generator().manual_seed(0)
    rhs = torch.randn((1, n, n, n), generator=g, dtype=torch.float64).to(mesh.dtype)
    rhs -= rhs.mean()
    bcs = O.make_bcs(mesh, cfg)
    x = torch.zeros(1, n, n, n, dtype=mesh.dtype)
    tabs = O.laplacian_tables(x, mesh, bcs)
    rhs += O.laplacian_rhs_adjust(x, mesh, bcs)
    terms = [O.OTerm("laplacian", tabs, 1.0, 1.0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        _, rep = O.cg(x, rhs, terms, mesh, bcs, -1.0, its - 1)
        dt = time.perf_counter() - t0
    val = n ** 3 * rep["itr"] / dt
    return {"value": val, "unit": "cell-updates*iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle CG (torch-CPU literal restatement), {n}^3 {dtype} {kind}, {rep['itr']} iterations, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--size", dest="n", default=None, help="override global node counts n0,n1,n2 (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-probe", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: anything native libraries print on file
    # descriptor 1 in the meantime (RCCL prints a version banner when a communicator comes up) goes
    # to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU path"
    # rehearsal switches (one-GPU box): BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0,
    # BENCH_BACKEND=gloo moves the planes through the host instead of RCCL
    if os.environ.get("BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.hip import lib as L
    from pyapes_amd.hip.context import context_for
    from pyapes_amd.variables import Field

    gn, dtype, kind, upper = WORKLOADS[args.workload]
    if args.n:
        gn = tuple(int(v) for v in args.n.split(","))
    esize = 8 if dtype == "double" else 4

    dist = None
    force_slab = bool(os.environ.get("BENCH_FORCE_SLAB"))  # rehearsal: drive the slab path with one rank
    if world > 1 or force_slab:
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    mesh = Mesh(Box([0.0, 0.0, 0.0], list(upper)), None, list(gn), "cuda", dtype,
                slab=(rank, world) if (world > 1 or force_slab) else None)
    var = Field("p", 1, mesh, {"domain": make_bcs(kind), "obstacle": None})
    rhs = synth_rhs(mesh, kind)
    log(f"rank {rank}/{world}: workload {args.workload} global {gn} local {tuple(mesh.nx)} {dtype} {kind}")
    cells_global = gn[0] * gn[1] * gn[2]
    W, K = args.warmup, args.steps
    terms = [{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}]

    if world == 1 and not force_slab:
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        ctx.set_terms(terms)
        ctx.rhs_adjust(rhs[0])
        # tolerance -1: the stop test can never end the solve; max_it beyond W + K
        ctx.cg_begin(var()[0], rhs[0], -1.0, W + K + 10)
        ctx.cg_iterate(W)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ctx.stream)
        ctx.cg_iterate(K)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        log(f"timed {K} iterations: {wall*1e3/K:.4f} ms/iter")
        rep = ctx.report()
        assert rep.itr == W + K and rep.status == 0, f"work was skipped: itr={rep.itr} status={rep.status}"
        secs = wall
        finite = bool(torch.isfinite(var()).all())
        assert finite, "iterate became non-finite inside the timed region"
        roof = None
        if not args.no_roofline_probe:
            # per-kernel durations of the two dominant kernels, HIP events on the launch stream
            ctx.profile(True)
            ctx.cg_iterate(min(K, 10))
            pr = ctx.profile_read()
            ctx.profile(False)
            local_cells = mesh.N
            dom = "b" if pr["phase_b_ms"] >= pr["phase_a_ms"] else "a"
            ms = pr[f"phase_{dom}_ms"]
            alg_bytes = ALG_PASSES_PHASE[dom] * esize * local_cells
            ach = alg_bytes / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"cg_phase_{dom}", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "phase_a_ms": pr["phase_a_ms"], "phase_b_ms": pr["phase_b_ms"],
                    "alg_bytes_per_launch": alg_bytes}
        ctx.cg_end()
    else:
        from pyapes_amd.slab import SlabCG
        drv = SlabCG(mesh, var, rhs, terms, dist)
        drv.begin(-1.0, W + K + 10)
        drv.iterate(W)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.iterate(K)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        secs_local = time.perf_counter() - t0
        t = torch.tensor([secs_local], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
        ev_ms = secs * 1e3
        rep = drv.be.report()
        assert rep.itr == W + K and rep.status == 0, f"work was skipped: itr={rep.itr} status={rep.status}"
        assert bool(torch.isfinite(var()).all()), "iterate became non-finite inside the timed region"
        roof = None
        if not args.no_roofline_probe:
            pr = drv.profile(min(K, 10))
            dom = "b" if pr["phase_b_ms"] >= pr["phase_a_ms"] else "a"
            ms = pr[f"phase_{dom}_ms"]
            alg_bytes = ALG_PASSES_PHASE[dom] * esize * mesh.N
            ach = alg_bytes / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"cg_phase_{dom}", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "phase_a_ms": pr["phase_a_ms"], "phase_b_ms": pr["phase_b_ms"],
                    "alg_bytes_per_launch": alg_bytes, "scope": "rank 0, per GPU"}
        drv.end()
        comm_kind = "rccl-in-library" if drv.lib_comm else "torch.distributed-stepwise"
        if drv.lib_comm:            # release the library's communicator while every rank is still alive
            torch.cuda.synchronize()
            drv.be.comm_destroy()
            drv.be.comm_ready = None

    if rank == 0:
        value = cells_global * K / secs
        out = {
            "metric": "cell-updates*iters/sec, 3-D Poisson CG",
            "value": value,
            "unit": "cell-updates*iters/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": secs * 1e3 / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if dtype == "double" else "f32",
            "data": "synthetic",
            "config": {"workload": f"3-D Poisson {gn[0]}x{gn[1]}x{gn[2]} {dtype}, {kind} BCs, CG (BASELINE config {args.workload})",
                       "global_cells": cells_global, "parallelism": (f"slab{world} ({comm_kind})" if (world > 1 or force_slab) else "single")},
            "hbm_alg_GBs": ALG_PASSES_CG * esize * cells_global * K / secs / 1e9,
            "hbm_alg_frac_of_peak": ALG_PASSES_CG * esize * cells_global * K / secs / 1e9 / (HBM_PEAK_GBS * world),
            "stream_ms_per_step": ev_ms / K,
        }
        if roof is not None:
            tr = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tr):
                try:
                    with open(tr) as f:
                        t = json.load(f)
                    key = f"{args.workload}:{roof['kernel']}"
                    if key in t and not args.n:
                        roof["traffic"] = t[key]["bytes_per_launch"]
                        roof["traffic_source"] = t[key].get("source")
                except Exception:
                    pass
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(kind, dtype)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
