#!/usr/bin/env python3
"""bench_ops.py -- secondary measurements for the other rows of SURVEY section 8 (not the driver's
contract; bench.py is).  One JSON line per workload: explicit Laplacian apply, the explicit
adv-diff Euler march (BASELINE config 4), Jacobi (config 1 and 3-D), BiCGSTAB, 2-D CG.
achieved GB/s uses the ALGORITHMIC bytes of SURVEY 8d (apply 2 passes, Euler 2-3, Jacobi 3,
CG 10, BiCGSTAB 22 = 2 applies x 2 + 9 axpy/dot passes x 2) against the 8 TB/s HBM peak.

    python bench_ops.py [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PEAK = 8000.0


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def emit(name, cells, ms, passes, esize, extra=None):
    gbs = passes * esize * cells / (ms * 1e-3) / 1e9
    out = {"workload": name, "ms": ms, "cell_updates_per_s": cells / (ms * 1e-3), "alg_passes": passes,
           "alg_GBs": gbs, "frac_of_hbm_peak": gbs / PEAK}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sections", default="ops,euler,small,big",
                    help="comma-separated subset of: ops (explicit operators 512^3), euler (config 4 march), small (the reference's "
                         "own mesh sizes, resident vs launch per phase), big (Jacobi / BiCGSTAB 256^3, 2-D 4096^2, odd extents)")
    args = ap.parse_args()
    sections = set(args.sections.split(","))
    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.solver.fdc import FDC
    from pyapes_amd.solver.fdm import FDM
    from pyapes_amd.solver.march import euler_march, euler_step
    from pyapes_amd.solver.ops import Solver
    from pyapes_amd.testing.poisson import poisson_bcs, poisson_rhs_nd
    from pyapes_amd.variables import Field
    from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs
    warnings.simplefilter("ignore")
    q = args.quick

    # --- explicit operators, 512^3: the C call on a pre-allocated output (no torch.empty, no Python wrapper
    #     beyond ctypes), fp64 and fp32.  Laplacian / Div: 2 passes; Grad: 1 read + 3 writes = 4 passes --------
    from pyapes_amd.hip import lib as L
    from pyapes_amd.hip.context import context_for
    n = 256 if q else 512
    for dt, es in ((("double", 8), ("single", 4)) if "ops" in sections else ()):
        mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", dt)
        var = Field("p", 1, mesh, {"domain": homogeneous_bcs(3, 0.0, "neumann"), "obstacle": None}, init_val="random")
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        x = var()[0]
        y = torch.empty_like(x)
        g3 = torch.empty((3, n, n, n), dtype=x.dtype, device=x.device)
        f = "f64" if dt == "double" else "f32"
        ms = timed(lambda: ctx.laplacian(x, False, out=y), 20)
        emit(f"laplacian apply {n}^3 {f} neumann (C call, pre-allocated output)", n ** 3, ms, 2, es)
        ms = timed(lambda: ctx.div(L.OP_DIV_UPWIND, 1.0, x, out=y), 20)
        emit(f"div apply {n}^3 {f} upwind, scalar speed (C call, pre-allocated output)", n ** 3, ms, 2, es)
        ms = timed(lambda: ctx.grad(x, False, out=g3), 20)
        emit(f"grad apply {n}^3 {f} -> 3 components (C call, pre-allocated output)", n ** 3, ms, 4, es)
        # the ceiling these are measured against: a device copy of the same array (1 read + 1 write)
        ms = timed(lambda: y.copy_(x), 20)
        emit(f"torch copy_ {n}^3 {f} (reference point: 1 read + 1 write)", n ** 3, ms, 2, es)
        del var, mesh, ctx, x, y, g3

    # --- config 4: explicit adv-diff march 256^3 fp32, upwind, Neumann / Symmetry ------------------
    if "euler" in sections:
        euler_rows(q, emit)
    solver_rows(q, emit, sections)


def euler_rows(q, emit):
    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.solver.march import euler_march, euler_step
    from pyapes_amd.variables import Field
    from pyapes_amd.variables.bcs import mixed_bcs
    n = 128 if q else 256
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "single")
    bcs = mixed_bcs([0.0, 0.0, None, None, None, None],
                    ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
    phi = Field("phi", 1, mesh, {"domain": bcs, "obstacle": None})
    phi.set_var_tensor(torch.exp(-((mesh.X - 0.5) ** 2 + (mesh.Y - 0.5) ** 2 + (mesh.Z - 0.5) ** 2) / 0.02)
                       .unsqueeze(0).contiguous())
    phi.apply_bcs()
    nu = 1e-3
    dx = mesh.dx_list[0]
    dt = 0.2 * min(dx * dx / (6 * nu), dx / 1.0)
    cfg = {"div": {"limiter": "upwind"}}
    ms = timed(lambda: euler_step(phi, 1.0, nu, dt, cfg), 100)
    emit(f"euler adv-diff step {n}^3 f32 upwind scalar u, neumann/symmetry (config 4)", n ** 3, ms, 2, 4)
    ut = torch.ones_like(phi()) * 0.7
    ms = timed(lambda: euler_step(phi, ut, nu, dt, cfg), 100)
    emit(f"euler adv-diff step {n}^3 f32 upwind speed tensor (config 4)", n ** 3, ms, 3, 4)
    ms = timed(lambda: euler_march(phi, 1.0, nu, dt, 50, cfg), 4) / 50
    emit(f"euler_march (50 steps per call) {n}^3 f32 upwind scalar u (config 4)", n ** 3, ms, 2, 4)
    ms = timed(lambda: euler_march(phi, ut, nu, dt, 50, cfg), 4) / 50
    emit(f"euler_march (50 steps per call) {n}^3 f32 upwind speed tensor (config 4)", n ** 3, ms, 3, 4)
    assert bool(torch.isfinite(phi()).all())
    del phi, mesh, ut
    if not q:   # the same march at 512^3 fp32: 512 MiB per array, outside the 256 MiB Infinity Cache
        mesh = Mesh(Box[0:1, 0:1, 0:1], None, [512, 512, 512], "cuda", "single")
        phi = Field("phi", 1, mesh, {"domain": bcs, "obstacle": None})
        phi.set_var_tensor(torch.exp(-((mesh.X - 0.5) ** 2 + (mesh.Y - 0.5) ** 2 + (mesh.Z - 0.5) ** 2) / 0.02)
                           .unsqueeze(0).contiguous())
        phi.apply_bcs()
        dx = mesh.dx_list[0]
        dt = 0.2 * min(dx * dx / (6 * nu), dx / 1.0)
        ms = timed(lambda: euler_march(phi, 1.0, nu, dt, 20, cfg), 3) / 20
        emit("euler_march (20 steps per call) 512^3 f32 upwind scalar u", 512 ** 3, ms, 2, 4)
        del phi, mesh



def solver_rows(q, emit, sections):
    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.solver.fdm import FDM
    from pyapes_amd.solver.ops import Solver
    from pyapes_amd.testing.poisson import poisson_bcs, poisson_rhs_nd
    from pyapes_amd.variables import Field
    from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs

    # --- solvers: ms per iteration from fixed-iteration solves ------------------------------------
    def solver_ms(meshf, bcsf, method, K, rhs_fn=None, extra_cfg=None, resident=True):
        os.environ["PYAPES_HIP_RESIDENT"] = "1" if resident else "0"   # read when the mesh's context is created
        mesh = meshf()
        var = Field("p", 1, mesh, {"domain": bcsf, "obstacle": None})
        rhs = rhs_fn(mesh, var) if rhs_fn else torch.randn_like(var())
        cfg = {"method": method, "tol": -1.0, "max_it": K - 1, "report": False}
        cfg.update(extra_cfg or {})
        # one short untimed solve first (same mesh / context): code-object load, scratch allocation
        warm = Field("p", 1, mesh, {"domain": bcsf, "obstacle": None})
        sw = Solver({"fdm": dict(cfg, max_it=3)})
        sw.set_eq(FDM().laplacian(1.0, warm) == rhs.clone())
        sw.solve()
        s = Solver({"fdm": cfg})
        s.set_eq(FDM().laplacian(1.0, var) == rhs)
        t0 = time.perf_counter()
        rep = s.solve()
        wall = (time.perf_counter() - t0) * 1e3
        from pyapes_amd.hip.context import context_for as _cf
        solver_ms.boxes = _cf(mesh).resident_used()
        return var.last_gpu_ms / rep["itr"], wall / rep["itr"], rep["itr"], mesh.N

    K = 200
    # the sizes pyapes users run (the reference's tests and demos): the whole solve is ONE cooperative launch with
    # the fields in LDS (pa_resident.hip); beside it the launch-per-phase loops it replaces (DESIGN.md "small meshes")
    m2 = lambda nn: (lambda: Mesh(Box[0:1, 0:1], None, [nn, nn], "cuda", "double"))
    m3 = lambda nn: (lambda: Mesh(Box[0:1, 0:1, 0:1], None, [nn, nn, nn], "cuda", "double"))
    mixbc = mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann"] * 3)
    m1 = lambda nn: (lambda: Mesh(Box[0:1], None, [nn], "cuda", "double"))
    small = [("cg 1-D 101 nodes f64 dirichlet (the reference's 1-D Poisson test: one box, no grid-wide step)", m1(101), poisson_bcs(1), "cg", 100,
              10, poisson_rhs_nd),
             ("bicgstab 1-D 101 nodes f64 dirichlet", m1(101), poisson_bcs(1), "bicgstab", 60, 22, poisson_rhs_nd),
             ("cg 2-D 32x32 f64 dirichlet (one box)", m2(32), poisson_bcs(2), "cg", 100, 10, poisson_rhs_nd),
             ("jacobi 2-D 128x128 f64 dirichlet (config 1)", m2(128), poisson_bcs(2), "jacobi", 1000, 3, poisson_rhs_nd),
             ("cg 2-D 128x128 f64 dirichlet (config 1 inputs)", m2(128), poisson_bcs(2), "cg", 271, 10, poisson_rhs_nd),
             ("bicgstab 2-D 128x128 f64 dirichlet (config 1 inputs)", m2(128), poisson_bcs(2), "bicgstab", 100, 22, poisson_rhs_nd)]
    # the reference's own solver tests at their sizes: x-periodic 101^2 (tests/test_solver.py:164-207) and the
    # axisymmetric 101^2 Poisson problem (tests/test_solver.py:309-358; round 3: the resident solver's lean stencil
    # takes the r rows from an LDS copy of pa_coord_set's table)
    from pyapes_amd.geometry import Cylinder
    from pyapes_amd.testing.poisson import poisson_rz_bcs, poisson_rz_rhs
    from pyapes_amd.variables.bcs import CylinderBoundary
    rzc = poisson_rz_bcs()
    rz_bcs = CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0.0}, ru={"bc_type": "dirichlet", "bc_val": rzc[1]["bc_val"]},
                              zl={"bc_type": "dirichlet", "bc_val": rzc[2]["bc_val"]},
                              zu={"bc_type": "dirichlet", "bc_val": rzc[3]["bc_val"]})()
    mrz = lambda: Mesh(Cylinder[0:1, 0:1], None, [101, 101], "cuda", "double")
    xper = mixed_bcs([None, None, 0, 0], ["periodic", "periodic", "dirichlet", "dirichlet"])
    small += [("bicgstab 2-D 101x101 f64 x-periodic / y-dirichlet (the reference's test_poisson_2d_mixed_periodic)", m2(101), xper,
               "bicgstab", 100, 22, None),
              ("bicgstab axisymmetric (rz) 101x101 f64 (the reference's test_poisson_rz)", mrz, rz_bcs, "bicgstab", 100, 22,
               poisson_rz_rhs),
              ("cg axisymmetric (rz) 101x101 f64", mrz, rz_bcs, "cg", 100, 10, poisson_rz_rhs)]
    for nn in (32, 64):
        for meth, its, passes in (("jacobi", 200, 3), ("cg", 60, 10), ("bicgstab", 40, 22)):
            small.append((f"{meth} 3-D {nn}^3 f64 dirichlet/neumann faces (BC fill every iteration)", m3(nn), mixbc, meth, its,
                          passes, None))
    for name, meshf, bcs_, meth, its, passes, rfn in (small if "small" in sections else []):
        for res in (True, False):
            torch.manual_seed(0)
            ms, wall, itr, N = solver_ms(meshf, bcs_, meth, its, rfn, resident=res)
            how = f"resident, {solver_ms.boxes} workgroups" if solver_ms.boxes else "launch per phase"
            emit(f"{name} [{how}]", N, ms, passes, 8, {"wall_ms_per_iter": wall, "iters": itr})
    os.environ["PYAPES_HIP_RESIDENT"] = "1"
    if "big" not in sections:
        return
    n = 128 if q else 256
    ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double"),
                                 homogeneous_bcs(3, 0.0, "dirichlet"), "jacobi", K)
    emit(f"jacobi 3-D {n}^3 f64 dirichlet", N, ms, 3, 8, {"wall_ms_per_iter": wall, "iters": itr})
    ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double"),
                                 mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann"] * 3), "bicgstab", 100)
    emit(f"bicgstab 3-D {n}^3 f64 mixed", N, ms, 22, 8, {"wall_ms_per_iter": wall, "iters": itr})
    if not q:
        # round 4: the three solver loops at BASELINE config 3's size (512^3 fp64), Dirichlet and fully periodic -- BiCGSTAB is
        # the method that converges on the periodic problem; passes = the algorithmic count
        per3 = mixed_bcs([None] * 6, ["periodic"] * 6)
        for bname, bcs3 in (("dirichlet", homogeneous_bcs(3, 0.0, "dirichlet")), ("periodic", per3)):
            for meth, its, passes in (("jacobi", 60, 3), ("cg", 60, 10), ("bicgstab", 40, 22)):
                ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1, 0:1], None, [512, 512, 512], "cuda", "double"), bcs3, meth, its)
                emit(f"{meth} 3-D 512^3 f64 {bname}", N, ms, passes, 8, {"wall_ms_per_iter": wall, "iters": itr})
                torch.cuda.empty_cache()
    def advdiff_ms(n):
        mesh = Mesh(Box[0:1, 0:1, 0:1], None, [n, n, n], "cuda", "double")
        var = Field("p", 1, mesh, {"domain": homogeneous_bcs(3, 0.0, "dirichlet"), "obstacle": None})
        s = Solver({"fdm": {"method": "bicgstab", "tol": -1.0, "max_it": 60, "report": False}})
        fdm = FDM({"div": {"limiter": "upwind", "edge": False}})
        s.set_eq(fdm.div(1.0, var) - fdm.laplacian(0.05, var) == torch.ones_like(var()))
        rep = s.solve()
        return var.last_gpu_ms / rep["itr"], rep["itr"], mesh.N

    ms, itr, N = advdiff_ms(n)
    emit(f"bicgstab 3-D {n}^3 f64 steady advection-diffusion (upwind Div + Laplacian)", N, ms, 22, 8, {"iters": itr})
    m2 = 1024 if q else 4096
    ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1], None, [m2, m2], "cuda", "double"),
                                 homogeneous_bcs(2, 0.0, "dirichlet"), "cg", 100)
    emit(f"cg 2-D {m2}x{m2} f64 dirichlet (marching kernel k_cg2d from 1.5 M cells on)", N, ms, 10, 8, {"wall_ms_per_iter": wall, "iters": itr})
    # round 3, second session: the Jacobi sweep and the BiCGSTAB phases of large 2-D meshes on the marching kernel too;
    # odd extents (node-based meshes: 2^k + 1) through the PITCH layout
    ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1], None, [m2, m2], "cuda", "double"),
                                 homogeneous_bcs(2, 0.0, "dirichlet"), "jacobi", 100)
    emit(f"jacobi 2-D {m2}x{m2} f64 dirichlet (k_cg2d)", N, ms, 3, 8, {"wall_ms_per_iter": wall, "iters": itr})
    ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1], None, [m2, m2], "cuda", "double"),
                                 homogeneous_bcs(2, 0.0, "dirichlet"), "bicgstab", 60)
    emit(f"bicgstab 2-D {m2}x{m2} f64 dirichlet (k_cg2d phases 6 / 8)", N, ms, 22, 8, {"wall_ms_per_iter": wall, "iters": itr})
    n3 = 129 if q else 257
    mixbc3 = mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann"] * 3)
    for meth, its, passes in (("cg", 100, 10), ("bicgstab", 60, 22)):
        ms, wall, itr, N = solver_ms(lambda: Mesh(Box[0:1, 0:1, 0:1], None, [n3, n3, n3], "cuda", "double"), mixbc3, meth, its)
        emit(f"{meth} 3-D {n3}^3 f64 mixed (odd rows: PITCH layout)", N, ms, passes, 8, {"wall_ms_per_iter": wall, "iters": itr})


if __name__ == "__main__":
    main()
