"""The oracle (CPU restatement) against the golden vectors produced by the reference.

These pin the oracle: operator / BC-fill / rhs outputs bit-exact, solver results
<= 1e-13 rel (fp64) with identical iteration counts.  CPU only.
"""
import warnings

import numpy as np
import pytest
import torch

import pyapes_oracle as O
from conftest import golden_cases, golden_load

warnings.filterwarnings("ignore")


def _cfg(case):
    nd = len(case["lower"])
    if case["bcs"] == "poisson":
        return O.poisson_cfg(nd)
    if case["bcs"] == "poisson_rz":
        return O.poisson_rz_cfg()
    if case["bcs"] == "robin":
        return O.robin_cfg(nd)
    faces = O.FACES_RZ if case.get("coord", "xyz") == "rz" else O.FACES
    return [{"bc_face": faces[i], "bc_type": t, "bc_val": v} for i, (t, v) in enumerate(case["bcs"])]


def _mesh(case):
    return O.OMesh(case["lower"], case["upper"], case["spacing"], case["dtype"], case.get("coord", "xyz"))


def _eq(a, b, what):
    assert torch.equal(torch.as_tensor(a), torch.as_tensor(b)), f"{what}: not bit-exact"


@pytest.mark.parametrize("case", golden_cases("ops"), ids=lambda c: c["name"])
def test_oracle_ops_bit_exact(case):
    g = golden_load(case["name"])
    mesh = _mesh(case)
    nd = mesh.dim
    bcs = O.make_bcs(mesh, _cfg(case))
    x = torch.from_numpy(g["x0"]).clone()
    O.bc_fill(x, bcs)
    _eq(x, g["bc_fill"], "bc_fill")
    tabs = O.laplacian_tables(x, mesh, bcs)
    rhs = torch.from_numpy(g["rhs0"]).clone() + O.laplacian_rhs_adjust(x, mesh, bcs)
    _eq(rhs, g["rhs_set_eq"], "rhs_set_eq")
    coeff, sign = case.get("coeff", 1.0), case.get("sign", 1.0)
    _eq(O.Aop(x, [O.OTerm("laplacian", tabs, coeff, sign)], nd), g["aop"], "aop")
    lap = O.apply_laplacian(tabs, x, nd)
    _eq(lap, g["lap"], "lap")
    O.edge_laplacian(lap, x, mesh)
    _eq(lap, g["lap_edge"], "lap_edge")
    gr = O.apply_grad(O.grad_tables(x, mesh, bcs), x, nd)
    _eq(gr, g["grad"], "grad")
    O.edge_grad(gr, x, mesh)
    _eq(gr, g["grad_edge"], "grad_edge")
    _eq(O.grad_rhs_adjust(x, mesh, bcs), g["grad_rhs_adj"], "grad_rhs_adj")
    ut = torch.from_numpy(g["u_tensor"])
    u = case.get("u", 1.5)
    if "div_none_f" in g:
        _eq(O.apply_div(O.div_tables(u, x, mesh, bcs, "none"), x, nd), g["div_none_f"], "div_none_f")
        _eq(O.apply_div(O.div_tables(ut, x, mesh, bcs, "none"), x, nd), g["div_none_t"], "div_none_t")
    _eq(O.apply_div(O.div_tables(u, x, mesh, bcs, "upwind"), x, nd), g["div_upwind_f"], "div_upwind_f")
    _eq(O.apply_div(O.div_tables(ut, x, mesh, bcs, "upwind"), x, nd), g["div_upwind_t"], "div_upwind_t")


@pytest.mark.parametrize("case", golden_cases("solve"), ids=lambda c: c["name"])
def test_oracle_solve(case):
    g = golden_load(case["name"])
    mesh = _mesh(case)
    rtol = 1e-13 if case["dtype"] == "double" else 1e-6
    for K in case["max_its"]:
        rep_ref = g["_reports"][str(K)]
        x, rep = O.solve_poisson(mesh, _cfg(case), torch.from_numpy(g["rhs0"]).clone(),
                                 method=case["method"], tol=case["tol"], max_it=K,
                                 coeff=case.get("coeff", 1.0), sign=case.get("sign", 1.0))
        assert rep["itr"] == rep_ref["itr"]
        assert rep["converge"] == rep_ref["converge"]
        ref = torch.from_numpy(g[f"x_K{K}"]).double()
        err = float(torch.linalg.norm(x.double() - ref)) / max(float(torch.linalg.norm(ref)), 1e-300)
        assert err <= rtol, (case["name"], K, err)


@pytest.mark.parametrize("case", golden_cases("euler"), ids=lambda c: c["name"])
def test_oracle_euler_pieces(case):
    """Explicit Euler steps composed of reference operators (make_golden.run_euler): the oracle's pieces in
    the same composition are bit-exact; for the central scheme that composition IS O.euler_step."""
    g = golden_load(case["name"])
    mesh = _mesh(case)
    nd = mesh.dim
    bcs = O.make_bcs(mesh, _cfg(case))
    S = O.interior_slicer(nd, bcs)
    nu, dt = case["nu"], case["dt"]
    ut = torch.from_numpy(g["u_tensor"])
    for tag in ("compat_f", "compat_t", "none_f", "none_t"):
        if f"{tag}_s1" not in g:
            continue
        lim = "upwind" if tag.startswith("compat") else "none"
        u = case["u"] if tag.endswith("_f") else ut
        phi = torch.from_numpy(g["phi0"]).clone()
        O.bc_fill(phi, bcs)
        alt = phi.clone()
        for step in range(1, max(case["steps"]) + 1):
            lap = O.apply_laplacian(O.laplacian_tables(phi, mesh, bcs), phi, nd)
            adv = O.apply_div(O.div_tables(u, phi, mesh, bcs, lim), phi, nd)
            new = phi.clone()
            new[0][S] = phi[0][S] + dt * (nu * lap[0][S] - adv[0][S])
            O.bc_fill(new, bcs)
            phi = new
            if lim == "none":
                alt = O.euler_step(alt, u, nu, dt, mesh, bcs, "none")
                _eq(alt, phi, f"{tag} O.euler_step, step {step}")
            if step in case["steps"]:
                _eq(phi, g[f"{tag}_s{step}"], f"{tag} step {step}")


def test_known_answers():
    """Iteration counts recorded by the reference's demo notebook / survey probes
    (demos/poisson_equations/pure_dirichlet.ipynb:107-108)."""
    g = golden_load("cg2d_poisson_n100_f64")["_reports"]["1000"]
    assert g["itr"] == 210 and abs(g["tol"] - 9.661285603057063e-07) < 1e-15
    assert golden_load("cg2d_poisson128_f64")["_reports"]["1000"]["itr"] == 271
    assert golden_load("cg3d_mix33_f64")["_reports"]["1000"]["itr"] == 402
    r = golden_load("cg3d_poisson_dx01_f64")["_reports"]["1000"]
    assert r["itr"] == 2 and abs(r["tol"] - 2.5114204125896697e-17) < 1e-25
    # the reference's own axisymmetric test (tests/test_solver.py:309-358) at its size, run here: 321 its
    r = golden_load("rz_bicg_poisson101_f64")["_reports"]["1000"]
    assert r["itr"] == 321 and r["converge"] and abs(r["tol"] - 5.595559978087123e-06) < 1e-16


def test_rz_reference_test_assertion():
    """tests/test_solver.py:355-358: the axisymmetric solution matches exp(-z) cos(r) to 1e-3"""
    case = [c for c in golden_cases("solve") if c["name"] == "rz_bicg_poisson101_f64"][0]
    g = golden_load(case["name"])
    mesh = _mesh(case)
    torch.testing.assert_close(torch.from_numpy(g["x_K1000"])[0], O.poisson_rz_exact(mesh), atol=1e-3, rtol=1e-3)
    assert torch.equal(torch.from_numpy(g["rhs0"]), O.poisson_rz_rhs(mesh))


def test_rz_periodic_faces_raise_in_the_solver():
    """mesh/tools.py:11-13 looks the face letter up in the xyz table: rz + periodic cannot be solved"""
    mesh = O.OMesh([0.0, 0.0], [1.0, 1.0], [8, 8], "double", "rz")
    cfg = O.mixed_cfg([0.0, 0.0, None, None], ["neumann", "dirichlet", "periodic", "periodic"], O.FACES_RZ)
    with pytest.raises(IndexError):
        O.solve_poisson(mesh, cfg, torch.zeros(1, 8, 8, dtype=torch.float64), method="bicgstab")


def test_reference_csv_fixture():
    """tests/test_solver.py:91-151 of the reference: BiCGSTAB heat conduction vs its CSV."""
    case = [c for c in golden_cases("solve") if c["name"] == "bicg2d_heat_f64"][0]
    g = golden_load(case["name"])
    ref = golden_load("ref_heat_10x10")["sol"]
    mesh = _mesh(case)
    x, rep = O.solve_poisson(mesh, _cfg(case), torch.from_numpy(g["rhs0"]).clone(), method="bicgstab",
                             tol=case["tol"], max_it=1000, coeff=None)
    np.testing.assert_allclose(x[0][:-1, :-1].numpy(), ref, atol=0.01, rtol=0.01)


def test_jacobi_reaches_cg_solution():
    """[NEW] Jacobi has no reference: converged Jacobi == reference CG solution to solver tol."""
    mesh = O.OMesh([0.0, 0.0], [1.0, 1.0], [17, 17], "double")
    rhs = O.poisson_rhs(mesh)
    xj, rj = O.solve_poisson(mesh, O.poisson_cfg(2), rhs.clone(), method="jacobi", tol=1e-10, max_it=20000)
    xc, rc = O.solve_poisson(mesh, O.poisson_cfg(2), rhs.clone(), method="cg", tol=1e-12, max_it=1000)
    assert rj["converge"] and rc["converge"]
    assert float((xj - xc).abs().max()) < 1e-7


def test_upwind_intended_closed_form():
    """[NEW] intended upwind: u (phi_i - phi_{i-1}) / dx for u>0 (reference tests/test_fdm.py:239)."""
    mesh = O.OMesh([0.0], [1.0], [11], "double")
    phi = torch.zeros(1, 11, dtype=torch.float64)
    phi[0] = mesh.grid[0] ** 2
    out = O.div_upwind_intended(2.0, phi, mesh)
    expect = 2.0 * (phi[0][1:-1] - phi[0][:-2]) / mesh.dx[0]
    assert torch.allclose(out[0][1:-1], expect, rtol=0, atol=1e-14)


def test_reference_algorithm_is_summation_order_sensitive():
    """Documents WHY long BiCGSTAB / periodic-CG runs cannot be pinned to 1e-10: the reference
    algorithm itself moves by far more than that when only the order of its dot-product
    summations changes (same inputs, same arithmetic otherwise).  SPD CG is stable."""
    from helpers import summation_band
    from conftest import golden_cases, golden_load
    cases = {c["name"]: c for c in golden_cases("solve")}
    b_bicg, its_bicg = summation_band(cases["bicg2d_xper_f64"], golden_load("bicg2d_xper_f64")["rhs0"], 1000)
    b_pcg, _ = summation_band(cases["cg2d_xper101_f64"], golden_load("cg2d_xper101_f64")["rhs0"], 30)
    b_cg, its_cg = summation_band(cases["cg3d_mix33_f64"], golden_load("cg3d_mix33_f64")["rhs0"], 1000)
    assert b_bicg > 1e-6 and len(set(its_bicg)) > 1
    assert b_pcg > 1e-6
    assert b_cg < 1e-13 and len(set(its_cg)) == 1


def test_committed_hull_is_what_the_oracle_gives():
    """tests/golden/hulls.npz holds the summation-order hulls the GPU suite grades long BiCGSTAB / periodic-CG runs
    against (tests/golden/make_hulls.py: 29 oracle solves per case, minutes of CPU time in all).  The cheapest one is
    recomputed here on every CPU run: a fixture that no longer is what the oracle gives -- the oracle, the inputs or
    helpers.summation_hull changed -- fails here, not silently on the GPU box."""
    import os
    import sys
    import numpy as np
    from conftest import GOLDEN, golden_cases, golden_load
    from helpers import summation_hull
    sys.path.insert(0, GOLDEN)
    from make_hulls import unpack
    z = np.load(os.path.join(GOLDEN, "hulls.npz"))
    cases = {c["name"]: c for c in golden_cases("solve") if c.get("sensitive")}
    # every long run of a sensitive case is in the fixture
    for name, c in cases.items():
        for K in c["max_its"]:
            if K > 10:
                assert f"{name}|K{K}|band" in z.files, (name, K)
    name, K = "bicg2d_heat_f64", 1000        # 11 x 11 nodes: sums this short do not depend on the host's thread count
    c, g = cases[name], golden_load(name)
    hs = []
    band, diam, its = summation_hull(c, g["rhs0"], K, g[f"x_K{K}"], histories=hs)
    fband, fdiam, fits, fhs = unpack(z, f"{name}|K{K}")
    assert its == fits
    assert abs(band - fband) <= 1e-12 * max(band, 1e-300) and abs(diam - fdiam) <= 1e-12 * max(diam, 1e-300)
    assert len(hs) == len(fhs) == 29
    for a, b in zip(hs, fhs):
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-13, atol=0, equal_nan=True)


def _rfp_inputs(case, g):
    mesh = _mesh(case)
    t = lambda k: torch.from_numpy(g[k])   # noqa: E731
    return mesh, t("pdf"), t("H"), t("G"), t("ut")


@pytest.mark.parametrize("case", golden_cases("rfp"), ids=lambda c: c["name"])
def test_oracle_general_div_diffflux_rfp_bit_exact(case):
    """general Div (Jac advection / vector target / edge in n-D), DiffFlux and the rz Fokker-Planck
    operators: oracle == reference outputs, bit for bit"""
    g = golden_load(case["name"])
    mesh, pdf, H, G, ut = _rfp_inputs(case, g)
    nd = mesh.dim
    jo, ho = O.jacobian(H[0], mesh), O.hessian(G[0], mesh)
    fo = O.diff_flux(ho, pdf[0], mesh)
    _eq(fo, g["flux"], "diffFlux")
    for lim in ("none", "upwind"):
        for edge in (True, False):
            tag = f"{lim}_{'edge' if edge else 'noedge'}"
            e = lambda va: (mesh, va) if edge else None   # noqa: E731
            _eq(O.apply_div(O.div_tables(jo, pdf, mesh, [], lim), pdf, nd, e(jo)), g[f"div_jac_{tag}"], "div_jac_" + tag)
            _eq(O.apply_div(O.div_tables(1.0, fo, mesh, [], lim), fo, nd, e(1.0)), g[f"div_vec_f_{tag}"], "div_vec_f_" + tag)
            _eq(O.apply_div(O.div_tables(ut, fo, mesh, [], lim), fo, nd, e(ut)), g[f"div_vec_t_{tag}"], "div_vec_t_" + tag)
    if case.get("coord") == "rz":
        _eq(O.rfp_friction(jo, pdf[0], mesh), g["friction"], "friction")
        _eq(O.rfp_diffusion(ho, pdf[0], mesh), g["diffusion"], "diffusion")
        _eq(O.mc_limiter(torch.from_numpy(g["mc_a"]), torch.from_numpy(g["mc_b"])), g["mc"], "mc_limiter")


def test_edge_div_scalar_target_raises_like_the_reference():
    mesh = O.OMesh([0.0, 0.0], [1.0, 1.0], [6, 7], "double")
    v = torch.rand(1, 6, 7, dtype=torch.float64)
    for adv in (1.5, torch.rand(1, 6, 7, dtype=torch.float64)):
        with pytest.raises(IndexError):
            O.apply_div(O.div_tables(adv, v, mesh, [], "none"), v, 2, (mesh, adv))


@pytest.mark.parametrize("case", [c for c in golden_cases("solve") if c.get("sensitive") and c["dtype"] == "double"],
                         ids=lambda c: c["name"])
def test_oracle_scalar_history_is_the_references(case):
    """The fixtures of the summation-order-sensitive cases hold every dot product and stop-test value of the
    REFERENCE's own run (make_golden.run_solve).  The oracle, summing in torch's order like the reference, must
    reproduce that history -- alpha / beta (CG), alpha / omega / rho_next (BiCGSTAB) -- for as long as the
    reference algorithm is reproducible at all (first 12 iterations: 1e-10), which pins the oracle's LOOP, not
    only its end result; tests/test_gpu_parity_golden.py::test_scalar_history_vs_reference holds the HIP solvers
    to the same record."""
    import numpy as np
    from helpers import SumTap, oracle_solve, scalar_history
    g = golden_load(case["name"])
    K = max(case["max_its"])
    ref = scalar_history(case["method"], list(g[f"hist_sums_K{K}"]))
    tap = SumTap()
    _, rep = oracle_solve(case, g["rhs0"], K, 0, tap=tap)
    mine = scalar_history(case["method"], tap.vals)
    assert rep["itr"] == g["_reports"][str(K)]["itr"] == len(ref)
    n = min(12, len(ref))
    assert np.allclose(mine[:n], ref[:n], rtol=1e-10, atol=0, equal_nan=True), (mine[:n], ref[:n])
    per_it = 1 if case["method"] == "cg" else 2
    assert len(g[f"hist_tol_K{K}"]) in (per_it * len(ref), per_it * len(ref) - 1)
