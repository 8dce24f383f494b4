"""CPU: the C-ABI library loads and exports every symbol include/pyapes_hip.h declares; the
pyapes-compatible host layer (mesh, fields, BC factories, equation DSL) behaves like the
reference's; compute entry points refuse to run without the GPU (no CPU fallback)."""
import os
import re
import warnings

import pytest
import torch

import pyapes_oracle as O
from conftest import ROOT
from pyapes_amd.geometry import Box
from pyapes_amd.hip import lib as L
from pyapes_amd.mesh import Mesh
from pyapes_amd.mesh.tools import boundary_slicer, inner_slicer
from pyapes_amd.solver.fdc import FDC
from pyapes_amd.solver.fdm import FDM
from pyapes_amd.solver.linalg import terms_of
from pyapes_amd.solver.ops import Solver
from pyapes_amd.testing.poisson import poisson_bcs, poisson_rhs_nd
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import BoxBoundary, homogeneous_bcs, mixed_bcs


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pyapes_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = L.load_library()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pyapes_hip.h but not exported"
    assert set(names) == set(L.SIGNATURES), set(names) ^ set(L.SIGNATURES)
    assert b"gfx950" in lib.pa_version()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_ctx_create_fails_loudly_without_gpu():
    import ctypes as C
    lib = L.load_library()
    h = C.c_void_p()
    rc = lib.pa_ctx_create(0, None, C.byref(h))
    assert rc == L.PA_E_HIP and b"no HIP device" in lib.pa_last_error(None)


def test_missing_library_is_a_hard_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load_library(str(tmp_path / "nope.so"))


@pytest.mark.parametrize("spacing", [[11, 17, 9], [0.1, 0.05, 0.125]])
@pytest.mark.parametrize("dtype", ["double", "single"])
def test_mesh_matches_reference_conventions(spacing, dtype):
    mesh = Mesh(Box[0:1, 0:2, -1:0.5], None, spacing, "cpu", dtype)
    om = O.OMesh([0, 0, -1], [1, 2, 0.5], spacing, dtype)   # oracle mesh is pinned to the reference
    assert list(mesh.nx) == om.nx
    assert mesh.dx_list == om.dx_list
    for a in range(3):
        assert torch.equal(mesh.x[a], om.x[a])
        assert torch.equal(mesh.grid[a], om.grid[a])
    assert mesh.dim == 3 and mesh.coord_sys == "xyz" and mesh.N == om.nx[0] * om.nx[1] * om.nx[2]
    assert torch.equal(mesh.d_mask["yu"], om.face_mask("yu"))
    assert float(mesh.face_dxf("xl")) == float(om.x[0][0] - om.x[0][1])
    assert float(mesh.face_dxf("zu")) == float(om.x[2][-1] - om.x[2][-2])


def test_box_face_order_and_geometry():
    assert Box[0:1].face == ["xl", "xu"]
    assert Box[0:1, 0:1].face == ["yl", "yu", "xl", "xu"]          # geometry/basis.py:163-180
    assert Box[0:1, 0:1, 0:1].face == ["xl", "xu", "yl", "yu", "zl", "zu"]
    b = Box([0, 0], [2, 3])
    assert b.dim == 2 and b.size == 6.0 and b.type == "box" and b.lower == [0.0, 0.0]
    with pytest.raises(AssertionError):
        Box[0:1:2]


def test_bc_factories_and_field():
    assert [c["bc_face"] for c in homogeneous_bcs(2, 0.0, "dirichlet")] == ["xl", "xu", "yl", "yu"]
    m = mixed_bcs([0, 1, None, None], ["dirichlet", "neumann", "periodic", "periodic"])
    assert m[1] == {"bc_face": "xu", "bc_type": "neumann", "bc_val": 1, "bc_val_opt": None}
    bb = BoxBoundary(xl={"bc_type": "dirichlet", "bc_val": 0.4}, yu={"bc_type": "symmetry", "bc_val": None})()
    assert [c["bc_face"] for c in bb] == ["xl", "yu"]
    mesh = Mesh(Box[0:1, 0:1], None, [5, 6], "cpu", "double")
    var = Field("p", 1, mesh, {"domain": m, "obstacle": None}, init_val=0.5)
    assert var().shape == (1, 5, 6) and float(var()[0, 2, 3]) == 0.5
    assert [bc.bc_type for bc in var.bcs] == ["dirichlet", "neumann", "periodic", "periodic"]
    assert var.bcs[1].bc_n_dir == 1 and var.bcs[1].bc_face_dim == 0 and var.bcs[1].bc_treat
    assert torch.equal(var.bcs[0].bc_mask_prev, torch.roll(mesh.d_mask["xl"], 1, 0))
    assert boundary_slicer(2, var.bcs) == [slice(1, -1), slice(None, None)]
    assert inner_slicer(2, 2) == [slice(2, -2), slice(2, -2)]
    with pytest.raises(AssertionError):
        Field("q", 1, mesh, {"domain": m[:3], "obstacle": None})
    w = var.copy("w")
    w += 1.0
    assert float(var()[0, 0, 0]) == 0.5 and float(w()[0, 0, 0]) == 1.5 and w.name == "w"
    var <<= torch.ones(1, 5, 6, dtype=torch.float64)
    assert float(var().sum()) == 30.0


def test_equation_dsl_mirrors_reference():
    mesh = Mesh(Box[0:1], None, [11], "cpu", "double")
    var = Field("U", 1, mesh, {"domain": homogeneous_bcs(1, 0.0, "dirichlet"), "obstacle": None}, init_val=0.5)
    fdm = FDM({"div": {"limiter": "upwind", "edge": False}})
    eq = fdm.grad(var) - fdm.laplacian(0.5, var) == 1.0
    assert [eq.ops[k]["name"] for k in eq.ops] == ["Grad", "Laplacian"]
    assert eq.ops[0]["sign"] == 1.0 and eq.ops[1]["sign"] == -1 and eq.ops[1]["param"] == (0.5,)
    assert eq.rhs.shape == var().shape and float(eq.rhs[0, 3]) == 1.0
    assert set(eq.ops[0]) == {"name", "Aop", "target", "param", "sign", "other", "A_coeffs", "adjust_rhs"}
    terms, bcs = terms_of(eq.ops)
    assert [t["kind"] for t in terms] == [L.OP_GRAD, L.OP_LAPLACIAN] and terms[1]["sign"] == -1
    neg = -FDM().laplacian(var)
    assert neg.ops[0]["sign"] == -1 and neg.ops[0]["param"] == (None,)
    d = fdm.div(2.0, var)
    t, _ = terms_of(d.ops)
    assert t[0]["kind"] == L.OP_DIV_UPWIND and t[0]["u"] == 2.0
    fdm2 = FDM({"div": {"limiter": "upwind", "edge": False, "compat": True}})
    assert terms_of(fdm2.div(var).ops)[0][0]["kind"] == L.OP_DIV_UPWIND_COMPAT
    with pytest.raises(AssertionError):
        FDM().laplacian(1.0, var) == torch.zeros(2, 11)
    # operator objects are per instance (reference shares class-level singletons)
    assert FDM().laplacian is not FDM().laplacian


def test_no_cpu_compute_path():
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, [0.1, 0.1, 0.1], "cpu", "double")
    var = Field("p", 1, mesh, {"domain": poisson_bcs(3), "obstacle": None})
    rhs = poisson_rhs_nd(mesh, var)
    solver = Solver({"fdm": {"method": "cg", "tol": 1e-6, "max_it": 10, "report": False}})
    with pytest.raises(RuntimeError, match="no CPU"):
        solver.set_eq(FDM().laplacian(1.0, var) == rhs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="no CPU"):
            FDC({"laplacian": {"edge": False}}).laplacian(var)
    with pytest.raises(RuntimeError, match="no CPU"):
        var.apply_bcs()


def test_product_does_not_import_the_oracle():
    import pathlib
    for p in pathlib.Path(ROOT, "pyapes_amd").rglob("*.py"):
        assert "pyapes_oracle" not in p.read_text(), p


# ---- behaviours the reference's own tests pin (tests/test_variables.py, tests/test_mesh.py) -------
@pytest.mark.parametrize("nd", [1, 2, 3])
def test_shifted_bc_masks(nd):
    """reference tests/test_variables.py::test_field_bc_mask_individual"""
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, [5] * nd, "cpu", "double")
    var = Field("t", 1, mesh, {"domain": homogeneous_bcs(nd, 0.0, "dirichlet"), "obstacle": None})
    for i, bc in enumerate(var.bcs):
        n_dir = -1 if i % 2 == 0 else 1
        m = bc.bc_mask
        assert torch.equal(bc.bc_mask_prev, torch.roll(m, -n_dir, i // 2))
        assert torch.equal(bc.bc_mask_prev2, torch.roll(m, -2 * n_dir, i // 2))
        assert torch.equal(bc.bc_mask_forward, torch.roll(m, n_dir, i // 2))
        assert torch.equal(bc.bc_mask_forward2, torch.roll(m, 2 * n_dir, i // 2))
        assert int(m.sum()) == 5 ** (nd - 1)


def test_box_boundary_config_dicts():
    """reference tests/test_variables.py::test_bc_config (Box part)"""
    cfg = BoxBoundary(xl={"bc_type": "dirichlet", "bc_val": 0.44}, xu={"bc_type": "neumann", "bc_val": 0},
                      yl={"bc_type": "periodic", "bc_val": None}, yu={"bc_type": "symmetry", "bc_val": None})()
    assert cfg == [{"bc_face": "xl", "bc_type": "dirichlet", "bc_val": 0.44, "bc_val_opt": None},
                   {"bc_face": "xu", "bc_type": "neumann", "bc_val": 0, "bc_val_opt": None},
                   {"bc_face": "yl", "bc_type": "periodic", "bc_val": None, "bc_val_opt": None},
                   {"bc_face": "yu", "bc_type": "symmetry", "bc_val": None, "bc_val_opt": None}]


@pytest.mark.parametrize("nd", [1, 2, 3])
def test_field_arithmetic(nd):
    """reference tests/test_variables.py::test_fields"""
    mesh = Mesh(Box([0.0] * nd, [1.0] * nd), None, [0.1] * nd, "cpu", "double")
    var = Field("any", 1, mesh, {"domain": None, "obstacle": None})
    t = torch.rand(*var.size, dtype=torch.float64)
    var += t
    assert torch.equal(var(), t)
    var /= var
    assert torch.allclose(var(), torch.ones_like(t))
    var *= 10
    assert torch.allclose(var(), torch.full_like(t, 10.0))
    var -= var
    assert torch.equal(var(), torch.zeros_like(t))
    var += 2.5
    c = var.copy()
    assert torch.equal(c(), torch.full_like(t, 2.5)) and c() is not var()
    assert float(var.zeros_like(name="z")().abs().max()) == 0.0 and var.zeros_like(name="z").name == "z"
    assert var.copy(name="cp").name == "cp"


def test_mesh_masks_and_geometry():
    """reference tests/test_mesh.py::test_basic_mask (no-obstacle part), ::test_geometries (Box part)"""
    mesh = Mesh(Box[0:1, 0:1], None, [0.1, 0.1], "cpu", "single")
    assert torch.allclose(mesh.dg[0][0].mean(), mesh.dx[0] / 2)
    tm = mesh.t_mask
    assert torch.equal(tm[:, 0], tm[0]) and torch.equal(tm[:, -1], tm[0]) and torch.equal(tm[-1, :], tm[0])
    assert bool(tm[0].all()) and not bool(tm[1:-1, 1:-1].any())
    for i, box in enumerate([Box[0:2], Box[0:2, 0:2], Box[0:2, 0:2, 0:2]]):
        assert box.type == "box" and box.dim == i + 1 and box.size == pytest.approx(2 ** (i + 1))
        assert box.lower == [0] * (i + 1) and box.upper == [2] * (i + 1)


def test_mesh_d_mask_shift_is_the_references_literal_roll():
    """_mesh.py:138-175: ``d_mask_dir`` compares the side letter with "r" (faces are named l / u), so the roll is +shift on
    both sides -- a lower face's mask moves one node inwards, an upper face's mask wraps to the low end."""
    mesh = Mesh(Box[0:1, 0:2], None, [5, 7], "cpu", "double")
    assert mesh.d_mask_dir("xl") == -1 and mesh.d_mask_dir("xu") == -1 and mesh.d_mask_dim("yl") == 1
    m = mesh.d_mask_shift("xl", 1)
    assert bool(m[1].all()) and int(m.sum()) == 7
    m = mesh.d_mask_shift("xu", 1)
    assert bool(m[0].all()) and int(m.sum()) == 7           # wrapped: the reference's literal behaviour
    m = mesh.d_mask_shift("yl", 2)
    assert bool(m[:, 2].all()) and int(m.sum()) == 5


def test_install_as_pyapes_aliases_are_the_same_modules():
    """unmodified pyapes imports (current layout and the notebooks' pyapes.core.* layout) resolve to
    this package's own module objects"""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r)
import pyapes_amd
pyapes_amd.install_as_pyapes()
from pyapes.geometry import Box, Cylinder
from pyapes.mesh import Mesh
from pyapes.solver.fdm import FDM
from pyapes.solver.ops import Solver
from pyapes.variables import Field
from pyapes.variables.bcs import CylinderBoundary, homogeneous_bcs
from pyapes.core.solver.fdm import FDM as FDM2
from pyapes.core.variables.bcs import homogeneous_bcs as hb2
from pyapes.testing.poisson import poisson_bcs
import pyapes_amd.solver.fdm as real
assert FDM is real.FDM and FDM2 is real.FDM and hb2 is homogeneous_bcs
assert sys.modules["pyapes.solver.fdm"] is real
m = Mesh(Box[0:1, 0:1], None, [5, 5])            # host logic works without a GPU
f = Field("p", 1, m, {"domain": poisson_bcs(2), "obstacle": None})
assert f().shape == (1, 5, 5)
print("ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_cylinder_mesh_host_logic():
    """axisymmetric containers are host logic: faces, axis numbering, coordinate accessors (no GPU)"""
    from pyapes_amd.geometry import Cylinder
    from pyapes_amd.variables.bcs import CylinderBoundary
    cyl = Cylinder[0:2, -1:3]
    assert cyl.type == "cylinder" and cyl.dim == 2 and cyl.lower == [0.0, -1.0] and cyl.upper == [2.0, 3.0]
    assert [f["face"] for f in cyl.config.values()] == ["zl", "zu", "rl", "ru"]     # reference order (2-D: second axis first)
    with pytest.raises(AssertionError):
        Cylinder([-1.0, 0.0], [1.0, 1.0])
    mesh = Mesh(cyl, None, [5, 9], "cpu", "double")
    assert mesh.coord_sys == "rz" and mesh.R is mesh.grid[0] and mesh.Z is mesh.grid[1] and mesh.Y.numel() == 0
    assert [mesh.face_index(f) for f in ("rl", "ru", "zl", "zu")] == [0, 1, 2, 3]
    assert mesh.d_mask["ru"][-1].all() and not mesh.d_mask["ru"][:-1].any()
    bc = CylinderBoundary(rl={"bc_type": "neumann", "bc_val": 0.0}, ru={"bc_type": "dirichlet", "bc_val": 1.0},
                          zl={"bc_type": "symmetry", "bc_val": None}, zu={"bc_type": "periodic", "bc_val": None})()
    var = Field("u", 1, mesh, {"domain": bc, "obstacle": None})
    assert [(b.bc_face, b.bc_face_dim, b.bc_n_dir, b.bc_type) for b in var.bcs] == [
        ("rl", 0, -1, "neumann"), ("ru", 0, 1, "dirichlet"), ("zl", 1, -1, "symmetry"), ("zu", 1, 1, "periodic")]
    with pytest.raises(IndexError):          # the reference's slicer looks 'z' up in the xyz table: axis 2 of a 2-D mesh
        boundary_slicer(2, var.bcs)
    with pytest.raises(KeyError):            # Box meshes do not know R
        Mesh(Box[0:1, 0:1], None, [4, 4]).R


def test_div_operand_resolution_mirrors_the_reference_indexing():
    """fdc._div_plan: which tensor multiplies which axis (fdc.py:93-102, 292-311, 708-792), host logic only"""
    from pyapes_amd.solver.fdc import _div_plan
    from pyapes_amd.variables.container import Jac
    mesh = Mesh(Box[0:1, 0:1], None, [4, 5])
    sc = Field("s", 1, mesh, {"domain": None, "obstacle": None}, init_val="random")
    vec = Field("v", 2, mesh, {"domain": None, "obstacle": None}, init_val="random")
    jac = Jac(x=torch.rand(4, 5), y=torch.rand(4, 5))
    x, u, ui, ue = _div_plan(jac, sc, True)                      # scalar target: Jac.x inside, Jac[axis] on the edges
    assert all(t is sc()[0] or torch.equal(t, sc()[0]) for t in x) and ui[0] is jac.x and ui[1] is jac.x
    assert ue[0] is jac.x and ue[1] is jac.y
    x, u, ui, ue = _div_plan(1.5, vec, True)                     # vector target, float speed
    assert torch.equal(x[1], vec()[1]) and u == 1.5 and ui == [None, None] and ue == [None, None]
    t = torch.rand(2, 4, 5)
    x, u, ui, ue = _div_plan(t, vec, False)
    assert torch.equal(ui[1], t[1]) and ue == [None, None]
    for adv in (1.5, torch.rand(1, 4, 5), sc.copy()):            # scalar target + edge: the reference indexes [1] of a size-1 axis
        with pytest.raises(IndexError):
            _div_plan(adv, sc, True)


def test_bc_callable_probe_and_varo_flag_host_side():
    """Round 2 host logic (no GPU): a callable bc_val that reads the field is told apart from one of
    (grid, mask) only -- the device solvers refuse the former (linalg._run) --, Dirichlet / Neumann without a
    value assert like the reference (bcs.py:200, 224), and Field.VARo raises once marked stale and comes
    back with save_old() (fields.py:129-136)."""
    mesh = Mesh(Box[0:1, 0:1], None, [6, 7], "cpu", "double")
    reads = lambda grid, mask, var, opt: 0.5 * var[0][torch.roll(mask, 1, 0)]   # noqa: E731
    plain = lambda grid, mask, var, opt: grid[1][mask] * 2.0                     # noqa: E731
    cfg = [{"bc_face": f, "bc_type": "dirichlet", "bc_val": v, "bc_val_opt": None}
           for f, v in zip(("xl", "xu", "yl", "yu"), (reads, plain, 0.25, torch.ones(6)))]
    var = Field("p", 1, mesh, {"domain": cfg, "obstacle": None}, init_val="random")
    assert [bc.depends_on_var(var()) for bc in var.bcs] == [True, False, False, False]
    s, arr = var.bcs[1].resolve(var(), 0)
    assert arr is not None and torch.equal(arr, mesh.grid[1][var.bcs[1].bc_mask] * 2.0)
    none_cfg = [{"bc_face": f, "bc_type": t, "bc_val": None, "bc_val_opt": None}
                for f, t in zip(("xl", "xu", "yl", "yu"), ("neumann", "symmetry", "periodic", "periodic"))]
    v2 = Field("q", 1, mesh, {"domain": none_cfg, "obstacle": None})
    with pytest.raises(AssertionError, match="bc_val is not specified"):
        v2.bcs[0].resolve(v2(), 0)
    assert v2.bcs[1].resolve(v2(), 0) == (0.0, None)
    var.save_old()
    assert torch.equal(var.VARo, var())
    var.mark_old_stale("test")
    with pytest.raises(RuntimeError, match="VARo"):
        var.VARo
    var.save_old()
    assert torch.equal(var.VARo, var())
