"""-m gpu: the explicit Euler march in its "BC on load" form (csrc/pa_sf_kernel.h, BCL): no BC-fill launch between
the steps -- every step forms the face values its stencil reads from its own operands -- and one ordered fill after
the last step.  Must be BIT-IDENTICAL to the step-by-step sequence (step kernel + ordered fill, every step), which
tests/test_gpu_parity_golden.py::test_euler_steps_vs_reference_pieces pins against steps composed of the
REFERENCE's own operators and BC.apply."""
import pytest
import torch

from helpers import bit_equal
from pyapes_amd.geometry import Box
from pyapes_amd.hip.context import context_for
from pyapes_amd.mesh import Mesh
from pyapes_amd.solver.march import euler_march, euler_step
from pyapes_amd.variables import Field
from pyapes_amd.variables.bcs import mixed_bcs

pytestmark = pytest.mark.gpu

NEUSYM = ([0.0, 0.0, None, None, None, None], ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
ALLNEU = ([0.3, -0.2, 0.1, 0.0, -0.4, 0.25], ["neumann"] * 6)
MIXED = ([0.5, 0.1, None, 1.0, -0.3, None], ["dirichlet", "neumann", "symmetry", "dirichlet", "neumann", "symmetry"])
ALLDIR = ([0.0, 1.0, 0.25, -0.5, 2.0, 0.0], ["dirichlet"] * 6)
CASES = [
    ("config4_family_f32", [40, 36, 72], "single", NEUSYM, 1.0, 7),
    ("all_neumann_f64", [24, 20, 66], "double", ALLNEU, -0.8, 6),       # negative speed: the other upwind branch
    ("mixed_f64", [21, 19, 34], "double", MIXED, 0.6, 5),
    ("all_dirichlet_f32", [18, 22, 132], "single", ALLDIR, 1.3, 6),
    ("speed_field_f32", [20, 24, 64], "single", NEUSYM, "field", 6),
    ("two_row_waves_f64", [80, 6, 32], "double", MIXED, 1.0, 5),        # n1 = 6: two rows per wave
]


def _march(n, dtype, bcs, u, steps, bcl, phi0, ufield):
    mesh = Mesh(Box[0:1, 0:1, 0:1], None, n, "cuda", dtype)
    context_for(mesh).set_option("bcl", bcl)
    phi = Field("phi", 1, mesh, {"domain": mixed_bcs(*bcs), "obstacle": None})
    phi.set_var_tensor(phi0.to(mesh.dtype.float).cuda())
    phi.apply_bcs()
    uu = ufield.to(mesh.dtype.float).cuda() if u == "field" else u
    dx = float(mesh.dx_list[0])
    nu = 1e-3
    dt = 0.2 * min(dx * dx / (6 * nu), dx / 1.3)
    euler_march(phi, uu, nu, dt, steps, {"div": {"limiter": "upwind"}})
    return phi().clone(), mesh, (nu, dt, uu)


@pytest.mark.parametrize("name,n,dtype,bcs,u,steps", CASES, ids=[c[0] for c in CASES])
def test_bc_on_load_march_is_bit_identical_to_the_step_by_step_sequence(name, n, dtype, bcs, u, steps):
    g = torch.Generator().manual_seed(9)
    phi0 = torch.rand((1, *n), generator=g, dtype=torch.float64)
    ufield = torch.randn((1, *n), generator=g, dtype=torch.float64)
    a, _, _ = _march(n, dtype, bcs, u, steps, True, phi0, ufield)
    b, mesh, (nu, dt, uu) = _march(n, dtype, bcs, u, steps, False, phi0, ufield)
    assert bit_equal(a, b), float((a - b).abs().max())
    # ... and to single steps (each with its own ordered fill)
    phi = Field("phi", 1, mesh, {"domain": mixed_bcs(*bcs), "obstacle": None})
    phi.set_var_tensor(phi0.to(mesh.dtype.float).cuda())
    phi.apply_bcs()
    for _ in range(steps):
        euler_step(phi, uu, nu, dt, {"div": {"limiter": "upwind"}})
    assert bit_equal(a, phi())


def test_bc_on_load_is_really_taken_and_declines_where_it_must(capfd):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch\nfrom test_gpu_bcl import _march, NEUSYM\n"
            "n = [40, 36, 72]; p = torch.rand((1, *n), dtype=torch.float64)\n"
            "_march(n, 'single', NEUSYM, 1.0, 3, True, p, p)\n"
            "per = ([None] * 6, ['periodic', 'periodic', 'symmetry', 'symmetry', 'symmetry', 'symmetry'])\n"
            "_march(n, 'single', per, 1.0, 3, True, p, p)\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ, PYAPES_HIP_DEBUG="1", PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle"), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if "k_sf phase 3" in ln]
    assert sum("(BC on load)" in ln for ln in lines) == 3, r.stderr[-2000:]       # the three steps of the first march
    assert sum("(BC on load)" not in ln for ln in lines) >= 3                      # a periodic face: the classic sequence
