"""pytest configuration: markers and shared helpers."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the oracle is test infrastructure: importable from tests only
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases(kind=None):
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        cases = json.load(f)
    return [c for c in cases if kind is None or c["kind"] == kind]


def golden_load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {k: z[k] for k in z.files}
    if "_reports" in out:
        out["_reports"] = json.loads(bytes(out["_reports"]).decode())
    return out


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


# Modules written around the launch-per-phase solver loops (tiled vs generic kernels, folded scalar steps, BC-fill
# kernel variants ...): on the small meshes they use, the resident solver (pa_resident.hip) would run instead and
# the comparison would be of that path with itself.  They switch it off; tests/test_gpu_resident.py pins the
# resident path, tests/test_gpu_parity_golden.py runs the reference goldens through both, and the restated
# reference tests / demos / rz suites run the default (resident where it applies).
LAUNCH_PER_PHASE_MODULES = {
    "test_gpu_fold", "test_gpu_fastpath", "test_gpu_fuzz", "test_gpu_tiled_2d", "test_gpu_tiled_advdiff",
    "test_gpu_tiled_bicgstab", "test_gpu_tiled_ops", "test_gpu_bc_fused", "test_gpu_bc_pair", "test_gpu_more_ops",
    "test_gpu_properties",
}


@pytest.fixture(autouse=True)
def _solver_loop_under_test(request, monkeypatch):
    if request.module.__name__.split(".")[-1] in LAUNCH_PER_PHASE_MODULES:
        monkeypatch.setenv("PYAPES_HIP_RESIDENT", "0")
    yield
