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
