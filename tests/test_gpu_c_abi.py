"""-m gpu: the C ABI used from plain C (examples/c_abi_poisson.c: gcc, no Python / PyTorch in the
process) -- the boundary of include/pyapes_hip.h is language-neutral.  Known answers: the reference's
3-D Poisson test stops after 2 CG iterations on the discrete eigen-solution; the 33^3 mixed
Dirichlet / Neumann problem needs the reference's 402 iterations."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_program_against_known_answers(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    exe = str(tmp_path / "c_abi_poisson")
    lib = os.path.join(ROOT, "pyapes_amd", "lib")
    cmd = [gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "c_abi_poisson.c"),
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L" + lib, "-lpyapes_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    m1 = re.search(r"poisson11: itr (\d+) converge (\d) tol (\S+) max\|x - rhs/lambda_h\| (\S+)", out.stdout)
    m2 = re.search(r"mixed33: itr (\d+) converge (\d) tol (\S+) x\[16,16,16\] (\S+)", out.stdout)
    assert m1 and m2, out.stdout
    assert int(m1.group(1)) == 2 and m1.group(2) == "1" and float(m1.group(4)) < 1e-12, out.stdout
    assert int(m2.group(1)) == 402 and m2.group(2) == "1" and float(m2.group(3)) <= 1e-10, out.stdout
