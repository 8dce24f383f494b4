"""Build-quality gate: no kernel of libpyapes_hip may carry a private segment (scratch memory).

Measured on MI355X: a kernel with a private segment costs ~10 us more per dispatch (64^3 CG iteration
29 -> 52 us when the phase-A kernel had 64 bytes of it), which is most of an iteration on the meshes
pyapes users run.  The usual cause is a by-value kernel-argument struct that ends up indexed at run
time ("scratch memory").  Reads the gfx950 code object out of the built library with the
LLVM tools of the ROCm image; CPU-only.
"""
import os
import re
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(__file__), "..", "pyapes_amd", "lib", "libpyapes_hip.so")


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="ROCm LLVM tools not present")
def test_no_kernel_uses_scratch_memory():
    assert os.path.exists(LIB), "libpyapes_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", LIB,
                        os.path.join(d, "unused.so")], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"     # one bundle per translation unit, back to back
        starts = [m.start() for m in re.finditer(magic, blob)]
        assert len(starts) >= 4
        notes = ""
        for q, a in enumerate(starts):
            piece = os.path.join(d, f"bundle{q}.bin")
            with open(piece, "wb") as fh:
                fh.write(blob[a:starts[q + 1] if q + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                           check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                                    capture_output=True, text=True).stdout
    names = re.findall(r"\.name:\s+(\S+)", notes)
    sizes = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
    assert len(names) == len(sizes) == len(spills) and len(names) > 300
    offenders = [(n, s) for n, s in zip(names, sizes) if s != 0]
    assert not offenders, f"kernels with a private segment: {offenders[:5]}"
    assert not any(spills), "a kernel spills vector registers"
