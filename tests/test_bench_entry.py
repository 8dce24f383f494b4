"""bench.py as the driver runs it: ``python bench.py --gpus N`` must start its N ranks itself (no
external torchrun), relay ONE JSON line and return the worst child exit code.  The GPU tests run the
real kernels with every rank on cuda:0 (BENCH_SINGLE_DEVICE=1; planes through the host with gloo, or a
1-rank RCCL communicator for the library-side loop); the CPU test checks the launcher mechanics only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)
    return p


def test_self_launch_without_gpu_fails_loudly_and_returns():
    """No GPU here: the ranks must die on their own assertion (there is no CPU path), the parent must
    relay a non-zero exit code and must not hang or print a JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = _run(["--gpus", "2", "--size", "12,8,8", "--steps", "2", "--warmup", "1"], {}, timeout=300)
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
    assert "needs the MI355X" in p.stderr or "GPUs visible" in p.stderr


@pytest.mark.gpu
def test_self_launch_two_ranks_one_gpu():
    """the driver's command shape, N = 2, through the NEW entry path (parent -> torch.distributed.run -> ranks)"""
    p = _run(["--gpus", "2", "--size", "48,40,136", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             {"BENCH_SINGLE_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 2
    assert rec["config"]["parallelism"].startswith("slab2")
    assert rec["value"] > 0 and rec["roofline"]["kernel"].startswith("cg_phase_")


@pytest.mark.gpu
def test_slab_bench_library_side_rccl_one_rank():
    """BENCH_FORCE_SLAB: the slab path with the library-side RCCL loop on a 1-rank communicator;
    n_gpus is what RCCL saw and the parallelism string says where the collectives ran"""
    p = _run(["--size", "48,40,136", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             {"BENCH_FORCE_SLAB": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29731", "RANK": "0",
              "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert rec["n_gpus"] == 1
    assert rec["config"]["parallelism"] == "slab1 (rccl-in-library)"


@pytest.mark.gpu
@pytest.mark.parametrize("wl,size", [("c4", "40,36,72"), ("c4t", "40,36,72"), ("c1", "64,64"), ("c2", "40,36,72"),
                                     ("c5", "40,36,72")])
def test_every_workload_runs(wl, size):
    p = _run(["--workload", wl, "--size", size, "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], {})
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert rec["steps"] == 6 and rec["value"] > 0 and rec["n_gpus"] == 1
    assert "roofline" in rec and rec["roofline"]["kernel_ms"] > 0
