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


def test_launcher_budget_and_first_attempt_record(monkeypatch, capsys):
    """launch_ranks on the host only (the attempts are stand-ins): a first attempt that times out is followed by
    ONE attempt on the stepwise driver with what is left of the budget -- both inside the driver's 600 s -- and the
    relayed record carries how the first attempt ended; a refused configuration is not retried."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []
    clock = [0.0]

    def fake_run(n, argv, env, tmo):
        calls.append((dict(env), tmo))
        if env.get("PYAPES_HIP_COMM") == "0":
            clock[0] += 50.0
            return 0, json.dumps({"metric": "m", "value": 1.0, "config": {"parallelism": "slab2 (torch.distributed-stepwise/nccl)"}})
        clock[0] += tmo       # the library-side attempt hangs until its limit
        return None, None

    monkeypatch.setattr(bench, "_run_ranks", fake_run)
    monkeypatch.setattr(bench.time, "monotonic", lambda: clock[0])
    monkeypatch.delenv("PYAPES_HIP_COMM", raising=False)
    monkeypatch.delenv("BENCH_RANKS_TIMEOUT", raising=False)
    monkeypatch.delenv("BENCH_RANKS_BUDGET", raising=False)
    rc = bench.launch_ranks(8, ["--gpus", "8"])
    assert rc == 0 and len(calls) == 2
    assert calls[0][1] + calls[1][1] <= 560 and clock[0] < 600      # both attempts fit the driver's limit
    assert calls[1][0]["PYAPES_HIP_COMM"] == "0"
    rec = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rec["first_attempt"]["outcome"].startswith("timeout after")

    # a rank that refuses the configuration: one attempt, no retry, non-zero exit
    calls.clear()

    def fake_refuse(n, argv, env, tmo):
        calls.append(env)
        with open(env["BENCH_STATUS_FILE"], "a") as f:
            f.write("config\n")
        return 1, None

    monkeypatch.setattr(bench, "_run_ranks", fake_refuse)
    assert bench.launch_ranks(8, ["--gpus", "8"]) == 1 and len(calls) == 1


HOSTRING = {"BENCH_SINGLE_DEVICE": "1", "BENCH_BACKEND": "gloo",
            "BENCH_COMM_LIB": os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpa_hostring.so")}


@pytest.mark.gpu
def test_two_ranks_library_side_loop_through_the_entry_path():
    """python bench.py --gpus 2 with the library-side loop REALLY running between two ranks (they share cuda:0, so
    the wire is the test stand-in of tests/lib/pa_hostring.hip, handed over by bench.py's rehearsal switch BENCH_COMM_LIB; everything else is what runs over RCCL)"""
    p = _run(["--gpus", "2", "--size", "48,40,136", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], HOSTRING)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    # (the record names the implementation behind the library-side loop: here the stand-in, never "rccl")
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"].startswith("slab2 (library loop over [custom: hostring")
    assert "first_attempt" not in rec


@pytest.mark.gpu
def test_hung_collective_in_the_warm_up_falls_back_inside_the_run():
    """Rank 1's first row all-reduce of the folded loop never completes (injected).  Both ranks must notice within
    BENCH_WARMUP_TIMEOUT, abort the library's communicators, agree over the host and finish on the stepwise
    driver IN THE SAME PROCESSES -- the case of the driver's own torch.distributed.run launch, which has no parent
    to retry -- and the record must say so."""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--size", "48,40,136", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             dict(HOSTRING, PYAPES_HIP_HOSTRING_FAIL="1:hang:1", BENCH_WARMUP_TIMEOUT="4", PYAPES_HIP_HOSTRING_TIMEOUT="9"))
    assert p.returncode == 0, p.stderr[-3000:]
    assert time.time() - t0 < 300
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert rec["config"]["parallelism"].startswith("slab2 (torch.distributed-stepwise")
    assert "timeout after 4 s in the warm-up" in rec["first_attempt"]["outcome"]
    assert rec["value"] > 0 and rec["steps"] == 4


@pytest.mark.gpu
def test_failed_first_attempt_is_retried_once_by_the_launcher():
    """The second line of defence: the ranks' own watchdog is out of the way (BENCH_WARMUP_TIMEOUT far away) and a
    collective of the first attempt fails for good (the stand-in gives up on it after 5 s and refuses further
    work, as RCCL does after an asynchronous error): the ranks die with an error, the launching parent runs ONE
    attempt on the stepwise driver and the record carries how the first attempt ended.  (That a HUNG attempt is
    stopped after BENCH_RANKS_TIMEOUT and both attempts fit 600 s is pinned on the host by
    test_launcher_budget_and_first_attempt_record: killing ranks in the middle of a GPU wait is not something to
    rehearse on a shared box.)"""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--size", "48,40,136", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
             dict(HOSTRING, PYAPES_HIP_HOSTRING_FAIL="1:hang:1", BENCH_WARMUP_TIMEOUT="900", PYAPES_HIP_HOSTRING_TIMEOUT="5",
                  BENCH_RANKS_TIMEOUT="200", BENCH_RANKS_BUDGET="400"))
    assert p.returncode == 0, p.stderr[-3000:]
    assert time.time() - t0 < 400
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert rec["first_attempt"]["outcome"].startswith("rc="), rec["first_attempt"]
    assert rec["config"]["parallelism"].startswith("slab2 (torch.distributed-stepwise")
