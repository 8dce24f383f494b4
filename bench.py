#!/usr/bin/env python3
"""bench.py -- headline benchmark: cell-updates x iterations / second of the hot path on the
BASELINE.json configurations, plus the HBM roofline of the dominant kernel and the CPU baseline
timed on the host cores of the same box.

A "step" is ONE pass of the hot path over the whole grid: a CG iteration (stencil apply + AXPYs +
two dot reductions + BC fill + stop test) for c2 / c3 / c5, an explicit Euler step (stencil + BC
fill) for c4, a Jacobi sweep for c1.  Default workload at every N: BASELINE config 3, the
configuration the metric is quoted on -- 3-D Poisson 512^3 fp64, periodic BCs, CG -- slab-
decomposed along axis 0 for N > 1 (strong scaling: the global grid is fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5|c4|c4t|c4_512|c1|c3b|c3j]
                    [--size n0,n1,n2]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
``python -m torch.distributed.run`` on 127.0.0.1, one rank per GPU; the parent never touches the GPU),
relays rank 0's JSON line and returns the child's exit code.  Under an external
``torch.distributed.run`` each process is one rank.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md:36, :296)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 streaming copy reaches on this part (same guide: 6.29 TB/s measured)
# algorithmic HBM bytes per cell per step in array passes
ALG_PASSES_CG = 10
ALG_PASSES_PHASE = {"a": 4, "b": 6}

WORKLOADS = {
    # name: (solver, global nodes, dtype, bc kind, box upper, BASELINE config)
    "c3": ("cg", (512, 512, 512), "double", "periodic", (1.0, 1.0, 1.0), 3),
    "c2": ("cg", (256, 256, 256), "double", "dirichlet", (1.0, 1.0, 1.0), 2),
    "c5": ("cg", (1024, 1024, 512), "single", "mixed", (1.0, 1.0, 0.5), 5),
    # explicit adv-diff march, upwind Div + Laplacian, Neumann(x) / Symmetry(y, z): scalar u (2 passes),
    # speed tensor (3 passes), and a size that does not sit in the 256 MiB Infinity Cache
    "c4": ("euler", (256, 256, 256), "single", "neusym", (1.0, 1.0, 1.0), 4),
    "c4t": ("euler_t", (256, 256, 256), "single", "neusym", (1.0, 1.0, 1.0), 4),
    "c4_512": ("euler", (512, 512, 512), "single", "neusym", (1.0, 1.0, 1.0), 4),
    "c1": ("jacobi", (128, 128), "double", "poisson2d", (1.0, 1.0), 1),
    # config 3's mesh under the other two solver loops (round 4): BiCGSTAB is the method that CONVERGES on the periodic
    # problem (linalg.py:162-279), Jacobi the sweep of BASELINE config 1 at a size where bytes count
    "c3b": ("bicgstab", (512, 512, 512), "double", "periodic", (1.0, 1.0, 1.0), 3),
    "c3j": ("jacobi3", (512, 512, 512), "double", "periodic", (1.0, 1.0, 1.0), 3),
}
ALG_PASSES_ONESHOT = {"bicgstab": 22, "jacobi3": 3}   # array passes per iteration, algorithmic count


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
#  N > 1 without an external launcher: the parent starts the ranks (and never touches the GPU)
# ------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(n: int, argv: list[str], extra_env: dict, timeout_s: float):
    """One attempt: (return code or None on timeout, JSON line or None)."""
    import signal
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BENCH_SELF_LAUNCHED"] = "1"
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    log(f"starting {n} ranks: {' '.join(cmd[1:])}" + (f"  [{extra_env}]" if extra_env else ""))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        log(f"no result within {timeout_s:.0f} s: stopping the ranks (process group {p.pid})")
        try:
            os.killpg(p.pid, signal.SIGTERM)     # exactly the group this call started
            p.communicate(timeout=20)
        except Exception:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except Exception:
                pass
            p.communicate()
        return None, None
    line = None
    for ln in out.splitlines():
        st = ln.strip()
        if st.startswith("{") and '"metric"' in st:
            line = st
        elif st:
            print(st, file=sys.stderr)
    return p.returncode, line


EXIT_CONFIG = 3   # a rank refused the configuration (too few GPUs, single-GPU workload ...): running it again cannot help


def refuse_config(msg: str):
    """A rank that cannot run the requested configuration at all: say so on stderr, tell the launching parent
    (torch.distributed.run flattens every child exit code to 1, so through a file it named) and leave."""
    log(msg)
    path = os.environ.get("BENCH_STATUS_FILE")
    if path:
        try:
            with open(path, "a") as f:
                f.write("config\n")
        except OSError:
            pass
    raise SystemExit(EXIT_CONFIG)


def launch_ranks(n: int, argv: list[str]) -> int:
    """Start ``python -m torch.distributed.run --nproc-per-node n bench.py <argv>`` as a CHILD process,
    pass its stderr through, print the one JSON line rank 0 wrote, return the child's exit code.
    Nothing here imports torch.cuda or calls HIP: a process that has initialised the GPU must not be
    replaced or forked into ranks.

    Watchdog, sized for the driver's 600 s limit on the whole command: the first attempt gets
    BENCH_RANKS_TIMEOUT seconds (default 240); if it gives no result or fails -- and did not refuse the
    configuration -- ONE more attempt runs on the stepwise torch.distributed slab driver (PYAPES_HIP_COMM=0)
    with what is left of BENCH_RANKS_BUDGET (default 540 s for both).  A record produced by the second attempt
    says so: ``first_attempt`` = how the first one ended.  (The ranks bound their own first iterations of the
    library-side loop too -- BENCH_WARMUP_TIMEOUT in main() -- so this is the second line of defence.)"""
    import tempfile
    budget = float(os.environ.get("BENCH_RANKS_BUDGET", "540"))
    tmo = min(float(os.environ.get("BENCH_RANKS_TIMEOUT", "240")), budget)
    t0 = time.monotonic()
    fd, status = tempfile.mkstemp(prefix="bench_status_")
    os.close(fd)
    rc, line = _run_ranks(n, argv, {"BENCH_STATUS_FILE": status}, tmo)
    try:
        with open(status) as f:
            refused = "config" in f.read()
        os.unlink(status)
    except OSError:
        refused = False
    first = None
    if refused:
        log("a rank refused the configuration: not retried")
    elif (rc != 0 or line is None) and os.environ.get("PYAPES_HIP_COMM", "1") != "0":
        first = f"timeout after {tmo:.0f} s" if rc is None else (f"rc={rc}" if rc != 0 else "rc=0 without a JSON line")
        left = budget - (time.monotonic() - t0)
        if left >= 30:
            log(f"first attempt ended with {first}; ONE more on the stepwise driver (PYAPES_HIP_COMM=0), {left:.0f} s left")
            rc, line = _run_ranks(n, argv, {"PYAPES_HIP_COMM": "0"}, left)
        else:
            log(f"first attempt ended with {first}; {left:.0f} s left are too few for another")
    if line is not None:
        if first is not None:
            try:
                rec = json.loads(line)
                rec["first_attempt"] = {"path": "library-side RCCL loop (pa_cg_iterate_comm)", "outcome": first,
                                        "then": "all ranks restarted on the stepwise torch.distributed driver"}
                line = json.dumps(rec)
            except ValueError:
                pass
        print(line, flush=True)
    elif rc == 0:
        log("the ranks exited 0 but printed no JSON line")
        return 1
    return 1 if rc is None else rc


# ------------------------------------------------------------------------------------------------
#  synthetic inputs
# ------------------------------------------------------------------------------------------------
def make_bcs(kind):
    from pyapes_amd.variables.bcs import homogeneous_bcs, mixed_bcs
    if kind == "periodic":
        return homogeneous_bcs(3, None, "periodic")
    if kind == "dirichlet":
        return homogeneous_bcs(3, 0.0, "dirichlet")
    if kind == "neusym":
        return mixed_bcs([0.0, 0.0, None, None, None, None],
                         ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
    if kind == "poisson2d":
        from pyapes_amd.testing.poisson import poisson_bcs
        return poisson_bcs(2)
    return mixed_bcs([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])


def oracle_cfg(O, kind):
    if kind == "periodic":
        return O.homogeneous_cfg(3, None, "periodic")
    if kind == "dirichlet":
        return O.homogeneous_cfg(3, 0.0, "dirichlet")
    if kind == "neusym":
        return O.mixed_cfg([0.0, 0.0, None, None, None, None],
                           ["neumann", "neumann", "symmetry", "symmetry", "symmetry", "symmetry"])
    if kind == "poisson2d":
        return O.poisson_cfg(2)
    return O.mixed_cfg([0, 0, 0, 0, 1, 0], ["dirichlet", "neumann", "dirichlet", "neumann", "dirichlet", "neumann"])


def synth_rhs(gn, i_off, n0, kind, fdtype, dev):
    """Deterministic synthetic right-hand side as a function of the GLOBAL node index, so every
    decomposition (and the CPU baseline) sees the same problem family.  periodic: ring-mean-zero
    product of sines plus a zero-mean deterministic 'noise' (SURVEY 8d C3); otherwise
    sin(pi x) sin(pi y) sin(pi z) + noise."""
    import torch
    idx = [torch.arange(i_off, i_off + n0, device=dev, dtype=torch.float64),
           torch.arange(gn[1], device=dev, dtype=torch.float64),
           torch.arange(gn[2], device=dev, dtype=torch.float64)]
    I, J, K = torch.meshgrid(idx, indexing="ij")
    if kind == "periodic":
        base = torch.sin(2 * math.pi * I / gn[0]) * torch.sin(2 * math.pi * J / gn[1]) * torch.sin(2 * math.pi * K / gn[2])
        # zero-mean on the ring in every axis: cos of an integer number of periods
        noise = 0.25 * torch.cos(2 * math.pi * (3 * I / gn[0] + 5 * J / gn[1] + 7 * K / gn[2]))
    else:
        base = torch.sin(math.pi * I / (gn[0] - 1)) * torch.sin(math.pi * J / (gn[1] - 1)) * torch.sin(math.pi * K / (gn[2] - 1))
        noise = 0.25 * torch.sin(12.9898 * I + 78.233 * J + 37.719 * K)
    return (base + noise).to(fdtype).unsqueeze(0).contiguous()


def gaussian(X, Y, Z):
    import torch
    return torch.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) / 0.02)


def euler_params(dx, u=1.0, nu=1e-3):
    return nu, 0.2 * min(dx * dx / (6 * nu), dx / abs(u))


def host_mem_available() -> int:
    """Bytes of host memory this process may still take: MemAvailable, capped by the cgroup limit."""
    avail = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except Exception:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, max(0, int(lim) - cur)) if avail else max(0, int(lim) - cur)
    except Exception:
        pass
    return avail


def host_cores() -> int:
    """CPUs this process may actually use: min(os.cpu_count(), affinity, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


# ------------------------------------------------------------------------------------------------
#  clocks / power while the timed region runs (evidence for box-to-box spread: DESIGN.md "Measurement")
# ------------------------------------------------------------------------------------------------
class ClockSampler:
    """Samples the GPU's engine / memory / fabric clocks and socket power from sysfs every 10 ms on a host thread
    while the timed region runs (pp_dpm_* mark the active level with '*'; hwmon has the power).  Best effort: a box
    that hides these files yields an empty record.  Reported in the JSON line as ``clocks`` and on stderr."""

    def __init__(self, index: int = 0):
        import glob
        self.files = {}
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        self.card = None
        if cards:
            # sysfs shows every GPU of the host, the process sees its own: pick the card by PCI address (a box of the
            # pool lists eight cards; the first one is somebody else's GPU)
            dev = None
            try:
                import torch
                pr = torch.cuda.get_device_properties(index)
                bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
                for cpath in cards:
                    real = os.path.realpath(os.path.dirname(cpath))
                    if bdf in real.lower():
                        dev = os.path.dirname(cpath)
                        self.card = os.path.basename(os.path.dirname(dev)) + " " + os.path.basename(real)
                        break
            except Exception:
                dev = None
            if dev is None:
                dev = os.path.dirname(cards[min(index, len(cards) - 1)])
                self.card = "unmatched:" + os.path.basename(os.path.dirname(dev))
            for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
                if os.path.exists(os.path.join(dev, name)):
                    self.files[key] = os.path.join(dev, name)
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                for name in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(hw, name)):
                        self.files.setdefault("power_w", os.path.join(hw, name))
        self.samples = {k: [] for k in self.files}
        self._stop = False
        self._thread = None

    def _read(self, key, path):
        try:
            txt = open(path).read()
        except OSError:
            return None
        if key == "power_w":
            try:
                return int(txt.strip()) / 1e6
            except ValueError:
                return None
        for ln in txt.splitlines():
            if "*" in ln:
                try:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except (IndexError, ValueError):
                    return None
        return None

    def _run(self):
        while not self._stop:
            for k, pth in self.files.items():
                v = self._read(k, pth)
                if v is not None:
                    self.samples[k].append(v)
            time.sleep(0.01)

    def __enter__(self):
        import threading
        if self.files:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def record(self):
        out = {}
        if getattr(self, "card", None):
            out["card"] = self.card
        for k, v in self.samples.items():
            if v:
                out[k] = {"min": min(v), "avg": sum(v) / len(v), "max": max(v), "n": len(v)}
        return out


# ------------------------------------------------------------------------------------------------
#  CPU baseline: the oracle (checker code) timed on this box's host cores -- reported, never measured path
# ------------------------------------------------------------------------------------------------
def cpu_baseline(solver, kind, dtype, gn):
    """The oracle (literal torch-CPU restatement of the reference algorithm) on a BOUNDED sample of the
    same workload family (same BCs, same synthetic input formula, smaller grid where the full one would
    take minutes), on every host core the box grants."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyapes_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    fd = torch.float64 if dtype == "double" else torch.float32
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if solver in ("cg", "bicgstab", "jacobi3"):
            n = [min(256, gn[0]), min(256, gn[1]), min(256, gn[2] if kind != "mixed" else 128)]
            its = 24 if solver != "bicgstab" else 12
            why = ""
            cells_full = gn[0] * gn[1] * gn[2]
            if list(n) != list(gn):
                # The literal algorithm holds ~34 arrays (15 coefficient tables, rolled copies, r / d / Ad / x / x_old):
                # 34 GiB at 512^3 fp64.  Run the FULL size when the host has the memory for it and an iteration is
                # short enough for a bounded sample (3 iterations, ~10 s each on 16 threads); else say why not.
                need = 34 * cells_full * (8 if dtype == "double" else 4)
                avail = host_mem_available()
                per_it = cells_full / (1.4e7 * max(1, cores) / 16.0)
                if solver == "cg" and os.environ.get("BENCH_CPU_FULL", "1") != "0" and avail >= 1.5 * need and 3 * per_it <= 45.0:
                    n, its = list(gn), 3
                else:
                    why = (f" (full size {'x'.join(map(str, gn))} not run: needs ~{need / 2**30:.0f} GiB of host memory "
                           f"[{avail / 2**30:.0f} GiB available] and ~{per_it:.0f} s per iteration)")
            up = [1.0, 1.0, 0.5 if kind == "mixed" else 1.0]
            mesh = O.OMesh([0, 0, 0], up, n, dtype)
            bcs = O.make_bcs(mesh, oracle_cfg(O, kind))
            rhs = synth_rhs(n, 0, n[0], kind, fd, "cpu")
            x = torch.zeros(1, *n, dtype=mesh.dtype)
            tabs = O.laplacian_tables(x, mesh, bcs)
            rhs += O.laplacian_rhs_adjust(x, mesh, bcs)
            terms = [O.OTerm("laplacian", tabs, 1.0, 1.0)]
            log(f"cpu_baseline: oracle {solver} {n} on {cores} host threads ...")
            t0 = time.perf_counter()
            if solver == "cg":
                _, rep = O.cg(x, rhs, terms, mesh, bcs, -1.0, its - 1)
            elif solver == "bicgstab":
                _, rep = O.bicgstab(x, rhs, terms, mesh, bcs, -1.0, its)
            else:
                _, rep = O.jacobi(x, rhs, terms, mesh, bcs, -1.0, its - 1, 1.0)
            dt = time.perf_counter() - t0
            steps, what = rep["itr"], {"cg": "CG iterations", "bicgstab": "BiCGSTAB iterations", "jacobi3": "Jacobi sweeps"}[solver] + why
        elif solver in ("euler", "euler_t"):
            n = [min(256, v) for v in gn]
            mesh = O.OMesh([0, 0, 0], [1, 1, 1], n, dtype)
            bcs = O.make_bcs(mesh, oracle_cfg(O, kind))
            phi = gaussian(*mesh.grid).to(fd).unsqueeze(0).contiguous()
            O.bc_fill(phi, bcs)
            u = 1.0 if solver == "euler" else torch.full_like(phi, 0.7)
            nu, dts = euler_params(float(mesh.dx[0]))
            steps, what = 6, "explicit Euler steps (upwind Div + Laplacian + BC fill)"
            log(f"cpu_baseline: oracle Euler march {n} on {cores} host threads ...")
            t0 = time.perf_counter()
            for _ in range(steps):
                phi = O.euler_step(phi, u, nu, dts, mesh, bcs, "upwind")
            dt = time.perf_counter() - t0
        else:  # jacobi, config 1: the reference-runnable plumbing case, at its own size
            n = list(gn)
            mesh = O.OMesh([0, 0], [1, 1], n, dtype)
            steps, what = 2000, "Jacobi sweeps"
            rhs = O.poisson_rhs(mesh)
            log(f"cpu_baseline: oracle Jacobi {n} on {cores} host threads ...")
            t0 = time.perf_counter()
            _, rep = O.solve_poisson(mesh, O.poisson_cfg(2), rhs, method="jacobi", tol=-1.0, max_it=steps - 1)
            dt = time.perf_counter() - t0
            steps = rep["itr"]
    cells = 1
    for v in n:
        cells *= v
    try:
        import resource
        log(f"cpu_baseline: done in {dt:.1f} s; peak host RSS of this process {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GiB")
    except Exception:
        pass
    return {"value": cells * steps / dt, "unit": "cell-updates*iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU literal restatement of the reference algorithm), {'x'.join(map(str, n))} {dtype} "
                      f"{kind} BCs, same synthetic input family as the GPU leg, {steps} {what}, {dt:.1f} s"}


# ------------------------------------------------------------------------------------------------
#  roofline record
# ------------------------------------------------------------------------------------------------
def source_hash(csrc_dir: str | None = None) -> str:
    """Hash of the kernel sources the built library comes from -- comments and white space stripped, so that
    only a change of CODE invalidates what was measured on them: profiles/traffic.json entries carry it and
    are ignored once the kernels have changed."""
    import re
    h = hashlib.sha256()
    d = csrc_dir or os.path.join(ROOT, "pyapes_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name)) as f:
                txt = f.read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            txt = re.sub(r"//[^\n]*", "", txt)
            txt = re.sub(r"\s+", " ", txt)
            h.update(name.encode())
            h.update(txt.encode())
    return h.hexdigest()[:16]


def roofline(kernel, ms, alg_bytes, wl_key, custom_size, extra=None):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "frac_traffic": None,
            "achievable": HBM_ACHIEVABLE_GBS, "achievable_source": "MI355X_MICROARCH.md: 6.29 TB/s float4 copy (79 % of spec)",
            "kernel_ms": ms, "alg_bytes_per_launch": alg_bytes,
            "note": "achieved / frac count ALGORITHMIC bytes (SURVEY 8d); frac_traffic = PMC HBM bytes per launch / "
                    "kernel_ms / peak is what the memory system really moved"}
    if extra:
        roof.update(extra)
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr) and not custom_size:
        try:
            with open(tr) as f:
                t = json.load(f)
            e = t.get(f"{wl_key}:{kernel}")
            if e and e.get("source_hash") == source_hash():
                roof["traffic"] = e["bytes_per_launch"]
                roof["traffic_source"] = e.get("source")
                roof["frac_traffic"] = e["bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            elif e:
                roof["traffic_source"] = "profiles/traffic.json entry is from other kernel sources (hash mismatch): dropped"
        except Exception:
            pass
    return roof


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--size", dest="n", default=None, help="override global node counts n0,n1[,n2] (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-probe", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (before anything can touch the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch

    # stdout carries exactly ONE line, the JSON record: anything native libraries print on file
    # descriptor 1 in the meantime (RCCL prints a version banner when a communicator comes up) goes
    # to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    solver, gn, dtype, kind, upper, cfg_no = WORKLOADS[args.workload]
    if args.n:
        gn = tuple(int(v) for v in args.n.split(","))
        assert len(gn) == len(upper), "--size must have the workload's dimension"
    if world > 1 and solver != "cg":
        refuse_config(f"workload {args.workload} is single-GPU (replicas only); the slab path is the CG solve")

    # rehearsal switches (one-GPU box): BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0,
    # BENCH_BACKEND=gloo moves the planes through the host instead of RCCL, BENCH_COMM_LIB=<path> hands the library a
    # stand-in for librccl (tests/lib/libpa_hostring.so: pa_comm_use_impl, an explicit call -- the record's
    # config.parallelism names the implementation)
    single_dev = bool(os.environ.get("BENCH_SINGLE_DEVICE"))
    if world > 1 and not single_dev and torch.cuda.device_count() < world:
        refuse_config(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPUs visible")
    if not torch.cuda.is_available():
        refuse_config("bench.py needs the MI355X; there is no CPU path")
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from pyapes_amd.geometry import Box
    from pyapes_amd.mesh import Mesh
    from pyapes_amd.hip import lib as L
    from pyapes_amd.hip.context import context_for
    from pyapes_amd.variables import Field
    if os.environ.get("BENCH_COMM_LIB"):
        rc_impl = L.load_library().pa_comm_use_impl(os.environ["BENCH_COMM_LIB"].encode())
        assert rc_impl == 0, f"pa_comm_use_impl({os.environ['BENCH_COMM_LIB']}) -> {rc_impl}"

    esize = 8 if dtype == "double" else 4
    nd = len(gn)
    cells_global = 1
    for v in gn:
        cells_global *= v
    W, K = args.warmup, args.steps

    dist = None
    force_slab = bool(os.environ.get("BENCH_FORCE_SLAB")) and solver == "cg"  # rehearsal: the slab path with one rank
    if world > 1 or force_slab:
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    slab = world > 1 or force_slab
    # agreements that must not depend on the GPU's streams (watchdog of the slab warm-up) go over the host
    host_group = None
    if dist is not None:
        host_group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
    first_attempt = None
    clocks = None
    box = {}            # per-box yardstick (single-GPU CG branch): free device memory, plain copy rate

    mesh = Mesh(Box([0.0] * nd, list(upper)), None, list(gn), "cuda", dtype, slab=(rank, world) if slab else None)
    var = Field("p", 1, mesh, {"domain": make_bcs(kind), "obstacle": None})
    log(f"rank {rank}/{world}: workload {args.workload} global {gn} local {tuple(mesh.nx)} {dtype} {kind}")
    parallelism = "single"
    roof = None
    wl_text = None

    if solver == "cg" and not slab:
        rhs = synth_rhs(gn, 0, gn[0], kind, mesh.dtype.float, mesh.device)
        if os.environ.get("BENCH_FRESH_X"):
            # diagnostic (docs/HISTORY.md "Section 8", slow boxes): x and the right-hand side in blocks the driver hands out NOW,
            # not in blocks the caching allocator recycled from the mesh / input construction
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            var.set_var_tensor(var().clone())
            rhs = rhs.clone()
            torch.cuda.empty_cache()
        terms = [{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}]
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        ctx.set_terms(terms)
        ctx.rhs_adjust(rhs[0])
        # tolerance -1: the stop test is evaluated but can never end the solve; max_it beyond everything run below
        STEADY_MAX = int(os.environ.get("BENCH_STEADY_MAX", "160"))
        if os.environ.get("BENCH_PLACE") is not None:     # A/B switch: 0 = no online placement search in this run
            ctx.set_option("place", int(os.environ["BENCH_PLACE"]))
        # set-up of the solve (pa_cg_begin: scratch allocation, BC fill, first residual): outside the timed region, so
        # its cost goes into the record -- wall clock and stream time
        torch.cuda.synchronize()
        s0 = torch.cuda.Event(enable_timing=True)
        s1 = torch.cuda.Event(enable_timing=True)
        ts0 = time.perf_counter()
        s0.record(ctx.stream)
        ctx.cg_begin(var()[0], rhs[0], -1.0, W + 2 * K + STEADY_MAX + 20)
        s1.record(ctx.stream)
        torch.cuda.synchronize()
        setup = {"wall_ms": (time.perf_counter() - ts0) * 1e3, "stream_ms": s0.elapsed_time(s1),
                 "what": "pa_cg_begin: scratch allocation, BC fill, r = b - A x; nothing else runs before the first iteration"}
        ctx.cg_iterate(W)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        log("array bases: x %#x rhs %#x (ctx-owned r / d buffers: hipMalloc, 2 MiB-aligned)" % (var().data_ptr(), rhs.data_ptr()))
        box = {}
        try:   # how much of the card is free: a box whose GiB-sized allocations run slow is worth a look at this
            free_b, total_b = torch.cuda.mem_get_info()
            log(f"device memory: {free_b / 2**30:.1f} GiB free of {total_b / 2**30:.1f} GiB")
            box["free_gib"] = round(free_b / 2 ** 30, 1)
        except Exception:
            pass
        try:
            # The yardstick of THIS box: a plain device copy of an array of the workload's size (1 read + 1 write).  The CG
            # phases of 512^3 measure 853 us on some boxes and 990 us on others with the same bytes moved (DESIGN.md
            # section 8); the copy says what the box gives a kernel that does nothing else.
            src = var()[0]
            dst = torch.empty_like(src)
            for _ in range(2):
                dst.copy_(src)
            torch.cuda.synchronize()
            c0 = torch.cuda.Event(enable_timing=True)
            c1 = torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            cms = c0.elapsed_time(c1) / 5
            box["copy_ms"] = cms
            box["copy_GBs"] = 2 * src.numel() * src.element_size() / (cms * 1e-3) / 1e9
            log(f"this box: torch copy_ of one {src.numel() * src.element_size() / 2**20:.0f} MiB array {cms * 1e3:.1f} us = {box['copy_GBs']:.0f} GB/s")
            del dst
        except Exception:
            pass
        with ClockSampler(local_rank) as clk:
            t0 = time.perf_counter()
            e0.record(ctx.stream)
            ctx.cg_iterate(K)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
        clocks = clk.record()
        log(f"clocks / power during the timed region: {clocks}")
        ev_ms = e0.elapsed_time(e1)
        rep = ctx.report()
        assert rep.itr == W + K and rep.status == 0, f"work was skipped: itr={rep.itr} status={rep.status}"
        assert bool(torch.isfinite(var()).all()), "iterate became non-finite inside the timed region"
        # The online placement search (csrc/pa_place.hip) rides on the iterations above -- warm-up and timed region
        # alike, its copies and memsets included in `value`.  What it has done so far, and what the SAME solve reaches
        # once its pass is over (at most BENCH_STEADY_MAX more iterations, then K timed ones): `probe`.
        st = ctx.place_stats()
        probe = {"mode": "online (pa_place.hip): trials ride on the solve's own iterations, no set-up cost",
                 "state_after_timed_region": st["state"], "trials": int(st["trials"]), "kept": int(st["kept"])}
        if st["state"] != "off" and STEADY_MAX > 0:
            extra = 0
            while extra < STEADY_MAX and ctx.place_stats()["state"] == "searching":
                ctx.cg_iterate(8)
                torch.cuda.synchronize()
                extra += 8
            q0 = torch.cuda.Event(enable_timing=True)
            q1 = torch.cuda.Event(enable_timing=True)
            q0.record(ctx.stream)
            ctx.cg_iterate(K)
            q1.record(ctx.stream)
            torch.cuda.synchronize()
            st = ctx.place_stats()
            probe.update({"state": st["state"], "trials": int(st["trials"]), "kept": int(st["kept"]),
                          "allocations": int(st["allocations"]), "iterations_until_steady": W + K + extra,
                          "steady_ms_per_step": q0.elapsed_time(q1) / K,
                          "as_allocated_ms_per_step": st["first_pair_us"] / 2e3, "best_ms_per_step": st["best_pair_us"] / 2e3,
                          "spent_ms": st["spent_us"] / 1e3, "timed_ms": st["timed_us"] / 1e3})
            gain = (st["first_pair_us"] - st["best_pair_us"]) / 2.0
            probe["break_even_iterations"] = (st["spent_us"] / gain) if gain > 1.0 else None
            log(f"placement search: {probe}")
        if not args.no_roofline_probe:
            # per-kernel durations of the two dominant kernels, HIP events on the launch stream
            ctx.profile(True)
            ctx.cg_iterate(min(K, 10))
            pr = ctx.profile_read()
            ctx.profile(False)
            dom = "b" if pr["phase_b_ms"] >= pr["phase_a_ms"] else "a"
            roof = roofline(f"cg_phase_{dom}", pr[f"phase_{dom}_ms"], ALG_PASSES_PHASE[dom] * esize * mesh.N,
                            args.workload, bool(args.n),
                            {"phase_a_ms": pr["phase_a_ms"], "phase_b_ms": pr["phase_b_ms"]})
        ctx.cg_end()
        passes = ALG_PASSES_CG
    elif solver == "cg":
        from pyapes_amd.slab import SlabCG
        rhs = synth_rhs(gn, mesh.i_off, mesh.nx[0], kind, mesh.dtype.float, mesh.device)
        terms = [{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}]
        drv = SlabCG(mesh, var, rhs, terms, dist)
        drv.begin(-1.0, W + K + 10)
        drv.iterate(W)
        if drv.lib_comm:
            # The warm-up is the first time the library-side loop's collectives run between THESE ranks: its
            # completion is waited for with a deadline.  A rank that sees none aborts the library's communicators
            # (its stream is released), the ranks agree over the host (gloo: nothing of this may queue behind a
            # stuck stream), and all of them restart on the stepwise torch.distributed driver -- recorded in the line.
            wtmo = float(os.environ.get("BENCH_WARMUP_TIMEOUT", "120"))
            ok_here = drv.be.stream_wait(wtmo)
            if not ok_here:
                log(f"rank {rank}: no completion of the library-side warm-up within {wtmo:.0f} s: aborting its communicators")
                drv.be.comm_abort()
            flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=host_group)
            if int(flag.item()) == 0:
                bad = [None] * world
                dist.all_gather_object(bad, bool(ok_here), group=host_group)
                first_attempt = {"path": "library-side RCCL loop (pa_cg_iterate_comm)",
                                 "outcome": f"timeout after {wtmo:.0f} s in the warm-up iterations on rank(s) "
                                            f"{[q for q, v in enumerate(bad) if not v]}",
                                 "then": "communicators aborted, all ranks restarted on the stepwise torch.distributed driver"}
                if ok_here:
                    drv.be.comm_abort()
                drv.be.cg_abort()
                drv.be.slab_set(None)
                os.environ["PYAPES_HIP_COMM"] = "0"
                drv.be.set_option("comm", 0)          # (the option of THIS context: the variable is only read when one is created)
                var = Field("p", 1, mesh, {"domain": make_bcs(kind), "obstacle": None})
                drv = SlabCG(mesh, var, rhs, terms, dist)
                assert not drv.lib_comm
                drv.begin(-1.0, W + K + 10, adjust_rhs=False)
                drv.iterate(W)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.iterate(K)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        secs_local = time.perf_counter() - t0
        t = torch.tensor([secs_local], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
        ev_ms = secs * 1e3
        rep = drv.be.report()
        assert rep.itr == W + K and rep.status == 0, f"work was skipped: itr={rep.itr} status={rep.status}"
        assert bool(torch.isfinite(var()).all()), "iterate became non-finite inside the timed region"
        if not args.no_roofline_probe:
            pr = drv.profile(min(K, 10))
            dom = "b" if pr["phase_b_ms"] >= pr["phase_a_ms"] else "a"
            roof = roofline(f"cg_phase_{dom}", pr[f"phase_{dom}_ms"], ALG_PASSES_PHASE[dom] * esize * mesh.N,
                            args.workload, True,
                            {"phase_a_ms": pr["phase_a_ms"], "phase_b_ms": pr["phase_b_ms"], "scope": "rank 0, per GPU"})
        drv.end()
        n_seen = dist.get_world_size()
        impl = drv.be.comm_impl() if drv.lib_comm else ""
        comm_kind = (("rccl-in-library" if impl == "rccl" else f"library loop over [{impl}]") if drv.lib_comm
                     else f"torch.distributed-stepwise/{dist.get_backend()}")
        if drv.lib_comm:
            n_seen = drv.be.comm_size()
            torch.cuda.synchronize()      # release the library's communicators while every rank is still alive
            drv.be.comm_destroy()
            drv.be.comm_ready = None
        world = n_seen                      # n_gpus = the ranks the communicator really had
        parallelism = f"slab{n_seen} ({comm_kind})"
        passes = ALG_PASSES_CG
    elif solver in ("euler", "euler_t"):
        from pyapes_amd.solver.fdc import div_kind
        phi = gaussian(mesh.X, mesh.Y, mesh.Z).to(mesh.dtype.float).unsqueeze(0).contiguous()
        var.set_var_tensor(phi)
        var.apply_bcs()
        u = 1.0 if solver == "euler" else torch.full_like(var()[0], 0.7)
        nu, dts = euler_params(mesh.dx_list[0])
        kindc = div_kind("upwind", False)
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        a, b = var()[0], torch.empty_like(var()[0])
        cur = ctx.euler_march(a, b, kindc, u, nu, dts, W)
        oth = b if cur is a else a
        torch.cuda.synchronize()
        try:
            # The yardstick of THIS box for a step's arrays: a plain device copy of one of them into the other (1 read + 1
            # write, the same two allocations, warm).  At 256^3 fp32 both arrays (64 MiB each) sit in the 256 MiB Infinity
            # Cache, so this -- not 8 TB/s of HBM -- is the ceiling a 2-pass kernel can be held against.
            for _ in range(3):
                oth.copy_(cur)
            torch.cuda.synchronize()
            c0 = torch.cuda.Event(enable_timing=True)
            c1 = torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(20):
                oth.copy_(cur)
            c1.record()
            torch.cuda.synchronize()
            cms = c0.elapsed_time(c1) / 20
            box = {"copy_ms": cms, "copy_GBs": 2 * cur.numel() * cur.element_size() / (cms * 1e-3) / 1e9,
                   "what": f"torch copy_ between the step's two {cur.numel() * cur.element_size() / 2**20:.0f} MiB arrays, warm "
                           "(Infinity-Cache resident when they fit 256 MiB)"}
            log(f"this box: copy between the step's arrays {cms * 1e3:.1f} us = {box['copy_GBs']:.0f} GB/s")
        except Exception:
            pass
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ctx.stream)
        fin = ctx.euler_march(cur, oth, kindc, u, nu, dts, K)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        assert bool(torch.isfinite(fin).all()), "state became non-finite inside the timed region"
        passes = 2 if solver == "euler" else 3
        if not args.no_roofline_probe:
            ctx.profile(True)
            ctx.euler_march(fin, cur if fin is oth else oth, kindc, u, nu, dts, min(K, 10))
            pr = ctx.profile_read()
            ctx.profile(False)
            # The probe brackets ONE launch at a time with HIP events and the host waits in between: a launch from an
            # idle stream, which for a 30 us kernel reads ~8 us long.  In the march the steps run back to back -- with BC
            # on load a step IS one launch -- so the stream time per step bounds the kernel's duration from above.
            k_ms = min(pr["phase_a_ms"], ev_ms / K)
            extra = {"kernel_ms_probe": pr["phase_a_ms"], "step_stream_ms": ev_ms / K,
                     "kernel_ms_source": "min(per-launch HIP-event probe, stream time of the timed march / steps)"}
            arr_mib = esize * mesh.N / 2 ** 20
            if (passes + 0) * arr_mib <= 256:
                extra["remark"] = (f"{arr_mib:.0f} MiB per array: the step's arrays sit in the 256 MiB Infinity Cache, and the step "
                                   "carries ~32 VALU instructions per cell that may not be fused (13.8 us of pure issue at 256^3): "
                                   "bound by instruction issue plus one dependent launch per step, not by HBM -- the fraction "
                                   "of the HBM roofline is reported for comparison only (DESIGN.md section 4)")
            if box.get("copy_GBs"):
                # against the copy of the same arrays on this box (cache-resident or not): the honest ceiling of the step
                extra["ceiling_GBs"] = box["copy_GBs"]
                extra["ceiling_source"] = "torch copy_ between the step's own two arrays on this box, same run (box.copy_GBs)"
                extra["frac_of_ceiling"] = (passes * esize * mesh.N / (k_ms * 1e-3) / 1e9) / box["copy_GBs"]
            roof = roofline("euler_step", k_ms, passes * esize * mesh.N, args.workload, bool(args.n), extra)
        wl_text = (f"3-D advection-diffusion {'x'.join(map(str, gn))} {dtype}, Div(upwind)+Laplacian explicit Euler march, "
                   f"{'scalar u' if solver == 'euler' else 'speed tensor u(x)'}, Neumann/Symmetry BCs (BASELINE config 4)")
    elif solver in ("bicgstab", "jacobi3"):
        # one-shot library loops (pa_bicgstab / pa_jacobi) with a fixed iteration count: tolerance -1, the stop test is
        # evaluated every iteration and can never end the solve; the timed region is ONE call of K iterations, set-up
        # (scratch, BC fill, first residual) included -- it is part of what a user's solve() costs
        method = "bicgstab" if solver == "bicgstab" else "jacobi"
        rhs = synth_rhs(gn, 0, gn[0], kind, mesh.dtype.float, mesh.device)
        terms = [{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}]
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        ctx.set_terms(terms)
        ctx.rhs_adjust(rhs[0])
        mi = (lambda n: n) if method == "bicgstab" else (lambda n: n - 1)   # CG / Jacobi run max_it + 1 (linalg.py:144-157)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if W > 0:
                ctx.solve(method, var()[0], rhs[0], -1.0, mi(W))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rep = ctx.solve(method, var()[0], rhs[0], -1.0, mi(K))
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
        ev_ms = rep.gpu_ms
        assert rep.itr == K and rep.status == 0, f"work was skipped: itr={rep.itr} status={rep.status}"
        assert bool(torch.isfinite(var()).all()), "iterate became non-finite inside the timed region"
        passes = ALG_PASSES_ONESHOT[solver]
        moved = {"bicgstab": 15, "jacobi3": 3}[solver]
        roof = roofline(f"{method} iteration (all kernels of the loop: stream time of the iterations / K)", ev_ms / K,
                        passes * esize * mesh.N, args.workload, True,
                        {"remark": f"{passes} array passes per iteration by the algorithmic count, {moved} really moved "
                                   "(DESIGN.md section 4); per-kernel table: profiles/r04_bicgstab512_kernel_stats.csv; "
                                   "frac can exceed 1 on the algorithmic count because fused phases never store "
                                   "what the count assumes stored -- frac_moved prices the passes the loop really makes",
                         "passes_moved": moved,
                         "frac_moved": moved * esize * mesh.N / (ev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS})
        wl_text = (f"3-D Poisson {gn[0]}x{gn[1]}x{gn[2]} {dtype}, {kind} BCs, "
                   f"{'BiCGSTAB' if method == 'bicgstab' else 'Jacobi'} (BASELINE config {cfg_no}'s mesh)")
    else:  # jacobi (config 1)
        from pyapes_amd.testing.poisson import poisson_rhs_nd
        rhs = poisson_rhs_nd(mesh, var)
        terms = [{"kind": L.OP_LAPLACIAN, "sign": 1.0, "coeff": 1.0}]
        ctx = context_for(mesh)
        ctx.bind_bcs(var(), var.bcs, 0)
        ctx.set_terms(terms)
        ctx.rhs_adjust(rhs[0])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if W > 0:
                ctx.solve("jacobi", var()[0], rhs[0], -1.0, W - 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rep = ctx.solve("jacobi", var()[0], rhs[0], -1.0, K - 1)   # max_it = K - 1 -> K sweeps (linalg.py K+1 rule)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
        ev_ms = rep.gpu_ms
        assert rep.itr == K and rep.status == 0, f"work was skipped: itr={rep.itr}"
        assert bool(torch.isfinite(var()).all())
        passes = 3
        res_wg = ctx.resident_used()
        if res_wg > 0:
            # small mesh: the whole solve ran as ONE cooperative launch with the fields in LDS (pa_resident.hip);
            # there is no per-sweep kernel to bracket, a sweep costs stream time / sweeps
            roof = roofline("k_resident (Jacobi, whole solve in one launch)", ev_ms / K, passes * esize * mesh.N, args.workload,
                            bool(args.n), {"remark": f"128^2 fp64 = 128 KiB per array, resident in the LDS of {res_wg} workgroups: "
                                                     "bound by the grid-wide step of a sweep (one agent-scope arrive + wait), "
                                                     "not by HBM; kernel_ms = stream time of the solve / sweeps",
                                           "resident_workgroups": res_wg})
        elif not args.no_roofline_probe:
            ctx.profile(True)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ctx.solve("jacobi", var()[0], rhs[0], -1.0, min(K, 20) - 1)
            pr = ctx.profile_read()
            ctx.profile(False)
            if pr["phase_a_n"] > 0:
                roof = roofline("jacobi_sweep", pr["phase_a_ms"], passes * esize * mesh.N, args.workload, bool(args.n),
                                {"remark": "128^2 fp64 = 128 KiB per array: launch-latency bound, not HBM bound"})
        wl_text = f"2-D Poisson {gn[0]}x{gn[1]} {dtype}, Dirichlet BCs, Jacobi (BASELINE config 1)"

    if rank == 0:
        value = cells_global * K / secs
        if wl_text is None:
            wl_text = f"3-D Poisson {gn[0]}x{gn[1]}x{gn[2]} {dtype}, {kind} BCs, CG (BASELINE config {cfg_no})"
        out = {
            "metric": "cell-updates*iters/sec" + {"cg": ", 3-D Poisson CG", "bicgstab": ", 3-D Poisson BiCGSTAB",
                                                    "jacobi3": ", 3-D Poisson Jacobi"}.get(solver, ""),
            "value": value,
            "unit": "cell-updates*iters/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": secs * 1e3 / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if dtype == "double" else "f32",
            "data": "synthetic",
            "config": {"workload": wl_text, "global_cells": cells_global, "parallelism": parallelism},
            "hbm_alg_GBs": passes * esize * cells_global * K / secs / 1e9,
            "hbm_alg_frac_of_peak": passes * esize * cells_global * K / secs / 1e9 / (HBM_PEAK_GBS * world),
            "stream_ms_per_step": ev_ms / K,
        }
        if roof is not None:
            out["roofline"] = roof
        if first_attempt is not None:
            out["first_attempt"] = first_attempt
        if clocks:
            out["clocks"] = clocks
        if box:
            out["box"] = box
        if solver == "cg" and not slab:
            out["setup_ms"] = setup["wall_ms"]
            out["setup"] = setup
            out["probe"] = probe
        if not args.no_cpu_baseline and world == 1 and not slab:   # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(solver, kind, dtype, gn)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("BENCH_DUMP_MAPS"):
        # diagnostics: the address map of this process, so that raw frames of a crash inside exit() handlers
        # (profiles/README.md "c1 under rocprofv3") can be attributed to a library + offset afterwards
        with open("/proc/self/maps") as src, open(os.environ["BENCH_DUMP_MAPS"], "w") as dst:
            dst.write(src.read())


if __name__ == "__main__":
    main()
