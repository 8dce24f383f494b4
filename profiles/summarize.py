#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: per-kernel mean duration from the
kernel trace, per-kernel mean FETCH_SIZE / WRITE_SIZE (and SQ instruction counters, if collected)
from the PMC passes.

HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced (16 B/lane) streaming read and WRITE_SIZE reads the
bytes exactly for 16 B/lane streaming stores (MI355X_MICROARCH.md, section HBM).

    summarize.py <dir> [--traffic <workload>]     # --traffic: update profiles/traffic.json for bench.py
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def kernel_durations(root):
    out = defaultdict(list)
    for f in find(root, "*kernel_trace.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                try:
                    out[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
                except Exception:
                    pass
    return out


def counters(root):
    out = defaultdict(lambda: defaultdict(list))
    for f in find(root, "*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                try:
                    out[row.get("Counter_Name")][row.get("Kernel_Name", "")].append(float(row["Counter_Value"]))
                except Exception:
                    pass
    return out


PHASES = {0: "phaseA", 1: "phaseB", 2: "Ax", 3: "euler", 4: "jacobi", 5: "bicg_pv", 6: "bicg_st", 7: "grad", 8: "bicg_v",
          9: "jacobi_backwards"}


def short(name):
    m = re.search(r"k_cg3d<(\w+), (\d+), (\d+)(?:, (\w+), (\d+), (\w+))?>", name)
    if m:
        t = "f64" if m.group(1) == "double" else "f32"
        s = f"k_cg3d_{PHASES.get(int(m.group(3)), m.group(3))}_{t}_RJ{m.group(2)}"
        if m.group(4) == "true":
            s += "_CF"
        if m.group(5) and m.group(5) != "0":
            s += f"_kind{m.group(5)}"
        if m.group(6) == "true":
            s += "_narrow"
        return s
    m = re.search(r"k_sf<(\w+), (\d+), (\d+), (\d+), (\w+)(?:, (\w+))?(?:, (\d+))?>", name)   # (..., HASU[, BCL[, US]])
    if m:
        t = "f64" if m.group(1) == "double" else "f32"
        return (f"k_sf_{PHASES.get(int(m.group(3)), m.group(3))}_{t}_RJ{m.group(2)}_kind{m.group(4)}" +
                ("_ufield" if m.group(5) == "true" else "") + ("_bcl" if m.group(6) == "true" else "") +
                ({"1": "_upos", "2": "_uneg"}.get(m.group(7) or "0", "")))
    m = re.search(r"k_resident<(\w+), (\d+), (\w+), (\d+)>", name)
    if m:
        t = "f64" if m.group(1) == "double" else "f32"
        sv = {"0": "cg", "1": "jacobi", "2": "bicgstab"}.get(m.group(2), m.group(2))
        return f"k_resident_{sv}_{t}" + ("_lean" if m.group(3) == "true" else "_general") + f"_NT{m.group(4)}"
    m = re.search(r"(k_\w+)<(\w+)>", name)
    if m:
        return f"{m.group(1)}_{'f64' if m.group(2) == 'double' else 'f32'}"
    return name[:60]


def main():
    root = sys.argv[1]
    dur = kernel_durations(os.path.join(root, "kt"))
    pmc = counters(root)
    res = {}

    def trimmed(v):
        return v[len(v) // 5:] if len(v) > 10 else v   # skip the first (warm-up) launches

    for name, v in dur.items():
        vv = trimmed(v)
        res.setdefault(short(name), {}).update(
            {"launches": len(v), "avg_us": sum(vv) / len(vv), "min_us": min(vv), "max_us": max(vv)})
    for cname, per_kernel in pmc.items():
        key = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB"}.get(cname, cname)
        for name, v in per_kernel.items():
            vv = trimmed(v)
            res.setdefault(short(name), {})[key] = sum(vv) / len(vv)
    for k, d in res.items():
        if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024
            if "avg_us" in d:
                d["hbm_GBs"] = d["hbm_bytes_per_launch"] / d["avg_us"] / 1e3
    top = dict(sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("launches", 0))[:14])
    print(json.dumps(top, indent=1))

    if "--traffic" in sys.argv:
        wl = sys.argv[sys.argv.index("--traffic") + 1]
        sys.path.insert(0, os.path.dirname(HERE))
        from bench import source_hash
        # the GPU box only returns gpurun_out/: write the entries beside the summary; profiles/merge_traffic.py
        # folds them into profiles/traffic.json afterwards
        path = os.path.join(root, "traffic_entry.json")
        t = {}
        roles = {"phaseA": "cg_phase_a", "phaseB": "cg_phase_b", "euler": "euler_step", "jacobi": "jacobi_sweep"}
        for k, d in res.items():
            if "hbm_bytes_per_launch" not in d or d.get("launches", 0) < 5:
                continue
            for tag, role in roles.items():
                if f"_{tag}_" in k and not k.endswith("_narrow"):
                    t[f"{wl}:{role}"] = {
                        "bytes_per_launch": d["hbm_bytes_per_launch"], "avg_us": d.get("avg_us"), "kernel": k,
                        "source_hash": source_hash(),
                        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2*FETCH+WRITE KiB "
                                  f"(gfx950 correction); profiles/{os.path.basename(root.rstrip('/')).replace('prof_', '', 1)}_summary.json"}
        with open(path, "w") as f:
            json.dump(t, f, indent=1)


if __name__ == "__main__":
    main()
