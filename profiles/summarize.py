#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: per-kernel mean duration from the
kernel trace, per-kernel mean FETCH_SIZE / WRITE_SIZE from the PMC passes.

HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced (16 B/lane) streaming read and WRITE_SIZE reads the
bytes exactly for 16 B/lane streaming stores (MI355X_MICROARCH.md, section HBM)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


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


def counters(root, cname):
    out = defaultdict(list)
    for f in find(root, "*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != cname:
                    continue
                try:
                    out[row.get("Kernel_Name", "")].append(float(row["Counter_Value"]))
                except Exception:
                    pass
    return out


def short(name):
    import re
    m = re.search(r"k_cg3d<(\w+), (\d+), (\d+)>", name)
    if m:
        return f"k_cg3d_phase{'A' if m.group(3) == '0' else 'B'}_{'f64' if m.group(1) == 'double' else 'f32'}_RJ{m.group(2)}"
    m = re.search(r"(k_\w+)<(\w+)>", name)
    if m:
        return f"{m.group(1)}_{'f64' if m.group(2) == 'double' else 'f32'}"
    return name[:60]


def main():
    root = sys.argv[1]
    dur = kernel_durations(os.path.join(root, "kt"))
    fetch = counters(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write = counters(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    res = {}
    for name, v in dur.items():
        k = short(name)
        # skip the first (warm-up) launches
        vv = v[len(v) // 5:] if len(v) > 10 else v
        res.setdefault(k, {})
        res[k].update({"launches": len(v), "avg_us": sum(vv) / len(vv), "min_us": min(vv), "max_us": max(vv)})
    for src, key in ((fetch, "FETCH_SIZE_KiB"), (write, "WRITE_SIZE_KiB")):
        for name, v in src.items():
            k = short(name)
            vv = v[len(v) // 5:] if len(v) > 10 else v
            res.setdefault(k, {})[key] = sum(vv) / len(vv)
    for k, d in res.items():
        if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024
            if "avg_us" in d:
                d["hbm_GBs"] = d["hbm_bytes_per_launch"] / d["avg_us"] / 1e3
    top = dict(sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("launches", 0))[:12])
    print(json.dumps(top, indent=1))


if __name__ == "__main__":
    main()
