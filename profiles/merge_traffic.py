#!/usr/bin/env python3
"""Fold the traffic_entry.json files of profile runs (gpurun_out/prof_*/) into profiles/traffic.json, which
bench.py reads into roofline.traffic -- entries carry the hash of the kernel sources they were measured on and
bench.py drops them when the sources have changed since.

    merge_traffic.py gpurun_out/prof_r02_c3 gpurun_out/prof_r02_c2 ...
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(HERE, "traffic.json")
t = json.load(open(path)) if os.path.exists(path) else {}
for d in sys.argv[1:]:
    f = os.path.join(d, "traffic_entry.json")
    if os.path.exists(f):
        t.update(json.load(open(f)))
with open(path, "w") as fh:
    json.dump(t, fh, indent=1)
print(json.dumps({k: (v.get("kernel"), round(v["bytes_per_launch"] / 1e6, 1), v.get("source_hash")) for k, v in t.items()}, indent=1))
