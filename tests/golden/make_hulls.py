#!/usr/bin/env python3
"""Summation-order hulls of the REFERENCE ALGORITHM for the golden cases flagged ``sensitive`` (BiCGSTAB, CG on the
reference's non-symmetric periodic operator), long runs only -- what tests/test_gpu_parity_golden.py grades those runs
against (its docstring has the statistics).  A hull is 29 solves of the oracle (5 structured + 24 random orders of every
torch.sum: tests/helpers.summation_hull) and takes seconds to minutes of CPU time per case -- 160 s of the GPU suite for
two cases alone -- while depending on nothing but the oracle and the committed inputs.  So it is computed here, once, and
committed as data: tests/golden/hulls.npz (band, diam, iteration counts, per-iteration scalar histories per sample).

    python tests/golden/make_hulls.py          # needs no GPU and no reference: oracle + tests/golden/*.npz

tests/test_oracle_golden.py::test_committed_hull_is_what_the_oracle_gives recomputes the cheapest one on every CPU run.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def hull_keys(name, K):
    return f"{name}|K{K}"


def pack(band, diam, its, hs):
    n = max(len(h) for h in hs)
    cols = hs[0].shape[1]
    H = np.full((len(hs), n, cols), np.nan)
    for q, h in enumerate(hs):
        H[q, :len(h)] = h
    return {"band": np.float64(band), "diam": np.float64(diam), "its": np.asarray(its, dtype=np.int64),
            "hist": H, "hist_len": np.asarray([len(h) for h in hs], dtype=np.int64)}


def unpack(z, key):
    H, L = z[key + "|hist"], z[key + "|hist_len"]
    hs = [H[q, :int(L[q])].copy() for q in range(len(L))]
    return float(z[key + "|band"]), float(z[key + "|diam"]), [int(v) for v in z[key + "|its"]], hs


def main():
    from conftest import golden_cases, golden_load
    from helpers import summation_hull
    out = {}
    for case in golden_cases("solve"):
        if not case.get("sensitive"):
            continue
        g = golden_load(case["name"])
        for K in case["max_its"]:
            if K <= 10:
                continue
            t0 = time.time()
            hs = []
            band, diam, its = summation_hull(case, g["rhs0"], K, g[f"x_K{K}"], histories=hs)
            for k, v in pack(band, diam, its, hs).items():
                out[hull_keys(case["name"], K) + "|" + k] = v
            print(f"{case['name']} K={K}: band {band:.3e} diam {diam:.3e} its {min(its)}..{max(its)}  ({time.time() - t0:.1f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "hulls.npz"), **out)
    print("wrote", os.path.join(HERE, "hulls.npz"), os.path.getsize(os.path.join(HERE, "hulls.npz")), "bytes")


if __name__ == "__main__":
    main()
