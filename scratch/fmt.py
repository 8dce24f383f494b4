import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
    except Exception:
        print(l.rstrip()); continue
    print("%-78s %9.4f ms %10.3e c/s  alg %6.0f GB/s (%5.1f%%)  wall/it %.4f" % (d["workload"][:78], d["ms"], d["cell_updates_per_s"], d["alg_GBs"], 100*d["frac_of_hbm_peak"], d.get("wall_ms_per_iter", 0)))
