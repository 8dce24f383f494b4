#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py workload on the GPU box (run through gpurun):
#   bash profiles/run_profile.sh <tag> <workload> [bench args...]
# Writes raw output under gpurun_out/prof_<tag>/ plus summary.json, kernel_stats.csv and traffic_entry.json
# (copy summary + stats into profiles/, fold the traffic entry in with profiles/merge_traffic.py).
# Kernel trace / stats and the PMC passes are separate runs (gpurun refuses mixed runs).
set -uo pipefail
TAG="${1:-r02}"; WL="${2:-c3}"; shift; shift || true
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(--steps 20 --warmup 3); fi
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=(python3 "$REPO/bench.py" --workload "$WL" "${ARGS[@]}" --no-cpu-baseline)
echo "[profile] kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "${B[@]}" > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || echo "kernel-trace run failed"
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- "${B[@]}" --no-roofline-probe > "$OUT/bench_f.json" 2> "$OUT/bench_f.err" || echo "pmc fetch run failed"
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- "${B[@]}" --no-roofline-probe > "$OUT/bench_w.json" 2> "$OUT/bench_w.err" || echo "pmc write run failed"
echo "[profile] pmc SQ instruction counters"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- "${B[@]}" --no-roofline-probe > "$OUT/bench_s.json" 2> "$OUT/bench_s.err" || echo "pmc sq run failed"
python3 "$REPO/profiles/summarize.py" "$OUT" --traffic "$WL" > "$OUT/summary.json" 2> "$OUT/summarize.err" || echo "summarize failed"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
# keep only small files: drop the per-dispatch traces (can be tens of MB)
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
cat "$OUT/summary.json" | head -60
