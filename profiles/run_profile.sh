#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   bash profiles/run_profile.sh <tag> [bench args...]
# Writes raw output under gpurun_out/prof_<tag>/ and a summary gpurun_out/prof_<tag>/summary.json
# (copy that + the *_kernel_stats.csv into profiles/ to have them judged).
# Kernel trace/stats and the PMC passes are separate runs (gpurun refuses mixed runs).
set -uo pipefail
TAG="${1:-r01}"; shift || true
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(--steps 20 --warmup 3); fi
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[profile] kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$REPO/bench.py" "${ARGS[@]}" --no-cpu-baseline > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || echo "kernel-trace run failed"
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" "${ARGS[@]}" --no-cpu-baseline --no-roofline-probe > "$OUT/bench_f.json" 2> "$OUT/bench_f.err" || echo "pmc fetch run failed"
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" "${ARGS[@]}" --no-cpu-baseline --no-roofline-probe > "$OUT/bench_w.json" 2> "$OUT/bench_w.err" || echo "pmc write run failed"
python3 "$REPO/profiles/summarize.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summarize.err" || echo "summarize failed"
# keep only small files: drop the per-dispatch traces (can be tens of MB)
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
ls -la "$OUT" "$OUT"/kt/* 2>/dev/null | head -30
cat "$OUT/summary.json"
