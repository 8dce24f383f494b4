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
# Every pass's exit code is recorded (exit_codes.txt) and the script fails if one of them did: a collection that
# crashed must not read as clean.  Known: under rocprofv3 7.2 ANY process that has made a cooperative launch dies
# with SIGSEGV in libhsa-runtime64 inside exit() (profiles/README.md; tools/coop_exit_repro.sh shows it with an
# 80-line program) -- after the tool has written its output.  The one workload with such a launch, c1 (resident
# solver), is therefore collected with a plain launch of the same kernel and grid (option resident_coop 0, through PYAPES_HIP_OPTIONS).
if [ "$WL" = "c1" ]; then export PYAPES_HIP_OPTIONS="${PYAPES_HIP_OPTIONS:-resident_coop=0}"; fi
: > "$OUT/exit_codes.txt"
FAILED=0
pass() {   # pass <name> <stdout file> <stderr file> <rocprofv3 args...>
  local name="$1" so="$2" se="$3"; shift 3
  echo "[profile] $name"
  rocprofv3 "$@" > "$so" 2> "$se"
  local rc=$?
  echo "$name rc=$rc" >> "$OUT/exit_codes.txt"
  if [ $rc -ne 0 ]; then echo "[profile] $name FAILED with exit code $rc (see $se)"; FAILED=1; fi
}
pass "kernel trace + stats" "$OUT/bench_kt.json" "$OUT/bench_kt.err" --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "${B[@]}"
pass "pmc FETCH_SIZE" "$OUT/bench_f.json" "$OUT/bench_f.err" --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- "${B[@]}" --no-roofline-probe
pass "pmc WRITE_SIZE" "$OUT/bench_w.json" "$OUT/bench_w.err" --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- "${B[@]}" --no-roofline-probe
pass "pmc SQ counters" "$OUT/bench_s.json" "$OUT/bench_s.err" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- "${B[@]}" --no-roofline-probe
python3 "$REPO/profiles/summarize.py" "$OUT" --traffic "$WL" > "$OUT/summary.json" 2> "$OUT/summarize.err" || echo "summarize failed"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
# keep only small files: drop the per-dispatch traces (can be tens of MB)
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
cat "$OUT/summary.json" | head -60
cat "$OUT/exit_codes.txt"
exit $FAILED
