#!/bin/bash
# One collection of the rocprofv3 evidence on the current sources (run through gpurun, in two or three calls of <= 20 min):
#   bash profiles/collect_all.sh <round tag> <part>      part: cg | euler | rest | lines
# cg: c3 c2 c5; euler: c4 c4t c4_512; rest: c1 slab64 + the BiCGSTAB kernel table; lines: the bench.py lines of all
# workloads (after profiles/traffic.json was merged from the parts before: they carry roofline.traffic) + bench_ops.py.
# Every step is joined with && : after a failed or killed GPU step nothing else starts.
set -uo pipefail
R="${1:-r04}"; PART="${2:-cg}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
P=profiles/run_profile.sh
case "$PART" in
  cg)    bash $P ${R}_c3 c3 --steps 20 --warmup 3 && bash $P ${R}_c2 c2 --steps 20 --warmup 3 && bash $P ${R}_c5 c5 --steps 20 --warmup 3 ;;
  euler) bash $P ${R}_c4 c4 --steps 60 --warmup 3 && bash $P ${R}_c4t c4t --steps 60 --warmup 3 && bash $P ${R}_c4_512 c4_512 --steps 60 --warmup 3 ;;
  rest)  bash $P ${R}_c1 c1 --steps 500 --warmup 10 &&
         RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BENCH_FORCE_SLAB=1 bash $P ${R}_slab64 c3 --size 64,512,512 --steps 20 --warmup 3 &&
         mkdir -p gpurun_out/prof_${R}_bicg && (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${R}_bicg/kt" -- python3 "$REPO/profiles/tools/solver_probe.py" 512 "bicgstab 512^3 f64 periodic,jacobi 512^3 f64 periodic" > "$REPO/gpurun_out/prof_${R}_bicg/probe.jsonl" 2> "$REPO/gpurun_out/prof_${R}_bicg/probe.err"; echo "bicg kernel trace rc=$?" > "$REPO/gpurun_out/prof_${R}_bicg/exit_codes.txt") &&
         f=$(find gpurun_out/prof_${R}_bicg/kt -name "*kernel_stats.csv" | head -1) && cp "$f" gpurun_out/prof_${R}_bicg/kernel_stats.csv && find gpurun_out/prof_${R}_bicg -name "*kernel_trace.csv" -size +2M -delete ;;
  lines) : > gpurun_out/${R}_bench_lines.jsonl
         for w in c3 c2 c5 c4 c4t c4_512 c1; do
           python bench.py --workload $w --no-cpu-baseline >> gpurun_out/${R}_bench_lines.jsonl 2>> gpurun_out/${R}_bench_lines.err || exit 1
         done
         python bench_ops.py > gpurun_out/${R}_bench_ops.jsonl 2> gpurun_out/${R}_bench_ops.err ;;
  rest2) RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BENCH_FORCE_SLAB=1 bash $P ${R}_slab64 c3 --size 64,512,512 --steps 20 --warmup 3 &&
         mkdir -p gpurun_out/prof_${R}_bicg && (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${R}_bicg/kt" -- python3 "$REPO/profiles/tools/solver_probe.py" 512 "bicgstab 512^3 f64 periodic,jacobi 512^3 f64 periodic" > "$REPO/gpurun_out/prof_${R}_bicg/probe.jsonl" 2> "$REPO/gpurun_out/prof_${R}_bicg/probe.err"; echo "bicg kernel trace rc=$?" > "$REPO/gpurun_out/prof_${R}_bicg/exit_codes.txt") &&
         f=$(find gpurun_out/prof_${R}_bicg/kt -name "*kernel_stats.csv" | head -1) && cp "$f" gpurun_out/prof_${R}_bicg/kernel_stats.csv && find gpurun_out/prof_${R}_bicg -name "*kernel_trace.csv" -size +2M -delete ;;
  *) echo "unknown part $PART"; exit 2 ;;
esac
