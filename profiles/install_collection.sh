#!/bin/bash
# Copy what one collection (profiles/collect_all.sh, merged back under gpurun_out/) left into profiles/ under the names
# profiles/README.md lists, and fold the traffic entries into profiles/traffic.json:
#   bash profiles/install_collection.sh <round tag> <workload tag>...     e.g.  r04 c3 c2 c5
# Only files of passes that all ended with exit code 0 are installed; a collection with a failed pass is named and skipped.
set -uo pipefail
R="${1:?round tag}"; shift
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(dirname "$HERE")"
DIRS=()
for w in "$@"; do
  d="$ROOT/gpurun_out/prof_${R}_$w"
  if [ ! -f "$d/exit_codes.txt" ] || grep -qv "rc=0$" "$d/exit_codes.txt" || [ "$(wc -l < "$d/exit_codes.txt")" -lt 4 ]; then
    echo "skipped $w: no clean set of four passes in $d"; continue
  fi
  cp "$d/summary.json" "$HERE/${R}_${w}_summary.json"
  cp "$d/kernel_stats.csv" "$HERE/${R}_${w}_kernel_stats.csv"
  cp "$d/bench_kt.json" "$HERE/${R}_${w}_bench_under_rocprof.json"
  cp "$d/exit_codes.txt" "$HERE/${R}_${w}_exit_codes.txt"
  # a custom-size run (slab64 = workload c3 at 64 x 512 x 512) writes its traffic entry under the WORKLOAD's key: its
  # files are installed, its entry is never merged (bench.py reads no traffic for custom sizes)
  case "$w" in c1|c2|c3|c4|c4t|c4_512|c5) DIRS+=("$d") ;; esac
  echo "installed $w"
done
[ ${#DIRS[@]} -gt 0 ] && python3 "$HERE/merge_traffic.py" "${DIRS[@]}"
