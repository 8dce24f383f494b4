#!/bin/bash
# A process that has made ONE hipLaunchCooperativeKernel dies with SIGSEGV inside exit() under rocprofv3 7.2
# (--kernel-trace is enough), after the tool has written its output: shown here with gridbar.hip, an 80-line
# program that links nothing of this repository.  Round 3, MI355X box (gpurun_out/diag1):
#   ./gridbar                                  -> rc 0
#   rocprofv3 --kernel-trace --stats -- ./gridbar   -> rc 139; frames: __cxa_finalize -> libamdhip64.so (atexit handler)
#       -> libhsa-runtime64.so.1.18.70200 (+0x60097 ... +0x6359e) -> SIGSEGV at an address inside a /dev/dri/renderD*
#       mapping; no frame of libpyapes_hip.so (there is none in the process)
#   the same frames, same offsets, for `bench.py --workload c1` (resident solver = the one cooperative launch of
#   the library); with PYAPES_HIP_RESIDENT=0 (launch-per-phase loops) or any other workload: rc 0.
set -uo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-/tmp/coop_exit_repro}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/gridbar "$HERE/gridbar.hip" || exit 2
/tmp/gridbar > "$OUT/plain.out" 2>&1; echo "plain: rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- /tmp/gridbar > "$OUT/kt.out" 2> "$OUT/kt.err"; echo "under rocprofv3 --kernel-trace: rc=$?"
grep -A4 "SIGSEGV" "$OUT/kt.err" | head -8
