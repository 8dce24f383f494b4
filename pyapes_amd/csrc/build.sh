#!/bin/bash
# Build libpyapes_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : the stencil arithmetic must round every product and sum
#                       separately to reproduce the reference bit for bit (pa_device.h)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function -Wno-unused-variable \
  ${PA_EXTRA_FLAGS:-} \
  -o "$OUT/libpyapes_hip.so" "$HERE/pa_core.hip" "$HERE/pa_cg3d.hip" "$HERE/pa_comm.hip" "$HERE/pa_rfp.hip" -ldl
echo "built $OUT/libpyapes_hip.so"
