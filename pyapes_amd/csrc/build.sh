#!/bin/bash
# Build libpyapes_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : the stencil arithmetic must round every product and sum
#                       separately to reproduce the reference bit for bit (pa_device.h)
# The translation units are compiled in parallel (objects under build/, git-ignored), then linked.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/build"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable ${PA_EXTRA_FLAGS:-})
pids=()
for tu in pa_core pa_bc pa_ops pa_solver pa_cg3d pa_cg3d_b pa_sf pa_comm pa_comm_hostring pa_rfp pa_resident pa_place; do
  "$HIPCC" "${FLAGS[@]}" -c "$HERE/$tu.hip" -o "$OBJ/$tu.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libpyapes_hip.so" "$OBJ/pa_core.o" "$OBJ/pa_bc.o" "$OBJ/pa_ops.o" "$OBJ/pa_solver.o" "$OBJ/pa_cg3d.o" "$OBJ/pa_cg3d_b.o" "$OBJ/pa_sf.o" "$OBJ/pa_comm.o" "$OBJ/pa_comm_hostring.o" "$OBJ/pa_rfp.o" "$OBJ/pa_resident.o" "$OBJ/pa_place.o" -ldl -lrt -lpthread
echo "built $OUT/libpyapes_hip.so"
