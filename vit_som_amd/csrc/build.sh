#!/bin/bash
# Build libvitsom_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
mkdir -p "$HERE/obj"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
pids=()
newest_h="$(ls -t "$HERE"/*.h "$HERE/../../include/vitsom_hip.h" | head -1)"
for f in gemm_f32 som som_l1 layernorm attention misc bmu_x3 comm tape; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$newest_h" -nt "$HERE/obj/$f.o" ]; then
    hipcc $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvitsom_hip.so" "$HERE"/obj/{gemm_f32,som,som_l1,layernorm,attention,misc,bmu_x3,comm,tape}.o -ldl
echo "built $OUT/libvitsom_hip.so"
