#!/bin/bash
# Builds lib3dgan_hip.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -e
cd "$(dirname "$0")"
OUT=../lib3dgan_hip.so
EXTRA=""
if [ "$1" = "stamps" ]; then OUT=../lib3dgan_hip_stamps.so; EXTRA="-DTDG_STAMPS"; fi   # diagnostic build with in-kernel cycle stamps
FLAGS="$EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
pids=()
for f in tdg_igemm tdg_elementwise tdg_wgrad_patch; do
  hipcc $FLAGS -c $f.hip -o $f$1.o &
  pids+=($!)
done
g++ -O2 -std=c++17 -fPIC -Wall -c tdg_host.cpp -o tdg_host.o   # host-only helpers
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC tdg_igemm$1.o tdg_elementwise$1.o tdg_wgrad_patch$1.o tdg_host.o -o $OUT
# which commit the library was built from (the GPU box has no .git: tools/pmc_summary.py names the build in its summaries)
(git -C .. rev-parse --short HEAD 2>/dev/null | tr -d "\n"; git -C .. diff --quiet 2>/dev/null || printf "+"; echo) > ../BUILD_ID || true
echo "built $OUT"
