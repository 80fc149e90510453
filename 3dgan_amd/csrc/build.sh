#!/bin/bash
# Builds lib3dgan_hip.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -e
cd "$(dirname "$0")"
OUT=../lib3dgan_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
pids=()
for f in tdg_igemm tdg_elementwise; do
  hipcc $FLAGS -c $f.hip -o $f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC tdg_igemm.o tdg_elementwise.o -o $OUT
echo "built $OUT"
