#!/bin/bash
# Diagnostics: the forward LDS-DMA GEMM's loop with parts of its operand intake removed (results are garbage, only the
# time is read): TDG_DEBUG_ABLATE 0 = product, 4 / 5 / 6 = the A / B / both descriptors hold zero records (pieces are
# still issued, no L2 traffic), 7 = the A pieces are not issued at all.
# (the ablation switches are compiled into the diagnostic library only: build it with `bash 3dgan_amd/csrc/build.sh stamps`)
export TDG_LIB_PATH="$(dirname "$0")/../3dgan_amd/lib3dgan_hip_stamps.so"
for m in 0 4 5 6 7; do
  echo "== TDG_DEBUG_ABLATE=$m"
  TDG_DEBUG_ABLATE=$m python3 tools/bench_conv.py 1536 16 16 200 400 5 2
  TDG_DEBUG_ABLATE=$m python3 tools/bench_conv.py 1536 8 8 400 800 5 2
done
