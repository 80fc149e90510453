#!/bin/bash
# A/B on one device: bench_conv of the headline shapes under different builds of the library.  usage: ab_lib.sh lib1.so lib2.so ...
for r in 1 2; do
for l in "$@"; do
  echo "== $l"
  TDG_LIB_PATH=$PWD/3dgan_amd/$l python3 tools/bench_conv.py 1536 16 16 200 400 5 2
  TDG_LIB_PATH=$PWD/3dgan_amd/$l python3 tools/bench_conv.py 1536 8 8 400 800 5 2
done
done
