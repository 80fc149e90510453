#!/bin/bash
# A/B on one device: B-fragment prefetch distance of igemm_fwd_patch_kernel
for r in 1 2; do
for m in 2 3 4; do
  echo "== TDG_PATCH_PF=$m"
  TDG_PATCH_PF=$m python3 tools/bench_conv.py 1536 16 16 200 400 5 2
  TDG_PATCH_PF=$m python3 tools/bench_conv.py 1536 8 8 400 800 5 2
done
done
