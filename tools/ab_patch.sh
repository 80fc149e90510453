#!/bin/bash
# A/B on one device: igemm_fwd_dma_kernel (TDG_PATCH=0) vs igemm_fwd_patch_kernel (default) on the headline GEMM shapes
for r in 1 2; do
for m in 0 1; do
  echo "== TDG_PATCH=$m"
  TDG_PATCH=$m python3 tools/bench_conv.py 1536 16 16 200 400 5 2
  TDG_PATCH=$m python3 tools/bench_conv.py 1536 8 8 400 800 5 2
done
done
