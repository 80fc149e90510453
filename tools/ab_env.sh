#!/bin/bash
# A/B on one device: bench_conv of the headline shapes under values of one environment variable.  usage: ab_env.sh VAR v1 v2 ...
VAR=$1; shift
for r in 1 2; do
for m in "$@"; do
  echo "== $VAR=$m"
  env $VAR=$m python3 tools/bench_conv.py 1536 16 16 200 400 5 2
  env $VAR=$m python3 tools/bench_conv.py 1536 8 8 400 800 5 2
done
done
