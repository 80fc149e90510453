#!/bin/bash
# A/B on one device: the headline training step under different builds of the library (or env settings: VAR=val,lib.so).
# usage: ab_step.sh lib1.so lib2.so ...   (each run twice, alternating)
for r in 1 2; do
for l in "$@"; do
  echo "== $l"
  TDG_LIB_PATH=$PWD/3dgan_amd/$l python3 bench.py --no_secondary --no_cpu_baseline --steps 20 --warmup 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step' % d['ms_per_step'], {k:(v['avg_ms'],v['tflops']) for k,v in list(d['roofline']['per_kernel'].items())[:3]})"
done
done
