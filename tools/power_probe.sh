#!/bin/bash
# Diagnostics ("name the limiter"): socket power, power cap, clocks, temperatures and the throttle /
# limiter status amd-smi exposes, sampled once a second while ONE kernel runs in a loop -- the forward patch kernel, the
# filter-gradient patch kernel and hipBLASLt's 8192^3 bf16 GEMM, 12 s each.  usage: bash tools/power_probe.sh TAG
TAG=${1:-r04}
cd "$(dirname "$0")/.."
OUT=gpurun_out/${TAG}_power_probe.txt
{
  echo "== static"; rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30
  amd-smi static -g 0 --limit 2>&1 | head -40
} > $OUT
for what in fwd wgrad matmul; do
  echo "== $what" >> $OUT
  python3 tools/loop_kernel.py $what 12 >> $OUT 2>&1 &
  PID=$!
  sleep 4
  for i in 1 2 3 4 5 6; do
    echo "-- sample $i" >> $OUT
    rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|junction\|edge" | head -12 >> $OUT
    amd-smi metric -g 0 --power --clock --temperature 2>&1 | grep -v "^$" | head -60 >> $OUT
    amd-smi metric -g 0 --throttle 2>&1 | grep -v "^$" | head -40 >> $OUT
    sleep 1
  done
  wait $PID
done
echo "wrote $OUT"
