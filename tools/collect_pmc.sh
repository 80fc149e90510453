#!/bin/bash
# Counter evidence for profiles/ (run on the GPU box from the repo root; MI355X_MICROARCH.md, HBM / rocprofv3 sections):
# separate --pmc passes of the SAME bench command (FETCH_SIZE and WRITE_SIZE cannot share a pass), no trace domains
# beside --kernel-trace, the program itself right behind `--`.
#   usage: bash tools/collect_pmc.sh rNN_x   ->  gpurun_out/pmc_<tag>_{fetch,write,mfma}/ + gpurun_out/<tag>_pmc_summary.json
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fetch -- $CMD > gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_write -- $CMD > gpurun_out/pmc_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_mfma -- $CMD > gpurun_out/pmc_${TAG}_mfma.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_mfma > gpurun_out/${TAG}_pmc_summary.json
echo "wrote gpurun_out/${TAG}_pmc_summary.json"
