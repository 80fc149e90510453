#!/bin/bash
# Counter evidence for profiles/ (run on the GPU box from the repo root; MI355X_MICROARCH.md, HBM / rocprofv3 sections):
# separate --pmc passes of the SAME bench command (FETCH_SIZE and WRITE_SIZE cannot share a pass), no trace domains
# beside --kernel-trace, the program itself right behind `--`.
#   usage: bash tools/collect_pmc.sh rNN_x [headline|all]  ->  gpurun_out/pmc_<tag>_{fetch,write,mfma}/ + gpurun_out/<tag>_pmc_summary.json
#   `all` (default) also runs the pix2pix bs 64 and VAE bs 512 legs of config.secondary, so their kernels
#   (igemm_fwd_dma_kernel<bf16,256,128,3>, igemm_wgrad_dma_kernel<256,.>, bwd_col2im_kernel, thin_fwd_kernel<64>, conv_n1_fwd_kernel,
#   splitk_finish_kernel, the batch-norm passes) appear in the summary; `headline` is the round-2 form (--no_secondary).
set -e
TAG=${1:-r03}
WHAT=${2:-all}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "$WHAT" = "headline" ]; then
  CMD="python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary"
else
  CMD="python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --secondary_legs pix2pix,vae"
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fetch -- $CMD > gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_write -- $CMD > gpurun_out/pmc_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_mfma -- $CMD > gpurun_out/pmc_${TAG}_mfma.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_mfma > gpurun_out/${TAG}_pmc_summary.json
echo "wrote gpurun_out/${TAG}_pmc_summary.json"
