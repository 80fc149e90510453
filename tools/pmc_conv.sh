#!/bin/bash
# Diagnostics: SQ counters of one conv shape's kernels (MFMA busy, LDS conflicts, wait buckets).  usage: bash tools/pmc_conv.sh TAG n h w cin cout k s
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_conv_$TAG -- python3 tools/bench_conv.py "$@" > gpurun_out/pmc_conv_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob('gpurun_out/pmc_conv_$TAG/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'][:60]
        if 'igemm' not in k and 'fused' not in k and 'thin' not in k: continue
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, cs in acc.items():
    v = {c: t / n for c, (n, t) in cs.items()}
    print(k)
    print('   mfma_busy %.3f  lds_conflict/idx_active %.3f  wait_inst_lds/wave-ish %.3g  wait_inst_any %.3g  wait_any %.3g  active_inst %.3g' % (
        (v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (v['SQ_BUSY_CYCLES'] / 32), v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1),
        v['SQ_WAIT_INST_LDS'], v['SQ_WAIT_INST_ANY'], v['SQ_WAIT_ANY'], v['SQ_ACTIVE_INST_ANY']))
PY
