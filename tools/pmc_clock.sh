#!/bin/bash
# Diagnostics: effective shader clock and MFMA-busy share of every kernel of a command, from the SQ cycle counters
# (SQ_BUSY_CYCLES / 32 shader engines / duration).  usage: bash tools/pmc_clock.sh TAG command ...
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_clock_$TAG -- "$@" > gpurun_out/pmc_clock_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(list)
for path in glob.glob('gpurun_out/pmc_clock_$TAG/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        a = acc[row['Kernel_Name'][:70]][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for path in glob.glob('gpurun_out/pmc_clock_$TAG/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        dur[row['Kernel_Name'][:70]].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for k, c in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
    if not dur[k] or sum(dur[k]) / len(dur[k]) < 20000: continue
    busy = c['SQ_BUSY_CYCLES'][1] / c['SQ_BUSY_CYCLES'][0] / 32
    d = sum(dur[k]) / len(dur[k])
    print('%-70s %4d launches %8.1f us  %.2f GHz effective  MFMA busy %.3f' % (k, len(dur[k]), d / 1e3, busy / d, c['SQ_VALU_MFMA_BUSY_CYCLES'][1] / c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / 1024 / busy))
PY
