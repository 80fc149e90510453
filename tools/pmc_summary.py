"""Aggregates rocprofv3 --pmc counter_collection CSVs into per-kernel averages (the file bench.py's `roofline.traffic`
reads).  One --pmc pass per counter (MI355X_MICROARCH.md, HBM section):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_fetch_write_per_kernel.json
Values are the counters' own unit (KB per dispatch), uncorrected; bench.py applies the gfx950 correction."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out = {}
    for d in sys.argv[1:]:
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    a = acc[row['Counter_Name']][row['Kernel_Name'][:70]]
                    a[0] += 1
                    a[1] += float(row['Counter_Value'])
        for counter, kernels in acc.items():
            out[counter] = {k: {'dispatches': n, 'avg': tot / n} for k, (n, tot) in kernels.items()}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
