"""Aggregates rocprofv3 --pmc passes (tools/collect_pmc.sh) into per-kernel averages: the file bench.py's `roofline.traffic`
reads, plus the derived figures the north-star asks for -- HBM-side bytes and GB/s per launch, MFMA-busy share.

    python tools/pmc_summary.py DIR_FETCH DIR_WRITE DIR_MFMA > profiles/rNN_x_pmc_fetch_write_per_kernel.json

Counter values are kept in their own unit (FETCH_SIZE / WRITE_SIZE: KB per dispatch); `derived` applies the gfx950
corrections of MI355X_MICROARCH.md: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-B requests as
64 B); SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs, so
mfma_busy = (MFMA / 1024) / (BUSY / 32).  Durations come from the kernel trace of the same passes."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict


def main():
    out = {}
    dur = defaultdict(lambda: [0, 0.0])
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
        for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    e = dur[row['Kernel_Name'][:70]]
                    e[0] += 1
                    e[1] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    derived = {}
    for k, (n, tot) in dur.items():
        ns = tot / n
        e = {'avg_duration_us': ns / 1e3}
        f, w = out.get('FETCH_SIZE', {}).get(k), out.get('WRITE_SIZE', {}).get(k)
        if f and w:
            b = (2.0 * f['avg'] + w['avg']) * 1024.0
            e['hbm_side_bytes_per_launch'] = b
            e['hbm_side_TBps'] = b / ns / 1e3
            e['frac_of_8TBps'] = b / ns / 1e3 / 8.0
        m, bz = out.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get(k), out.get('SQ_BUSY_CYCLES', {}).get(k)
        if m and bz and bz['avg'] > 0:
            e['mfma_busy'] = (m['avg'] / 1024.0) / (bz['avg'] / 32.0)
        derived[k] = e
    out['derived'] = dict(sorted(derived.items(), key=lambda kv: -kv[1]['avg_duration_us'] * dur[kv[0]][0]))
    try:
        out['build'] = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        bid = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3dgan_amd', 'BUILD_ID')
        out['build'] = open(bid).read().strip() if os.path.exists(bid) else 'unknown'     # (the GPU box has no .git)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
