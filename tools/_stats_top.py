"""Diagnostics: top kernels of a rocprofv3 --kernel-trace --stats --output-format csv directory.  usage: python tools/_stats_top.py DIR [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print('%-95s %5s %8.3f ms %5.1f%% %7.1f us' % (r['Name'][:95], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                   100 * float(r['TotalDurationNs']) / tot, float(r['AverageNs']) / 1e3))
