"""Per-kernel totals of a `rocprofv3 --pmc ...` run (developer aid; reads *_counter_collection.csv).

    python tools/pmc_summary.py gpurun_out/pmc_fetch [more dirs] > profiles/rNN/pmc_<what>.csv

One line per (kernel, counter): launches, sum and mean of the counter, mean duration in microseconds.
FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; no correction is applied here.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace('ocrhip::(anonymous namespace)::', '').replace('ocrhip::', '')
    if name.startswith('void '):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):          # cut the argument list, keep template arguments
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return name[:i]
    return name


def main(dirs):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
            with open(path, newline='') as fh:
                for row in csv.DictReader(fh):
                    key = (short(row['Kernel_Name']), row['Counter_Name'])
                    a = acc[key]
                    a[0] += 1
                    a[1] += float(row['Counter_Value'])
                    a[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'counter', 'launches', 'sum', 'mean_per_launch', 'mean_duration_us'])
    for (k, c), (n, s, t) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
        w.writerow([k, c, n, '%.3f' % s, '%.3f' % (s / n), '%.2f' % (t / n)])


if __name__ == '__main__':
    main(sys.argv[1:] or ['.'])
