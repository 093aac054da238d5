"""L2 -> CU read traffic per kernel and launch shape from one `rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum
TCC_HIT_sum TCC_MISS_sum` pass over bench.py (developer aid).

    python tools/pmc_l2.py <dir> > profiles/rNN/pmc_l2_requests.csv

One row per (kernel, workgroups of the launch): launches, mean duration under the counter pass, read requests of the
CUs' vector caches to the L2 (128 bytes each for the 16-byte-per-lane loads these kernels issue: a launch whose filter
slab and raw tiles are known reproduces its byte count at 128 B per request), the rate that gives, and the L2 hit rate.
"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import stage_name  # noqa: E402


def main(d):
    disp = OrderedDict()
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for r in csv.DictReader(fh):
                e = disp.setdefault((path, r['Dispatch_Id']), {
                    'k': stage_name(r['Kernel_Name']), 'wgs': int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])),
                    'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
                e[r['Counter_Name']] = float(r['Counter_Value'])
    agg = defaultdict(list)
    for e in disp.values():
        agg[(e['k'], e['wgs'])].append(e)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'workgroups', 'launches', 'mean_ms', 'TCP_TCC_READ_REQ', 'TCC_REQ', 'read_TB_per_s_at_128B', 'l2_hit_rate'])
    for (k, wgs), v in sorted(agg.items(), key=lambda kv: -sum(e['ns'] for e in kv[1])):
        n = len(v)
        ms = sum(e['ns'] for e in v) / n / 1e6
        rd = sum(e.get('TCP_TCC_READ_REQ_sum', 0.0) for e in v) / n
        rq = sum(e.get('TCC_REQ_sum', 0.0) for e in v) / n
        hit = sum(e.get('TCC_HIT_sum', 0.0) for e in v)
        miss = sum(e.get('TCC_MISS_sum', 0.0) for e in v)
        if ms * n < 0.05:
            continue
        w.writerow([k, wgs, n, '%.3f' % ms, '%.4g' % rd, '%.4g' % rq, '%.2f' % (rd * 128 / (ms * 1e-3) / 1e12 if ms else 0.0),
                    '%.3f' % (hit / (hit + miss) if hit + miss else 0.0)])


if __name__ == '__main__':
    main(sys.argv[1])
