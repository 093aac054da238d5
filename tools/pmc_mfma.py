"""MFMA utilisation per kernel from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE` pass over bench.py (developer aid).

    python tools/pmc_mfma.py <dir> > profiles/rNN/pmc_mfma_util.csv

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs), elapsed cycles = GRBM_GUI_ACTIVE / 8
(the counter is summed over the 8 XCDs); clock_GHz = elapsed cycles / kernel duration.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import stage_name  # noqa: E402


def main(d):
    acc, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float)
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for r in csv.DictReader(fh):
                k = stage_name(r['Kernel_Name'])
                acc[k][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                    n[k] += 1
                    dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    names = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE']
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'launches', 'mean_us'] + names + ['mfma_util', 'clock_GHz'])
    for k in sorted(acc, key=lambda k: -dur[k]):
        if not n[k]:
            continue
        a = acc[k]
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0
        w.writerow([k, n[k], '%.1f' % (dur[k] / n[k] / 1e3)] + ['%.0f' % a[c] for c in names] +
                   ['%.3f' % (a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0) if cyc else 0.0),
                    '%.2f' % (cyc / dur[k] if dur[k] else 0.0)])


if __name__ == '__main__':
    main(sys.argv[1])
