"""Where the SIMDs' issue time goes, per kernel, from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass over
bench.py (developer aid).

    python tools/pmc_issue.py <dir> > profiles/rNN/pmc_issue.csv

Every column is a fraction of the SIMD time of the launch, elapsed cycles x 1024 SIMDs with elapsed cycles =
GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles;
the SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES counters count quad-cycles summed over waves (x 4 here), so `waves_per_simd` is
the average number of resident waves and an `active_*` column can exceed 1 where several waves of a SIMD have an
instruction of that kind in flight at once.
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
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'launches', 'mean_us', 'mfma_busy', 'active_valu', 'active_lds', 'active_vmem', 'active_any', 'waves_per_simd'])
    for k in sorted(acc, key=lambda k: -dur[k]):
        if not n[k]:
            continue
        a = acc[k]
        simd = a['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
        if not simd:
            continue
        w.writerow([k, n[k], '%.1f' % (dur[k] / n[k] / 1e3), '%.3f' % (a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd)] +
                   ['%.3f' % (4.0 * a[c] / simd) for c in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM',
                                                           'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES')])


if __name__ == '__main__':
    main(sys.argv[1])
