"""Per-kernel table of a profiled bench.py run (developer aid).

    python tools/profile_report.py profiles/r01 > profiles/r01/kernel_report.md

Joins the rocprofv3 `--kernel-trace --stats` summary (bench_kernel_stats.csv), the PMC traffic folded by
tools/pmc_traffic.py (../pmc_traffic.json) and the MFMA utilisation pass (pmc_mfma_util.csv) of the same command.
HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch; GB/s = traffic / mean duration of the counter pass.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import stage_name  # noqa: E402


def main(d):
    stats = {}
    with open(os.path.join(d, 'bench_kernel_stats.csv'), newline='') as fh:
        for r in csv.DictReader(fh):
            k = stage_name(r['Name'])
            s = stats.setdefault(k, [0, 0.0])
            s[0] += int(r['Calls'])
            s[1] += float(r['TotalDurationNs'])
    traffic = json.load(open(os.path.join(d, '..', 'pmc_traffic.json')))['kernels']
    mfma = {}
    with open(os.path.join(d, 'pmc_mfma_util.csv'), newline='') as fh:
        for r in csv.DictReader(fh):
            mfma[r['kernel']] = r
    total = sum(v[1] for v in stats.values())
    print('| kernel | launches | mean us | share of GPU time | HBM traffic / launch (PMC) | achieved GB/s | of 8 TB/s | MFMA busy | clock GHz |')
    print('|---|---|---|---|---|---|---|---|---|')
    for k, (n, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if ns / total < 1e-5:
            continue
        t = traffic.get(k)
        m = mfma.get(k)
        gbs = ''
        frac = ''
        tr = ''
        if t and m:
            gb = t['traffic_bytes_per_launch'] / (float(m['mean_us']) * 1e-6) / 1e9
            gbs, frac, tr = '%.0f' % gb, '%.2f' % (gb / 8000.0), '%.1f MB' % (t['traffic_bytes_per_launch'] / 1e6)
        print('| `%s` | %d | %.1f | %.1f %% | %s | %s | %s | %s | %s |' % (
            k, n, ns / n / 1e3, 100.0 * ns / total, tr, gbs, frac, m['mfma_util'] if m else '', m['clock_GHz'] if m else ''))


if __name__ == '__main__':
    main(sys.argv[1])
