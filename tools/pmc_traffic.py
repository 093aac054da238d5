"""Turn the FETCH_SIZE and WRITE_SIZE passes of `rocprofv3 --pmc` over bench.py into per-kernel HBM traffic.

    python tools/pmc_traffic.py <fetch dir> <write dir> > profiles/pmc_traffic.json

Kernel names are folded to the stage names bench.py reports (one name per convolution variant and tile shape).  traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch: on gfx950 FETCH_SIZE tallies the 128-byte
reads of 16-byte-per-lane loads at 64 bytes, WRITE_SIZE is exact for 16-byte-per-lane and dword stores.
Both counters are reported in KiB.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def stage_name(kernel):
    m = re.search(r'conv_mfma_f32_kernel<(\d+), (\d+), (\d+), (true|false), (\d+), (\d+)>', kernel)
    if m:
        ks, tw, mt = int(m.group(1)), int(m.group(5)), int(m.group(6))
        rows, cols = (256, 1) if ks == 1 else (4 * mt * (32 // tw), tw)
        return 'conv_mfma_f32<ks%d,n%d,pool%s,up%d,tile%dx%d>' % (ks, 32 * int(m.group(2)), m.group(3),
                                                                  m.group(4) == 'true', rows, cols)
    m = re.search(r'conv_wino4_f32_kernel<(\d+), (\d+)>', kernel)
    if m:
        if m.group(2) == '1':
            return 'conv_wino4_transform'
        kind = {'0': '', '2': '_mul', '3': '_h', '4': '_h'}[m.group(2)]
        return 'conv_wino4%s_f32<n%s,pool%s,tile%sx32>' % (kind, '128' if m.group(2) == '4' else '64', m.group(1),
                                                           '8' if m.group(2) in '34' else '16')
    m = re.search(r'conv_bf16t_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>', kernel)
    if m:
        ks, ck, bn, pool, f32, first = m.groups()
        if first == 'true':
            return 'conv_bf16t<first pair,n%s,pool%s>' % (bn, pool)
        if ks == '3':
            return 'conv_bf16t<ks3,n%s,k%s,pool%s>' % (bn, ck, pool)
        return 'gemm_bf16t<n%s,k%s,%s>' % (bn, ck, 'f32' if f32 == 'true' else 'bf16')
    m = re.search(r'lstm_layer_waves16_kernel', kernel)
    if m:
        return 'lstm_layer16(T steps)'
    m = re.search(r'lstm_layer(_waves)?_kernel', kernel)
    if m:
        return 'lstm_layer'
    m = re.search(r'conv_wino_f32_kernel<(\d+), (\d+)>', kernel)
    if m:
        return 'conv_wino_f32<n%d,pool%s,tile4x32>' % (32 * int(m.group(2)), m.group(1))
    m = re.search(r'(\w+)_kernel', kernel)
    if m:
        return m.group(1)
    return kernel.split('(anonymous namespace)::')[-1].split('(')[0].replace('void ', '')


def totals(d):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                a = acc[stage_name(row['Kernel_Name'])]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    return acc


def main(pairs):
    """pairs: (fetch_dir, write_dir) per measured command (float32 run, bfloat16 run): their kernels have different names."""
    out = {}
    for fetch_dir, write_dir in pairs:
        f, w = totals(fetch_dir), totals(write_dir)
        for k in sorted(set(f) & set(w)):
            if f[k][0] != w[k][0] or k in out:
                continue                  # the two passes must have seen the same launches; the first command that ran a kernel names it
            n = f[k][0]
            fetch, write = f[k][1] / n * 1024.0, w[k][1] / n * 1024.0
            out[k] = {'launches': n, 'fetch_size_bytes_per_launch': fetch, 'write_size_bytes_per_launch': write,
                      'traffic_bytes_per_launch': 2.0 * fetch + write}
    json.dump({'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py [--dtype bf16] --no-cpu-baseline --steps 1 --warmup 1',
               'correction': 'traffic = 2 x FETCH_SIZE + WRITE_SIZE', 'kernels': out}, sys.stdout, indent=1)


if __name__ == '__main__':
    a = sys.argv[1:]
    main([(a[i], a[i + 1]) for i in range(0, len(a) - 1, 2)])
