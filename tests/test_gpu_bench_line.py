"""The benchmark's JSON line: one small run of bench.py as a child process, every field a reader of the line relies on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_line_with_roofline_and_cpu_baseline():
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '2',
           '--size', '256', '--words', '12', '--cpu-pages', '1']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['unit'] == 'images/s' and line['higher_is_better'] is True and line['scaling'] == 'weak'
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['warmup'] == 1
    assert line['dtype'] == 'f32' and line['data'].startswith('synthetic') and line['vs_baseline'] is None
    assert '256x256' in line['metric'] and 'workload' in line['config']
    assert line['value'] > 0
    # value = pages of the timed region / its wall time
    assert abs(line['value'] - 2 * 3 / (line['ms_per_step'] * 3e-3)) < 1e-6 * line['value']
    roof = line['roofline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and roof['peak'] == pytest.approx(157.3)
    assert roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert 0.0 < roof['frac'] <= 1.0 and 'traffic' in roof and roof['launches'] > 0
    base = line['cpu_baseline']
    assert base['kind'] == 'port' and base['unit'] == 'images/s' and base['value'] > 0 and base['cores'] >= 1 and base['sample']
    assert base['batched']['value'] > 0 and base['batched']['unit'] == 'images/s' and base['batched']['sample']
    names = {e['kernel'] for e in line['roofline_hbm']}
    assert {'east_decode', 'lanms', 'crop_resize_pack'} <= names
    for e in line['roofline_hbm']:
        assert e['algorithmic_bytes'] > 0 and e['avg_ms'] > 0 and 0.0 < e['frac_of_8TBs'] < 1.0
