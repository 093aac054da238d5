"""Builds ocr_amd/libocrhip.so from the HIP sources in this directory (gfx950 only).

    python -m ocr_amd.csrc.build            # incremental
    python -m ocr_amd.csrc.build --force

hipcc cross-compiles without a GPU; the .so is built in-tree so that it travels with the
repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, 'libocrhip.so')
OBJ_DIR = os.path.join(HERE, '_obj')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
COMMON = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
          '-I', os.path.join(os.path.dirname(PKG), 'include')]

# Files whose floating-point results are compared bit for bit with the CPU oracle are built without
# multiply-add contraction.
SOURCES = {
    'api.hip': [],
    'decode.hip': ['-ffp-contract=off'],
    'lanms.hip': ['-ffp-contract=off'],
    'crop.hip': ['-ffp-contract=off'],
    'ctc.hip': ['-ffp-contract=off'],
    'ctc_beam.hip': ['-ffp-contract=off'],
    'conv_direct.hip': [],
    'conv_mfma.hip': [],
    'gemm_f32.hip': [],
    'conv_wino.hip': ['-fno-slp-vectorize'],
    'conv_wino4.hip': [],
    'conv_bf16.hip': [],
    'conv_bf16t.hip': [],
    'bf16_misc.hip': [],
    'rnn.hip': [],
    'rnn_persistent.hip': [],
    'pool.hip': [],
    'metrics.hip': ['-ffp-contract=off'],
    'rnms.hip': ['-ffp-contract=off'],
}
HEADERS = ['common.h', os.path.join('..', '..', 'include', 'ocrhip.h')]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src, flags, force):
    obj = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and _mtime(obj) > max(_mtime(d) for d in deps):
        return obj, False
    cmd = [HIPCC] + COMMON + flags + ['-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, ' '.join(cmd), r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda kv: _compile(kv[0], kv[1], force), SOURCES.items()))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB) or force:
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-8000:]))
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
