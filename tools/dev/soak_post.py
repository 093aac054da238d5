"""Development aid: randomised sweep of LANMS / decode / crop on degenerate inputs against the oracle (prints)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_postproc as tp
from oracle import nms as onms
from ocr_amd.engine import HipContext
ctx = HipContext(0)
rng = np.random.default_rng(11)
bad = 0
for trial in range(150):
    B = 3
    bl, sl = [], []
    for _ in range(B):
        n = int(rng.integers(0, 120))
        grid = float(rng.choice([8, 16, 32]))
        xy = rng.integers(0, int(grid * 0.8), (n, 2)).astype(np.float32) / grid
        wh = rng.integers(0, 6, (n, 2)).astype(np.float32) / grid          # zero-area boxes included
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        if n > 4:
            d = rng.integers(0, n, n // 4)
            b[d] = b[rng.integers(0, n, n // 4)]                            # exact duplicates
        if trial % 5 == 0 and n:
            b[rng.integers(0, n, max(1, n // 10))] = 0                      # the reference's padding boxes
        s = (rng.integers(0, 12, n).astype(np.float32) / 12 * 0.6 + 0.36).astype(np.float32)   # heavy score ties
        bl.append(b); sl.append(s)
    method = ['union', 'average', 'union'][trial % 3]
    kw = dict(merge_threshold=float(rng.choice([0.1, 0.25, 0.3, 0.5, 1 / 3])), nms_threshold=float(rng.choice([0.05, 0.2, 0.25, 0.6])),
              merge_method=method)
    use_scores = trial % 4 != 3
    got = tp._gpu_lanms(ctx, bl, sl if use_scores else None, **kw)
    for b in range(B):
        try:
            tp._compare_with_oracle(got, b, bl[b], sl[b] if use_scores else None, onms.lanms, exact=(method == 'union'), **kw)
        except AssertionError as e:
            bad += 1
            print('LANMS MISMATCH trial', trial, 'b', b, 'n', len(bl[b]), kw, str(e)[:300].replace('\n', ' | '), flush=True)
print('lanms trials 150 x 3, bad', bad)
ctx.close()
