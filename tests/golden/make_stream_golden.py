"""Generates tests/golden/stream_golden.npz with the reference's own frame-to-frame filter
(utils/image/bounding_box/filters.py: RepetitionFilter) and Dice coefficient (utils/distances.py: dice_coeff)
on seeded synthetic sequences.

    python tests/golden/make_stream_golden.py

Sequences: every frame holds a handful of paragraphs (1-3 rows each, rows made of 1-4 words) drawn from a fixed
set of places; a place appears or not per frame, with a small jitter, sometimes with a different number of rows.
Per scenario (n_repeat, max_unseen, use_memory, iou_threshold) the file stores what the reference kept per frame.
"""
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))

SCENARIOS = [
    dict(n_repeat=2, max_unseen=3, use_memory=False, iou_threshold=0.5),
    dict(n_repeat=3, max_unseen=1, use_memory=False, iou_threshold=0.5),
    dict(n_repeat=2, max_unseen=3, use_memory=True, iou_threshold=0.6),
    dict(n_repeat=1, max_unseen=3, use_memory=True, iou_threshold=0.5),
]
N_FRAMES, N_PLACES = 24, 7


def make_sequence(seed):
    """-> list of frames; frame = (boxes (P,4), rows list of (k,4), indices list of nested lists)."""
    rng = np.random.default_rng(seed)
    places = []
    for p in range(N_PLACES):
        x0, y0 = rng.uniform(0.02, 0.6), rng.uniform(0.02, 0.8)
        n_rows = int(rng.integers(1, 4))
        w, h = rng.uniform(0.15, 0.35), rng.uniform(0.03, 0.05)
        rows = np.array([[x0, y0 + r * h * 1.3, x0 + w, y0 + r * h * 1.3 + h] for r in range(n_rows)], dtype=np.float32)
        words = [int(rng.integers(1, 5)) for _ in range(n_rows)]
        places.append((rows, words))
    frames = []
    for f in range(N_FRAMES):
        boxes, rows_l, idx_l, nxt = [], [], [], 0
        for rows, words in places:
            if rng.random() < 0.45:
                continue
            r = rows + rng.normal(0, 0.004, rows.shape).astype(np.float32)
            w = list(words)
            if rng.random() < 0.15 and len(r) > 1:                 # a row lost by the detector
                r, w = r[:-1], w[:-1]
            if rng.random() < 0.1:                                 # moved away
                r = r + np.float32(0.2)
            idx = []
            for k in w:
                idx.append(list(range(nxt, nxt + k)))
                nxt += k
            boxes.append([r[:, 0].min(), r[:, 1].min(), r[:, 2].max(), r[:, 3].max()])
            rows_l.append(r.astype(np.float32))
            idx_l.append(idx)
        frames.append((np.array(boxes, dtype=np.float32).reshape(-1, 4), rows_l, idx_l))
    return frames


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from utils.image.bounding_box import filters as F
    from utils.distances import dice_coeff
    gold = {}
    for s, kw in enumerate(SCENARIOS):
        frames = make_sequence(100 + s)
        filt = F.RepetitionFilter(**kw)
        kept_all, waiting = [], []
        for boxes, rows, idx in frames:
            if len(boxes) == 0:
                kept_all.append(np.zeros((0,), np.int64))
            else:
                kept_all.append(np.asarray(filt(boxes=boxes, indices=idx, rows=rows), dtype=np.int64).reshape(-1))
            waiting.append(len(filt))
        gold['s%d_kept' % s] = np.concatenate(kept_all)
        gold['s%d_kept_count' % s] = np.array([len(k) for k in kept_all], np.int64)
        gold['s%d_waiting' % s] = np.array(waiting, np.int64)
    rng = np.random.default_rng(5)
    maps = (rng.random((6, 64, 64)) ** 4).astype(np.float32)
    maps[5] = 0
    pairs = [(0, 1), (0, 0), (2, 3), (4, 5), (5, 5)]
    gold['dice_maps'] = maps
    gold['dice_pairs'] = np.array(pairs, np.int64)
    gold['dice_values'] = np.array([float(dice_coeff(maps[a], maps[b])) for a, b in pairs], np.float64)
    np.savez_compressed(os.path.join(OUT, 'stream_golden.npz'), **gold)
    print({k: (v.shape, v.sum() if v.dtype.kind == 'i' else None) for k, v in gold.items()})


if __name__ == '__main__':
    main()
