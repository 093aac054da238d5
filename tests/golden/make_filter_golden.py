"""Generates tests/golden/filter_golden.npz with the reference's own box filters
(utils/image/bounding_box/filters.py: filter_boxes, RegionFilter, SizeFilter) on seeded random boxes.

    python tests/golden/make_filter_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ('region_overlap', 'RegionFilter', dict(region=[0.25, 0.25, 0.75, 0.6], mode='overlap', source='xyxy')),
    ('region_center', 'RegionFilter', dict(region=[0.25, 0.25, 0.75, 0.6], mode='center', source='xyxy')),
    ('size_h', 'SizeFilter', dict(min_h=0.05, max_h=0.3)),
    ('size_all', 'SizeFilter', dict(min_w=0.1, max_w=0.5, min_area=0.01, max_area=0.1)),
]


def sample_boxes():
    rng = np.random.default_rng(99)
    xy = rng.uniform(0, 0.8, (60, 2)).astype(np.float32)
    wh = rng.uniform(0.01, 0.5, (60, 2)).astype(np.float32)
    return np.concatenate([xy, np.minimum(xy + wh, 1.0).astype(np.float32)], axis=1)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from utils.image.bounding_box import filters as F
    boxes = sample_boxes()
    rows = [boxes[i:i + 1] for i in range(len(boxes))]
    gold = {'boxes': boxes}
    for name, cls, kw in CASES:
        kept, idx, _ = F.filter_boxes([getattr(F, cls)(**kw)], boxes, list(range(len(boxes))), rows)
        gold[name + '_indices'] = np.asarray(idx, np.int64)
        gold[name + '_boxes'] = kept
    kept, idx, _ = F.filter_boxes([F.RegionFilter(**CASES[0][2]), F.SizeFilter(**CASES[2][2])], boxes, list(range(len(boxes))), rows)
    gold['chain_indices'] = np.asarray(idx, np.int64)
    np.savez(os.path.join(OUT, 'filter_golden.npz'), **gold)
    print({k: v.shape for k, v in gold.items()})


if __name__ == '__main__':
    main()
