"""Box filters (ocr_amd/utils/filters.py) against the reference's own filter_boxes / RegionFilter / SizeFilter run on
the same seeded boxes (tests/golden/filter_golden.npz, made by tests/golden/make_filter_golden.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_filter_golden import CASES  # noqa: E402

from ocr_amd.utils import filters as F  # noqa: E402

GOLD = np.load(os.path.join(HERE, 'golden', 'filter_golden.npz'))


@pytest.mark.parametrize('name,cls,kw', CASES)
def test_filter_matches_reference(name, cls, kw):
    boxes = GOLD['boxes']
    rows = [boxes[i:i + 1] for i in range(len(boxes))]
    kept, idx, kept_rows = F.filter_boxes([getattr(F, cls)(**kw)], boxes, list(range(len(boxes))), rows)
    np.testing.assert_array_equal(np.asarray(idx), GOLD[name + '_indices'])
    np.testing.assert_array_equal(kept, GOLD[name + '_boxes'])
    assert len(kept_rows) == len(idx) and all(np.array_equal(r, boxes[i:i + 1]) for r, i in zip(kept_rows, idx))


def test_chain_empty_and_custom_callable():
    boxes = GOLD['boxes']
    rows = [boxes[i:i + 1] for i in range(len(boxes))]
    chain = [F.RegionFilter(**CASES[0][2]), F.SizeFilter(**CASES[2][2])]
    _, idx, _ = F.filter_boxes(chain, boxes, list(range(len(boxes))), rows)
    np.testing.assert_array_equal(np.asarray(idx), GOLD['chain_indices'])
    empty = np.zeros((0, 4), np.float32)
    assert F.filter_boxes(chain, empty, [], [])[0].shape == (0, 4)
    _, idx, _ = F.filter_boxes(lambda boxes, **_: [0, 2], boxes, list(range(len(boxes))), rows)      # any callable works
    assert idx == [0, 2]
    with pytest.raises(ValueError):
        F.RegionFilter([0, 0, 1, 1], mode='nope')(boxes, None, rows)
