"""Word -> line -> paragraph grouping against vectors produced by the reference's own functions and
against what the reference's tests assert (tests/test_utils_boxes.py:380-557)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from ocr_amd.utils.combination import combine_boxes, combine_boxes_horizontal, combine_boxes_vertical

GOLD = dict(np.load(os.path.join(GOLDEN_DIR, 'combine_vectors.npz')))
NAMES = [str(n) for n in GOLD['names']]


def _norm(groups):
    return json.loads(json.dumps(groups, default=lambda o: int(o) if isinstance(o, np.integer) else list(o)))


@pytest.mark.parametrize('name', NAMES)
def test_grouping_matches_reference(name):
    b = GOLD[name + '/boxes']
    hb, hg, _ = combine_boxes_horizontal(b.copy())
    assert _norm(hg) == json.loads(str(GOLD[name + '/h_groups']))
    np.testing.assert_array_equal(hb, GOLD[name + '/h_boxes'])
    vb, vg, _ = combine_boxes_vertical(np.array(hb).copy(), indices=hg)
    assert _norm(vg) == json.loads(str(GOLD[name + '/hv_groups']))
    np.testing.assert_array_equal(vb, GOLD[name + '/hv_boxes'])
    cb, cg, rows = combine_boxes({'boxes': b.copy(), 'format': 'xyxy'})
    assert _norm(cg) == json.loads(str(GOLD[name + '/c_groups']))
    np.testing.assert_array_equal(cb, GOLD[name + '/c_boxes'])
    ref_rows = json.loads(str(GOLD[name + '/c_rows']))
    assert len(rows) == len(ref_rows)
    for r, rr in zip(rows, ref_rows):
        np.testing.assert_array_equal(np.asarray(r), np.asarray(rr, dtype=np.asarray(r).dtype))


@pytest.mark.parametrize('name', [n for n in NAMES if n.startswith('test_')])
def test_reference_known_answers(name):
    b = GOLD[name + '/boxes']
    kw = {'h_factor': 1.} if name == 'test_with_lots_of_boxes' else {}      # as the reference's test calls it
    hb, hg, _ = combine_boxes_horizontal(b.copy(), **kw)
    assert _norm(hg) == json.loads(str(GOLD[name + '/target_groups_h']))
    np.testing.assert_allclose(hb, np.array(json.loads(str(GOLD[name + '/target_boxes_h']))), atol=1e-4)
    vb, vg, _ = combine_boxes_vertical(np.array(hb), indices=hg)
    assert _norm(vg) == json.loads(str(GOLD[name + '/target_groups_hv']))
    np.testing.assert_allclose(vb, np.array(json.loads(str(GOLD[name + '/target_boxes_hv']))), atol=1e-4)


def test_degenerate_inputs():
    one = np.array([[0.1, 0.1, 0.2, 0.2]], np.float32)
    b, idx, rows = combine_boxes(one)
    assert len(b) == 1 and _norm(idx) in ([[[0]]], [[0]], [0]) and len(rows) == 1
