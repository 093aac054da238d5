"""The CPU oracle against vectors produced by the reference itself and against the reference's
own known-answer tests (tests/test_utils_boxes.py, tests/test_utils_image.py)."""
import numpy as np
import pytest

from oracle import boxes as obox
from oracle import east_decode, image as oimg, nms as onms
from conftest import canonical_order

N_DECODE = 4


@pytest.mark.parametrize('k', range(N_DECODE))
def test_decode_matches_reference_bit_exact(golden, k):
    m = golden['decode%d_map' % k]
    h, w = m.shape[:2]
    res = east_decode.decode_map(m, (h * 4, w * 4), threshold=float(golden['decode%d_thr' % k]),
                                 trig='numpy32')
    ref = canonical_order(golden['decode%d_polys' % k], golden['decode%d_scores' % k])
    got = canonical_order(res['boxes'], res['scores'])
    assert got.shape == ref.shape
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize('k', range(N_DECODE))
def test_decode_rounded_trig_within_one_ulp(golden, k):
    m = golden['decode%d_map' % k]
    h, w = m.shape[:2]
    a = east_decode.decode_map(m, (h * 4, w * 4), trig='numpy32', return_index=True)
    b = east_decode.decode_map(m, (h * 4, w * 4), trig='rounded64', return_index=True)
    # a validity decision may flip only when a corner sits within rounding of the border
    common, ia, ib = np.intersect1d(a['index'], b['index'], return_indices=True)
    assert len(common) >= max(len(a['index']), len(b['index'])) - 2
    np.testing.assert_allclose(a['boxes'][ia], b['boxes'][ib], rtol=0, atol=2e-6)


@pytest.mark.parametrize('k', range(N_DECODE))
def test_poly_to_xyxy_matches_reference(golden, k):
    got = obox.poly_to_xyxy(golden['decode%d_polys' % k])
    np.testing.assert_array_equal(got, golden['decode%d_xyxy' % k])


def test_iou_matches_reference(golden):
    b = golden['iou_boxes']
    # The reference's NumPy shim for divide_no_nan (utils/keras/ops/numpy.py:58) leaves the 0/0
    # entries uninitialised; on the tensor path Keras' divide_no_nan returns 0 there.  Compare the
    # defined entries bit for bit and require 0 from the oracle on the others.
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    defined = (area[:, None] > 0) | (area[None, :] > 0)
    got = obox.compute_iou(b, as_matrix=True)
    np.testing.assert_array_equal(got[defined], golden['iou_matrix'][defined])
    assert np.all(got[~defined] == 0)
    gotp = obox.compute_iou(b[:100], b[100:], as_matrix=False)
    definedp = (area[:100] > 0) | (area[100:] > 0)
    np.testing.assert_array_equal(gotp[definedp], golden['iou_pairs'][definedp])


def test_iou_known_answers():
    # tests/test_utils_boxes.py:138-197
    box1 = np.array([[100, 101, 200, 201]])
    np.testing.assert_allclose(obox.compute_iou(box1, box1 + 1, as_matrix=False), [0.96097656633], atol=1e-6)
    y_true = np.array([[100, 101, 200, 201], [0, 2, 1, 3], [1300, 1400, 1500, 1401]], dtype='int32')
    y_pred = np.array([[101, 102, 201, 202], [0, 2, 1, 3], [1000, 1400, 1200, 1401]], dtype='int32')
    expected = np.array([[0.96097656633, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float32)
    np.testing.assert_allclose(obox.compute_iou(y_true, y_pred, as_matrix=True), expected, atol=1e-6)


def test_crop_matches_reference(golden):
    shape = tuple(golden['crop_image_shape'])
    image = np.arange(np.prod(shape), dtype=np.int64).reshape(shape)
    for b, ab, shp, s in zip(golden['crop_rel'], golden['crop_abs'], golden['crop_shapes'], golden['crop_sums']):
        pos, crop = obox.crop_box(image, b)
        assert list(pos) == list(ab)
        assert crop.shape == tuple(shp)
        assert int(crop.sum()) == int(s)


def test_crop_known_answers():
    # tests/test_utils_boxes.py:274-289
    image = np.arange(16).reshape(4, 4)
    _, crop = obox.crop_box(image, np.array([0, 1, 1, 3]))
    np.testing.assert_array_equal(crop, image[1:3, :1])
    xyxy = obox.convert_xyxy_xywh(np.array([[0, 1, 1, 3], [2, 2, 1, 2]]), 'xywh', 'xyxy')
    np.testing.assert_array_equal(obox.crop_box(image, xyxy[0])[1], image[1:4, :1])
    np.testing.assert_array_equal(obox.crop_box(image, xyxy[1])[1], image[2:4, 2:3])


def test_box_conversion_known_answers():
    # tests/test_utils_boxes.py:31-47
    rel_xywh = np.array([[0, 0, 1, 1], [0.25, 0.2, 0.1, 0.2], [0.5, 0.5, 0.5, 0.5]], dtype=np.float32)
    rel_xyxy = np.array([[0, 0, 1, 1], [0.25, 0.2, 0.35, 0.4], [0.5, 0.5, 1, 1]], dtype=np.float32)
    np.testing.assert_allclose(obox.convert_xyxy_xywh(rel_xywh, 'xywh', 'xyxy'), rel_xyxy, atol=1e-6)
    np.testing.assert_allclose(obox.convert_xyxy_xywh(rel_xyxy, 'xyxy', 'xywh'), rel_xywh, atol=1e-6)
    np.testing.assert_array_equal(obox.to_absolute(rel_xyxy, 720, 1024),
                                  (rel_xyxy * np.array([[1024, 720, 1024, 720]])).astype(np.int32))


def test_size_rules_match_reference(golden):
    for h, w, out in zip(golden['crnn_size_h'], golden['crnn_size_w'], golden['crnn_size_out']):
        (_, _), got = oimg.crnn_sizes(int(h), int(w))
        assert got == (int(out[0]), int(out[1])), (h, w)
    for (h, w), out in zip(golden['east_size_in'], golden['east_size_out']):
        assert oimg.east_size(int(h), int(w)) == (int(out[0]), int(out[1]))


def test_intermediate_height_63_cases():
    # float32 size arithmetic gives a 63-row resize for these crop heights (8..199)
    expected = {41, 47, 55, 61, 82, 83, 94, 97, 107, 109, 110, 115, 121, 122, 123, 141, 159, 164,
                165, 166, 188, 189, 194}
    got = set()
    for h in range(8, 200):
        # padded width chosen so that the height ratio 64/h is the binding one
        ih = int(oimg.intermediate_size((h, 10 * h), (64, 704))[0])
        assert ih in (63, 64)
        if ih == 63:
            got.add(h)
    assert got == expected


def test_normalisations_match_reference(golden):
    np.testing.assert_array_equal(oimg.normalize_mean_std(golden['norm_x3']), golden['norm_east'])
    np.testing.assert_array_equal(oimg.normalize_mean_std(golden['norm_x1']), golden['norm_easyocr'])
    # tests/test_utils_image.py:139-143
    x = np.random.default_rng(0).uniform(size=(4, 4, 3)).astype(np.float32)
    np.testing.assert_allclose(oimg.normalize_mean_std(x), (x - 0.5) / 0.5, atol=1e-7)


# ------------------------------------------------------------------ NMS known answers
# tests/test_utils_boxes.py:559-637

CHAIN = np.array([[0, 0, 0.2, 0.2], [0.1, 0.1, 0.3, 0.3], [0.2, 0.2, 0.4, 0.4], [0.3, 0.3, 0.5, 0.5]],
                 dtype='float32')


def test_standard_nms_known_answer():
    boxes, _, valid, _ = onms.fast_nms(CHAIN, nms_threshold=0.1)
    np.testing.assert_array_equal(boxes[valid], CHAIN[[0, 2]])


@pytest.mark.parametrize('kwargs,expected', [
    (dict(nms_threshold=0.1, merge_threshold=0.1), [[0, 0, 0.3, 0.3], [0.2, 0.2, 0.5, 0.5]]),
    (dict(nms_threshold=0.1, merge_threshold=0.1, merge_method='average'),
     [[0.05, 0.05, 0.25, 0.25], [0.25, 0.25, 0.45, 0.45]]),
    (dict(nms_threshold=0.01, merge_threshold=0.1), [[0, 0, 0.3, 0.3]]),
    (dict(nms_threshold=0.1, merge_threshold=0.01), [[0, 0, 0.5, 0.5]]),
])
def test_lanms_known_answers(kwargs, expected):
    boxes, _, valid, _, _ = onms.lanms(CHAIN, **kwargs)
    np.testing.assert_allclose(boxes[valid], np.array(expected, dtype='float32'), atol=1e-6)


def test_matrix_nms_equals_greedy_scan():
    rng = np.random.default_rng(5)
    for trial in range(150):
        n = int(rng.integers(1, 60))
        xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
        wh = rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)
        b = np.concatenate([xy, xy + wh], axis=1)
        thr = float(rng.choice([0.1, 0.2, 0.3, 0.5]))
        sel = onms.self_suppression(b, np.float32(thr))
        np.testing.assert_array_equal(onms.get_valid_mask(sel), onms.greedy_nms_keep(b, np.float32(thr)))


# ------------------------------------------------------------------ resize restatement

@pytest.mark.parametrize('hw,out', [((30, 100), (64, 213)), ((90, 333), (64, 236)), ((64, 50), (64, 50)),
                                    ((17, 23), (63, 85)), ((200, 40), (64, 12)), ((33, 500), (64, 969))])
def test_aa_resize_numpy_matches_torch(hw, out):
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    img = rng.uniform(0, 1, hw + (1,)).astype(np.float32)
    a = oimg.aa_resize_numpy(img, *out)
    b = oimg.aa_resize_torch(img, *out)
    np.testing.assert_allclose(a, b, rtol=0, atol=3e-7)


def test_pad_repeat_last():
    # shape / value rule of pad_image(pad_mode='repeat_last') (image_processing.py:125-134)
    img = np.arange(6, dtype=np.float32).reshape(2, 3, 1)
    out = oimg.pad_repeat_last(img, 4, 5)
    assert out.shape == (4, 5, 1)
    np.testing.assert_array_equal(out[:2, 3:, 0], np.array([[2, 2], [5, 5]], dtype=np.float32))
    np.testing.assert_array_equal(out[2:, :, 0], np.tile(out[1:2, :, 0], (2, 1)))


# ------------------------------------------------------------------ the C LANMS against the pinned NumPy one
# oracle/lanms_c.c serves the page-scale checks (thousands of candidates); it is itself held to oracle/nms.py, which the
# reference's known answers above pin: same boxes, mask, order and iteration count.

def _same_as_numpy_lanms(boxes, scores, **kwargs):
    from oracle import lanms_c
    ref = onms.lanms(boxes, scores, **kwargs)
    got = lanms_c.lanms(boxes, scores, **kwargs)
    if kwargs.get('merge_method', 'union') == 'union':
        np.testing.assert_array_equal(got[0], ref[0])               # min / max only: exact
    else:
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=2e-6)
    if scores is not None:
        np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    np.testing.assert_array_equal(got[3], ref[3])
    assert got[4] == ref[4]
    return ref


@pytest.mark.parametrize('kwargs', [
    dict(nms_threshold=0.1, merge_threshold=0.1), dict(nms_threshold=0.1, merge_threshold=0.1, merge_method='average'),
    dict(nms_threshold=0.01, merge_threshold=0.1), dict(nms_threshold=0.1, merge_threshold=0.01)])
def test_c_lanms_equals_numpy_lanms_on_the_reference_known_answers(kwargs):
    _same_as_numpy_lanms(CHAIN, None, **kwargs)


@pytest.mark.parametrize('seed,method', [(21, 'union'), (22, 'union'), (21, 'average')])
def test_c_lanms_equals_numpy_lanms_on_decoded_pages(seed, method):
    """Two synthetic 512 x 512 pages decoded by the oracle (more than 1000 candidates each), the detector's defaults
    (merge 0.3, NMS 0.2: both passes run)."""
    from ocr_amd import synth
    from oracle import boxes as obox, east_decode as odec
    _, maps, _ = synth.make_pages(seed, 1, 512, 512, 30)
    dec = odec.decode_map(maps[0], (512, 512), threshold=0.35)
    boxes, scores = obox.poly_to_xyxy(dec['boxes']), dec['scores']
    assert len(boxes) > 1000
    ref = _same_as_numpy_lanms(boxes, scores, nms_threshold=0.2, merge_threshold=0.3, merge_method=method)
    assert 10 <= int(ref[2].sum()) <= 60 and ref[4] >= 2


def test_batched_cpu_pipeline_gives_the_per_row_pipeline_strings():
    """bench.py reports a second CPU figure: the oracle pipeline batched the way a CPU implementation would (one
    recogniser forward per crop width).  Same arithmetic, so the same rows, strings and scores as the reference-faithful
    one-forward-per-row path."""
    from ocr_amd import synth
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
    from oracle import ctc as octc, nets, pipeline as opipe
    ep, cp = nets.init_east_params(seed=3), nets.init_crnn_params(vocab_size=97, seed=4)
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    images, maps, _ = synth.make_pages(5, 2, 128, 256, 8)
    per_row = [opipe.ocr_page(images[b], ep, cp, tok, maps_override=maps[b], run_network=False) for b in range(2)]
    batched = opipe.ocr_pages_batched(images, ep, cp, tok, maps_override=maps, run_network=False)
    assert sum(len(p['texts']) for p in per_row) >= 6
    for a, b in zip(per_row, batched):
        assert list(a['index']) == list(b['index']) and list(a['texts']) == list(b['texts'])
        np.testing.assert_array_equal(a['boxes'], b['boxes'])
        np.testing.assert_allclose(a['scores'], b['scores'], rtol=1e-4, atol=1e-5)
