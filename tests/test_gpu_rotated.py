"""Opt-in rotated extensions (not in the reference: parity is against oracle/rotated.py, itself pinned by known
answers and by the unrotated case, which must equal the reference-faithful axis-aligned path bit for bit)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def ctx():
    from ocr_amd.engine import HipContext
    return HipContext(0)


def _rot_rect(cx, cy, w, h, ang):
    c, s = math.cos(ang), math.sin(ang)
    pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    return (pts @ np.array([[c, s], [-s, c]]) + [cx, cy]).astype(np.float32)


def _nms_gpu(ctx, quads_list, scores_list, thr, cap, max_out=256):
    B = len(quads_list)
    polys = np.zeros((B, cap, 4, 2), np.float32)
    scores = np.zeros((B, cap), np.float32)
    counts = np.array([len(q) for q in quads_list], np.int32)
    for b in range(B):
        polys[b, :counts[b]] = quads_list[b]
        scores[b, :counts[b]] = scores_list[b]
    out = ctx.rotated_nms(torch.from_numpy(polys).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts).cuda(),
                          thr, max_out=max_out)
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_rotated_iou_known_answers_through_the_kernel(ctx):
    from oracle.rotated import quad_iou
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    half = sq + np.float32([0.5, 0])                                   # IoU 1/3
    diamond = _rot_rect(0.5, 0.5, 1, 1, math.pi / 4)                    # IoU 1/sqrt(2)
    assert quad_iou(sq, sq) == 1.0 and quad_iou(sq, sq[::-1]) == 1.0 and quad_iou(sq, sq + np.float32([1, 0])) == 0.0
    assert abs(quad_iou(sq, half) - 1 / 3) < 1e-12 and abs(quad_iou(sq, diamond) - 1 / math.sqrt(2)) < 1e-6
    for other, iou in ((half, 1 / 3), (diamond, 1 / math.sqrt(2)), (sq[::-1].copy(), 1.0)):
        for thr, n_kept in ((iou - 1e-3, 1), (iou + 1e-3, 2)):
            if thr > 1:
                continue
            out = _nms_gpu(ctx, [np.stack([sq, other])], [np.float32([0.9, 0.8])], thr, cap=8)
            assert out['counts'][0] == n_kept, (iou, thr)
            np.testing.assert_array_equal(out['polys'][0, 0], sq)
    # touching squares and far squares never suppress each other; equal scores: the lower index is taken first
    out = _nms_gpu(ctx, [np.stack([sq + np.float32([3, 0]), sq, sq + np.float32([1, 0])])], [np.float32([0.5, 0.5, 0.5])], 0.01, cap=4)
    assert out['counts'][0] == 3 and list(out['index'][0, :3]) == [0, 1, 2]


def test_rotated_nms_matches_the_oracle(ctx):
    from oracle.rotated import rotated_nms
    rng = np.random.default_rng(3)
    quads_list, scores_list = [], []
    for n_words in (40, 7, 0):
        quads = []
        for _ in range(n_words):
            cx, cy, w, h = rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.05, 0.2), rng.uniform(0.02, 0.05)
            ang = rng.uniform(-0.6, 0.6)
            for _ in range(int(rng.integers(20, 60))):
                quads.append(_rot_rect(cx + rng.normal(0, 0.004), cy + rng.normal(0, 0.004), w * rng.uniform(0.9, 1.1),
                                       h * rng.uniform(0.9, 1.1), ang + rng.normal(0, 0.03)))
        quads = np.array(quads, np.float32).reshape(-1, 4, 2)
        perm = rng.permutation(len(quads))
        quads_list.append(quads[perm])
        scores_list.append(rng.permutation(len(quads)).astype(np.float32) / max(1, len(quads)) * 0.6 + 0.35)
    cap = max(len(q) for q in quads_list) + 5
    thr = np.float32(0.2)
    out = _nms_gpu(ctx, quads_list, scores_list, float(thr), cap, max_out=512)
    for b, (q, s) in enumerate(zip(quads_list, scores_list)):
        want = rotated_nms(q, s, float(thr))
        assert out['counts'][b] == len(want), b
        np.testing.assert_array_equal(out['index'][b, :len(want)], want)
        np.testing.assert_array_equal(out['polys'][b, :len(want)], q[want])
        np.testing.assert_array_equal(out['scores'][b, :len(want)], s[want])
    assert out['counts'][0] >= 40 and out['counts'][2] == 0
    # max_out cuts the list, it does not change its head
    cut = _nms_gpu(ctx, quads_list[:1], scores_list[:1], float(thr), cap, max_out=10)
    assert cut['counts'][0] == 10
    np.testing.assert_array_equal(cut['index'][0], out['index'][0, :10])


def _pack_rotated(ctx, image, quads_abs):
    from ocr_amd.utils.image_sizes import plan_crops_rotated
    # pixel coordinates go in as they are (scale 1): a round trip through relative coordinates would move integer
    # corners by an ulp
    rel = np.asarray(quads_abs, np.float32).reshape(-1, 4, 2)
    plan = plan_crops_rotated(rel, 1, 1)
    off = np.concatenate([[0], np.cumsum(64 * plan['out_w'].astype(np.int64))])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    packed = ctx.crop_rotated_pack(d(image)[None], d(plan['geom']), d(np.zeros(len(off) - 1, np.int32)), d(plan['inter']),
                                   d(plan['out_w']), d(off[:-1].astype(np.int64)), int(off[-1])).cpu().numpy()
    return plan, [packed[off[k]:off[k + 1]].reshape(64, -1) for k in range(len(off) - 1)], rel


def test_unrotated_rectangle_is_the_axis_aligned_crop(ctx):
    from ocr_amd.utils.image_sizes import plan_crops
    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)
    rects = np.array([[10, 20, 150, 52], [0, 0, 300, 200], [33, 100, 60, 190], [200, 10, 290, 24], [5, 5, 6, 6]], np.int64)
    quads = np.array([[[x0, y0], [x1, y0], [x1, y1], [x0, y1]] for x0, y0, x1, y1 in rects], np.float32)
    plan_r, got, _ = _pack_rotated(ctx, image, quads)
    rel = (rects / np.array([300, 200, 300, 200])).astype(np.float32)
    plan_a = plan_crops(rel, 200, 300)
    np.testing.assert_array_equal(plan_a['rects'], rects[plan_a['index']])       # the relative boxes resolve to these slices
    np.testing.assert_array_equal(plan_r['out_w'], plan_a['out_w'])
    np.testing.assert_array_equal(plan_r['inter'], plan_a['inter'])
    off = np.concatenate([[0], np.cumsum(64 * plan_a['out_w'].astype(np.int64))])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    ref = ctx.crop_resize_pack(d(image)[None], d(plan_a['rects']), d(np.zeros(len(rects), np.int32)), d(plan_a['inter']),
                               d(plan_a['out_w']), d(off[:-1].astype(np.int64)), int(off[-1])).cpu().numpy()
    for k in range(len(rects)):
        np.testing.assert_array_equal(got[k].reshape(-1), ref[off[k]:off[k + 1]])


def test_rotated_crop_matches_the_oracle(ctx):
    from oracle.rotated import prepare_crnn_input_rotated
    rng = np.random.default_rng(6)
    image = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    image[60:180, 40:280] = np.repeat(rng.integers(0, 256, (15, 30, 3), dtype=np.uint8), 8, axis=0).repeat(8, axis=1)
    quads = [_rot_rect(160, 120, 150, 30, a) for a in (0.0, 0.17, -0.4, math.pi / 2, 2.5)]
    quads += [_rot_rect(10, 10, 60, 30, 0.3), _rot_rect(315, 230, 90, 12, -0.2),       # partly outside the page
              _rot_rect(160, 120, 300, 100, 0.05), _rot_rect(100, 100, 9.4, 40.2, 0.7),  # shrunk 1.5x; taller than wide
              _rot_rect(200, 50, 3, 2, 0.1)]
    # a sheared quadrilateral: the parallelogram of its top and left edge is what is cropped
    quads.append(np.float32([[50, 50], [150, 60], [170, 95], [60, 80]]))
    plan, got, rel = _pack_rotated(ctx, image, quads)
    assert len(plan['index']) == len(quads)
    for k, q in enumerate(rel):
        want = prepare_crnn_input_rotated(image, q)
        assert want is not None and got[k].shape == want.shape[:2], k
        np.testing.assert_array_equal(got[k], want[..., 0], err_msg='quad %d' % k)
    # non-finite and absurd quadrilaterals are dropped on the host, never sent to the kernel
    from ocr_amd.utils.image_sizes import plan_crops_rotated
    bad = np.float32([[[0, 0], [np.nan, 0], [1, 1], [0, 1]], [[0, 0], [1e6, 0], [1e6, 1], [0, 1]]])
    assert len(plan_crops_rotated(bad, 240, 320)['index']) == 0


def test_rotated_pipeline_matches_the_oracle(ctx):
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_crnn_params, random_east_params
    from oracle import ctc as octc, east_decode, nets
    from oracle.rotated import prepare_crnn_input_rotated, rotated_nms
    ep, cp = random_east_params(5), random_crnn_params(97, seed=6)
    pipe = OcrPipeline(ep, cp, device=0, rotated=True)
    H, W = 256, 384
    images, maps, _ = synth.make_pages(91, 2, H, W, 16, rotated_frac=1.0, max_angle_deg=12.0)
    res = pipe(torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda())
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    for b in range(2):
        dec = east_decode.decode_map(maps[b], (H, W), threshold=0.35, trig='rounded64')
        order = rotated_nms(dec['boxes'], dec['scores'], float(np.float32(0.2)))
        want = dec['boxes'][order]
        assert len(want) >= 8
        np.testing.assert_array_equal(res[b]['boxes'], want)
        texts, logits = [], []
        for q in want:
            inp = prepare_crnn_input_rotated(images[b], q * np.float32([W, H]))
            lg = nets.crnn_forward(cp, inp[None])[0].numpy()
            logits.append(lg)
            texts.append(tok.ctc_decode(lg[None])[0][0])
        assert all(res[b]['valid'])
        octc.assert_texts_match_or_tie(res[b]['texts'], texts, logits, tok)
    # at least one word of the pages really is rotated by more than a couple of degrees
    q = res[0]['boxes']
    ang = np.degrees(np.arctan2(q[:, 1, 1] - q[:, 0, 1], (q[:, 1, 0] - q[:, 0, 0]) * W / H))
    assert np.abs(ang).max() > 3


def test_rotated_through_the_model_classes(ctx):
    """`EAST.decode_output(nms_method='rotated')` -> 'poly' boxes -> `BaseOCR.infer` crops the quadrilaterals."""
    from ocr_amd import synth
    from ocr_amd.models.detection import EAST
    from ocr_amd.models.ocr import CRNN
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.utils.filters import SizeFilter
    from ocr_amd.weights import random_crnn_params, random_east_params
    ep, cp = random_east_params(5), random_crnn_params(97, seed=6)
    east, crnn = EAST(ep, name='east_rot'), CRNN(weights=cp, name='crnn_rot')
    H, W = 256, 384
    images, maps, _ = synth.make_pages(92, 1, H, W, 14, rotated_frac=1.0, max_angle_deg=12.0)
    boxes = east.decode_output(maps[0], inputs=np.zeros((1, H, W, 3), np.float32), nms_method='rotated')
    assert boxes['format'] == 'poly' and boxes['boxes'].shape[1:] == (4, 2) and len(boxes['boxes']) >= 8
    res = crnn.infer({'image': images[0], 'boxes': boxes}, method='greedy')
    ref = OcrPipeline(ep, cp, device=0, rotated=True)(torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda())[0]
    np.testing.assert_array_equal(boxes['boxes'], ref['boxes'])
    assert [r['text'] for r in res['ocr']] == list(ref['texts'])
    for r, q in zip(res['ocr'], boxes['boxes']):
        np.testing.assert_array_equal(r['quad'], q)
        np.testing.assert_array_equal(r['boxes'], np.concatenate([q.min(axis=0), q.max(axis=0)]))
    # filters see the axis-aligned bounds and the surviving quadrilaterals are the ones recognised
    big = SizeFilter(min_w=0.15)
    some = crnn.infer({'image': images[0], 'boxes': boxes}, method='greedy', box_filters=[big])
    keep = [i for i, q in enumerate(boxes['boxes']) if q[:, 0].max() - q[:, 0].min() >= 0.15]
    assert 0 < len(keep) < len(boxes['boxes'])
    assert [r['text'] for r in some['ocr']] == [res['ocr'][i]['text'] for i in keep]
    with pytest.raises(NotImplementedError):
        crnn.infer({'image': images[0], 'boxes': boxes}, method='greedy', dezoom_factor=1.2)
