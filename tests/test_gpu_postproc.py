"""GPU parity of the bandwidth / latency bound stages (decode, LANMS, crop-pack, CTC) against the
CPU oracle, through the C ABI.  Integer / index results and every float that comes out of a fixed
sequence of float32 (or float64) operations must match bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import boxes as obox, ctc as octc, east_decode, image as oimg, lanms_c, nms as onms  # noqa: E402
from conftest import canonical_order  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from ocr_amd.engine import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _gpu_decode(ctx, maps, in_hw, thr=0.35, cap=None):
    out = ctx.east_decode(_dev(maps), in_hw, thr, cap=cap, with_polys=True, with_pix=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


# ----------------------------------------------------------------------------- decode

@pytest.mark.parametrize('k', range(4))
def test_decode_golden_maps(ctx, golden, k):
    m = golden['decode%d_map' % k]
    h, w = m.shape[:2]
    got = _gpu_decode(ctx, m[None], (h * 4, w * 4))
    n = int(got['counts'][0])
    ref = east_decode.decode_map(m, (h * 4, w * 4), trig='rounded64', return_index=True)
    assert n == len(ref['boxes'])
    np.testing.assert_array_equal(got['pix'][0, :n], ref['index'])
    np.testing.assert_array_equal(got['polys'][0, :n], ref['boxes'])
    np.testing.assert_array_equal(got['scores'][0, :n], ref['scores'])
    np.testing.assert_array_equal(got['xyxy'][0, :n], obox.poly_to_xyxy(ref['boxes']))
    # Against the reference's own output.  Its sin/cos are NumPy float32 loops, at most 1 ulp away from the
    # correctly rounded values used here, so a box whose corner sits within rounding of the page border may pass
    # the validity filter on one side only.  The oracle's 'numpy32' mode IS the reference's output bit for bit
    # (asserted here as a set) and carries the map pixel of every box: boxes are matched by pixel, the symmetric
    # difference is at most 2 boxes, and EVERY common box is compared.
    gold = east_decode.decode_map(m, (h * 4, w * 4), trig='numpy32', return_index=True)
    np.testing.assert_array_equal(canonical_order(gold['boxes'], gold['scores']),
                                  canonical_order(golden['decode%d_polys' % k], golden['decode%d_scores' % k]))
    common, ig, ir = np.intersect1d(got['pix'][0, :n], gold['index'], return_indices=True)
    assert (n - len(common)) + (len(gold['index']) - len(common)) <= 2
    assert len(common) >= 0.99 * len(gold['index']) > 0
    np.testing.assert_allclose(got['polys'][0, :n][ig], gold['boxes'][ir], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(got['scores'][0, :n][ig], gold['scores'][ir])
    np.testing.assert_allclose(got['xyxy'][0, :n][ig], golden_xyxy_by_pixel(golden, k, gold)[ir], rtol=0, atol=2e-6)


def golden_xyxy_by_pixel(golden, k, gold):
    """The reference's own poly -> xyxy output re-ordered to the order of `gold` (oracle 'numpy32' decode, equal to the
    reference's polygons as a set): rows are matched on the polygon + score bits."""
    key = lambda polys, scores: [p.tobytes() + s.tobytes() for p, s in zip(polys.reshape(len(polys), -1), scores)]
    where = {kk: i for i, kk in enumerate(key(golden['decode%d_polys' % k], golden['decode%d_scores' % k]))}
    rows = [where[kk] for kk in key(gold['boxes'], gold['scores'])]
    return golden['decode%d_xyxy' % k][rows]


def test_decode_batched_pages_and_edge_cases(ctx):
    from ocr_amd import synth
    _, maps, _ = synth.make_pages(11, 3, 512, 768, 90)
    maps[2] = 0.01                                     # an empty page
    got = _gpu_decode(ctx, maps, (512, 768))
    for b in range(3):
        ref = east_decode.decode_map(maps[b], (512, 768), trig='rounded64', return_index=True)
        n = int(got['counts'][b])
        assert n == len(ref['boxes'])
        np.testing.assert_array_equal(got['pix'][b, :n], ref['index'])
        np.testing.assert_array_equal(got['polys'][b, :n], ref['boxes'])
        np.testing.assert_array_equal(got['xyxy'][b, :n], obox.poly_to_xyxy(ref['boxes']))
    assert int(got['counts'][2]) == 0
    # capacity smaller than the number of boxes: the count is still reported, the first `cap` are stored
    ref0 = east_decode.decode_map(maps[0], (512, 768), trig='rounded64')
    small = _gpu_decode(ctx, maps[:1], (512, 768), cap=100)
    assert int(small['counts'][0]) == len(ref0['boxes']) > 100
    np.testing.assert_array_equal(small['polys'][0], ref0['boxes'][:100])


def test_decode_rejects_bad_arguments(ctx):
    maps = torch.zeros((1, 8, 8, 6), dtype=torch.float32, device='cuda')
    with pytest.raises(ValueError):
        ctx.east_decode(maps, (30, 32))                # not a multiple of the map size


# ----------------------------------------------------------------------------- LANMS

CHAIN = np.array([[0, 0, 0.2, 0.2], [0.1, 0.1, 0.3, 0.3], [0.2, 0.2, 0.4, 0.4], [0.3, 0.3, 0.5, 0.5]],
                 dtype='float32')


def _gpu_lanms(ctx, boxes_list, scores_list, **kw):
    B = len(boxes_list)
    cap = max(1, max(len(b) for b in boxes_list))
    xy = np.zeros((B, cap, 4), np.float32)
    sc = np.zeros((B, cap), np.float32)
    cnt = np.zeros((B,), np.int32)
    for i, b in enumerate(boxes_list):
        xy[i, :len(b)] = b
        if scores_list is not None:
            sc[i, :len(b)] = scores_list[i]
        cnt[i] = len(b)
    out = ctx.lanms(_dev(xy), _dev(sc) if scores_list is not None else None, _dev(cnt), full=True, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize('kwargs,expected', [
    (dict(nms_threshold=0.1, merge_threshold=0.1), [[0, 0, 0.3, 0.3], [0.2, 0.2, 0.5, 0.5]]),
    (dict(nms_threshold=0.1, merge_threshold=0.1, merge_method='average'),
     [[0.05, 0.05, 0.25, 0.25], [0.25, 0.25, 0.45, 0.45]]),
    (dict(nms_threshold=0.01, merge_threshold=0.1), [[0, 0, 0.3, 0.3]]),
    (dict(nms_threshold=0.1, merge_threshold=0.01), [[0, 0, 0.5, 0.5]]),
])
def test_lanms_reference_known_answers(ctx, kwargs, expected):
    # utils tests of the reference: tests/test_utils_boxes.py:586-637 (scores=None: rank = input order)
    got = _gpu_lanms(ctx, [CHAIN], None, **kwargs)
    n = int(got['counts'][0])
    np.testing.assert_allclose(got['boxes'][0, :n], np.array(expected, dtype='float32'), atol=1e-6)


def _compare_with_oracle(got, b, boxes, scores, oracle_fn, exact=True, **kw):
    ob, osc, omask, order, n_iter = oracle_fn(boxes, scores, None, kw['nms_threshold'],
                                              merge_threshold=kw['merge_threshold'],
                                              merge_method=kw.get('merge_method', 'union'))
    n = int(got['counts'][b])
    assert n == int(omask.sum())
    assert int(got['n_iter'][b]) == n_iter
    cmp = np.testing.assert_array_equal if exact else (lambda x, y: np.testing.assert_allclose(x, y, rtol=0, atol=2e-6))
    cmp(got['boxes'][b, :n], ob[omask])
    np.testing.assert_array_equal(got['src'][b, :n], order[omask])
    if scores is not None:
        np.testing.assert_array_equal(got['scores'][b, :n], osc[omask])
    # input-order outputs: merged boxes and the valid mask, un-sorted
    m = len(boxes)
    merged_in = np.zeros((m, 4), np.float32)
    merged_in[order] = ob
    keep_in = np.zeros((m,), bool)
    keep_in[order] = omask
    cmp(got['merged'][b, :m], merged_in)
    np.testing.assert_array_equal(got['keep'][b, :m].astype(bool), keep_in)


def test_lanms_random_boxes_match_oracle(ctx):
    rng = np.random.default_rng(3)
    for trial in range(30):
        B = 4
        bl, sl = [], []
        for _ in range(B):
            n = int(rng.integers(0, 150))
            xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
            wh = rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)
            bl.append(np.concatenate([xy, xy + wh], 1))
            sl.append(rng.permutation(n).astype(np.float32) / max(n, 1) * 0.6 + 0.36)
        method = 'union' if trial % 3 else 'average'
        kw = dict(merge_threshold=float(rng.choice([0.1, 0.3, 0.5])), nms_threshold=float(rng.choice([0.05, 0.2, 0.6])),
                  merge_method=method)
        got = _gpu_lanms(ctx, bl, sl, **kw)
        for b in range(B):
            _compare_with_oracle(got, b, bl[b], sl[b], onms.lanms, exact=(method == 'union'), **kw)


def test_lanms_weighted_merge_extension(ctx):
    """merge_method='weighted' (not in the reference: the score-weighted merge of the original EAST LANMS inside the
    reference's iteration): a hand-computed case, and random sets against the oracle's float64 statement of it."""
    b = np.array([[0.10, 0.10, 0.50, 0.30], [0.12, 0.10, 0.54, 0.30], [0.70, 0.70, 0.90, 0.80]], np.float32)
    s = np.array([0.9, 0.3, 0.5], np.float32)
    got = _gpu_lanms(ctx, [b], [s], merge_threshold=0.3, nms_threshold=0.2, merge_method='weighted')
    assert int(got['counts'][0]) == 2
    want = (0.9 * b[0].astype(np.float64) + 0.3 * b[1].astype(np.float64)) / 1.2
    np.testing.assert_allclose(got['boxes'][0, 0], want, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(got['boxes'][0, 1], b[2])               # a box that absorbs nothing stays bit for bit
    rng = np.random.default_rng(12)
    for trial in range(10):
        bl, sl = [], []
        for _ in range(3):
            n = int(rng.integers(1, 150))
            xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
            bl.append(np.concatenate([xy, xy + rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)], 1))
            sl.append(rng.permutation(n).astype(np.float32) / max(n, 1) * 0.6 + 0.36)
        kw = dict(merge_threshold=float(rng.choice([0.3, 0.5])), nms_threshold=float(rng.choice([0.2, 0.6])), merge_method='weighted')
        got = _gpu_lanms(ctx, bl, sl, **kw)
        for i in range(3):
            _compare_with_oracle(got, i, bl[i], sl[i], onms.lanms, exact=False, **kw)


def test_lanms_score_ties_use_input_order(ctx):
    rng = np.random.default_rng(8)
    xy = rng.uniform(0, 0.8, (80, 2)).astype(np.float32)
    b = np.concatenate([xy, xy + rng.uniform(0.05, 0.3, (80, 2)).astype(np.float32)], 1)
    s = np.repeat(np.linspace(0.9, 0.4, 8), 10).astype(np.float32)       # heavy ties
    kw = dict(merge_threshold=0.3, nms_threshold=0.2)
    got = _gpu_lanms(ctx, [b], [s], **kw)
    _compare_with_oracle(got, 0, b, s, onms.lanms, **kw)


def test_lanms_decoded_pages_match_c_oracle(ctx):
    from ocr_amd import synth
    _, maps, _ = synth.make_pages(21, 2, 512, 512, 70)
    dec = ctx.east_decode(_dev(maps), (512, 512), 0.35)
    out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], merge_threshold=0.3, nms_threshold=0.2, full=True)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    xy, sc, cnt = dec['xyxy'].cpu().numpy(), dec['scores'].cpu().numpy(), dec['counts'].cpu().numpy()
    kw = dict(merge_threshold=0.3, nms_threshold=0.2)
    for b in range(2):
        n = int(cnt[b])
        assert n > 1000
        _compare_with_oracle(got, b, xy[b, :n], sc[b, :n], lanms_c.lanms, **kw)
        assert 20 <= int(got['counts'][b]) <= 200      # roughly one box per rendered word


@pytest.mark.parametrize('method', ['average', 'weighted'])
def test_lanms_order_dependent_merges_on_a_decoded_page(ctx, method):
    """The first iteration over thousands of candidates runs as separate launches; the merges that sum in list order
    ('average', and the 'weighted' extension) keep a target tile per wave there (the union merge shares a tile between
    waves).  One decoded 512x512 page against the NumPy oracle."""
    from ocr_amd import synth
    _, maps, _ = synth.make_pages(23, 1, 512, 512, 60)
    dec = ctx.east_decode(_dev(maps), (512, 512), 0.35)
    kw = dict(merge_threshold=0.3, nms_threshold=0.2, merge_method=method)
    out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], full=True, **kw)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items()}
    n = int(dec['counts'].cpu().numpy()[0])
    assert n > 1000
    _compare_with_oracle(got, 0, dec['xyxy'].cpu().numpy()[0, :n], dec['scores'].cpu().numpy()[0, :n], onms.lanms, exact=False, **kw)


def test_lanms_rejects_non_positive_threshold(ctx):
    with pytest.raises(ValueError):
        _gpu_lanms(ctx, [CHAIN], None, merge_threshold=0.0, nms_threshold=0.2)


# ----------------------------------------------------------------------------- crop / resize / pack

def test_crop_resize_pack_matches_oracle(ctx):
    from ocr_amd.utils import image_sizes as S
    rng = np.random.default_rng(4)
    B, H, W = 2, 300, 420
    images = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    rel = []
    for _ in range(40):
        x0, y0 = rng.uniform(0, 0.8), rng.uniform(0, 0.8)
        w, h = rng.uniform(0.03, 0.6), rng.uniform(0.03, 0.35)
        rel.append([x0, y0, min(x0 + w, 1.02), min(y0 + h, 1.01)])
    rel += [[0.0, 0.0, 1.0, 1.0], [-0.002, 0.2, 0.3, 0.4213], [0.2, 0.2, 0.2 + 64.0 / W, 0.2 + 64.0 / H],
            [0.1, 0.1, 0.1 + 128.5 / W, 0.1 + 64.2 / H], [-0.2, 0.1, 0.4, 0.6], [0.5, 0.5, 0.5, 0.9]]
    rel = np.array(rel, dtype=np.float32)
    box_img = (np.arange(len(rel)) % B).astype(np.int32)
    plans = S.plan_crops(rel, H, W)
    assert len(plans['index']) == len(rel) - 2            # the wrapped and the zero-width box give no input
    idx = plans['index']
    out_w = plans['out_w']
    off = np.concatenate([[0], np.cumsum(64 * out_w.astype(np.int64))])
    packed = ctx.crop_resize_pack(_dev(images), _dev(plans['rects']), _dev(box_img[idx]), _dev(plans['inter']),
                                  _dev(out_w), _dev(off[:-1].astype(np.int64)), int(off[-1]))
    torch.cuda.synchronize()
    packed = packed.cpu().numpy()
    worst_np, worst_t = 0.0, 0.0
    for j, i in enumerate(idx):
        img = images[box_img[i]]
        pos, crop = obox.crop_box(img, rel[i])
        r = plans['rects'][j]
        assert (int(r[3] - r[1]), int(r[2] - r[0])) == crop.shape[:2]
        ref_np = oimg.prepare_crnn_input(crop, resize='numpy')
        ref_t = oimg.prepare_crnn_input(crop, resize='torch')
        got = packed[off[j]:off[j + 1]].reshape(64, out_w[j], 1)
        assert ref_np.shape == got.shape
        worst_np = max(worst_np, float(np.abs(got - ref_np).max()))
        worst_t = max(worst_t, float(np.abs(got - ref_t).max()))
    assert worst_np == 0.0, worst_np                     # same float32 operation sequence as the NumPy restatement
    assert worst_t <= 2e-6, worst_t                      # torch may fuse multiply-adds


# ----------------------------------------------------------------------------- CTC greedy

@pytest.mark.parametrize('B,T,V', [(7, 40, 97), (3, 63, 6719), (2, 1, 5), (1, 300, 97), (3, 2351, 97), (2, 4100, 11)])
def test_ctc_greedy_matches_oracle(ctx, B, T, V):
    rng = np.random.default_rng(B * 1000 + T)
    logits = rng.normal(0, 1, (B, T, V)).astype(np.float32)
    logits[..., 0] += 2.0                                  # plenty of blanks
    logits[0, T // 2:, :] = logits[0, T // 2, :]           # repeats
    if V > 10:
        logits[-1, 0, 5] = logits[-1, 0, 9] = 50.0         # exact tie: the first maximum wins
    lengths = rng.integers(0, T + 1, (B,)).astype(np.int32)
    lengths[0] = T
    for lens in (None, lengths):
        tok, sc = ctx.ctc_greedy(_dev(logits), _dev(lens) if lens is not None else None, blank=0)
        torch.cuda.synchronize()
        rt, rs = octc.ctc_greedy(logits, lens, 0)
        np.testing.assert_array_equal(tok.cpu().numpy(), rt)
        np.testing.assert_array_equal(sc.cpu().numpy(), rs)
