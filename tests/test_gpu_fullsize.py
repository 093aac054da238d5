"""Properties at the benchmark's full size (1024x1024 pages, ~220 words each), where the CPU oracle would take
minutes: the three float32 convolution algorithms agree to the parity tolerance on the whole detector, a page's
result does not depend on the batch it travels in, LANMS is idempotent on its own output, decode order is the
documented row-major order, and CTC output never contains blanks or adjacent repeats of what it merged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def setup():
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_crnn_params, random_east_params
    pipe = OcrPipeline(random_east_params(1), random_crnn_params(97, seed=2))
    images, maps, _ = synth.make_pages(1234 + 4, 2, 1024, 1024, 220)
    return pipe, torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda()


def test_detector_algorithms_agree_at_full_size(setup):
    """Winograd F(4x4,3x3) (default), F(2x2,3x3) and the direct float32 fma chain over 1024x1024 pages: within
    rtol 1e-4 / atol 1e-5 of each other on the sigmoid maps (the tolerance of the oracle tests), and a page computed
    alone equals the same page computed in a batch bit for bit."""
    pipe, d_img, _ = setup
    ctx = pipe.ctx
    out = {}
    try:
        for opt in (0, 1, 2):
            ctx.set_option('conv3x3_winograd', opt)
            out[opt] = ctx.east_forward(d_img).cpu().numpy()
        alone = ctx.east_forward(d_img[1:2]).cpu().numpy()
    finally:
        ctx.set_option('conv3x3_winograd', 2)
    assert out[2].shape == (2, 256, 256, 6) and np.isfinite(out[2]).all()
    for opt in (1, 2):
        np.testing.assert_allclose(out[opt], out[0], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(alone[0], out[2][1])
    assert not np.array_equal(out[2], out[1])


def test_page_results_do_not_depend_on_the_batch(setup):
    """Boxes and strings of a page are the same whether it is processed alone or in a batch.  The recogniser's scores
    are equal bit for bit when every crop is convolved on its own (`wino4_side_by_side` = 0); under the default, crops
    of a launch share Winograd patches with their neighbours and the scores agree to rounding (a few ulp)."""
    pipe, d_img, d_maps = setup
    for side_by_side in (1, 0):
        pipe.ctx.set_option('wino4_side_by_side', side_by_side)
        try:
            both = pipe(d_img, maps_override=d_maps)
            for b in range(2):
                single = pipe(d_img[b:b + 1], maps_override=d_maps[b:b + 1])[0]
                np.testing.assert_array_equal(single['boxes'], both[b]['boxes'])
                assert single['texts'] == both[b]['texts']
                if side_by_side:
                    np.testing.assert_allclose(single['scores'], both[b]['scores'], rtol=2e-6, atol=0)
                else:
                    np.testing.assert_array_equal(single['scores'], both[b]['scores'])
                assert 150 <= len(both[b]['boxes']) <= 260                  # ~220 rendered words per page
        finally:
            pipe.ctx.set_option('wino4_side_by_side', 1)


def test_lanms_is_idempotent_on_its_output(setup):
    """Boxes that survived merging and suppression do not overlap each other above the thresholds: a second pass keeps
    all of them unchanged."""
    pipe, d_img, d_maps = setup
    res = pipe(d_img, maps_override=d_maps)
    n = [len(r['boxes']) for r in res]
    cap = max(n)
    xyxy = np.zeros((2, cap, 4), np.float32)
    for b in range(2):
        xyxy[b, :n[b]] = res[b]['boxes']
    scores = np.tile(np.linspace(0.99, 0.4, cap, dtype=np.float32), (2, 1))
    got = pipe.ctx.lanms(torch.from_numpy(xyxy).cuda(), torch.from_numpy(scores).cuda(),
                         torch.tensor(n, dtype=torch.int32).cuda(), merge_threshold=0.3, nms_threshold=0.2)
    counts = got['counts'].cpu().numpy()
    assert list(counts) == n
    for b in range(2):
        np.testing.assert_array_equal(got['boxes'][b, :n[b]].cpu().numpy(), res[b]['boxes'])


def test_decode_order_and_ctc_invariants(setup):
    pipe, d_img, d_maps = setup
    ctx = pipe.ctx
    dec = ctx.east_decode(d_maps, (1024, 1024), threshold=0.35)
    counts = dec['counts'].cpu().numpy()
    assert (counts > 5000).all()                                       # thousands of candidates per page
    sc = d_maps[..., 0].cpu().numpy()
    for b in range(2):
        want = sc[b][sc[b] > 0.35]                                     # row-major order of the score map
        got = dec['scores'][b, :counts[b]].cpu().numpy()
        assert len(got) <= len(want)                                   # the validity filter only removes
        it = iter(want.tolist())
        assert all(any(g == w for w in it) for g in got.tolist())      # got is a subsequence of want
    rng = np.random.default_rng(0)
    logits = torch.from_numpy(rng.normal(size=(64, 127, 97)).astype(np.float32)).cuda()
    lengths = torch.from_numpy(rng.integers(1, 128, 64).astype(np.int32)).cuda()
    toks = ctx.ctc_greedy(logits, lengths)[0].cpu().numpy()
    am = logits.argmax(-1).cpu().numpy()
    for i in range(64):
        t = toks[i][toks[i] >= 0]
        assert (t != 0).all()
        L = int(lengths[i])
        ref = [int(v) for k, v in enumerate(am[i, :L]) if v != 0 and (k == 0 or v != am[i, k - 1])]
        assert t.tolist() == ref


def test_decode_and_lanms_on_the_networks_own_maps_at_full_size(setup):
    """The detector's OWN output at 1024x1024 (random weights: scores hover around 0.5 and geometry / angles are
    arbitrary, so rotated boxes, boxes leaving the page and dense overlaps all occur) through decode + LANMS + NMS,
    against the oracle on the same maps.  The threshold is set so that 20 000 pixels pass it (about a tenth of them survive the validity filter)."""
    from oracle import boxes as obox, east_decode, lanms_c
    pipe, d_img, _ = setup
    ctx = pipe.ctx
    maps = ctx.east_forward(d_img)
    h_maps = maps.cpu().numpy()
    for b in range(2):
        thr = float(np.quantile(h_maps[b, ..., 0], 1.0 - 20000.0 / h_maps[b, ..., 0].size))
        dec = ctx.east_decode(maps[b:b + 1], (1024, 1024), threshold=thr, with_pix=True)
        out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], merge_threshold=0.3, nms_threshold=0.2)
        torch.cuda.synchronize()
        n = int(dec['counts'][0])
        ref = east_decode.decode_map(h_maps[b], (1024, 1024), threshold=thr, trig='rounded64', return_index=True)
        assert n == len(ref['boxes']) and 500 <= n <= 20000
        np.testing.assert_array_equal(dec['pix'][0, :n].cpu().numpy(), ref['index'])
        xy = dec['xyxy'][0, :n].cpu().numpy()
        np.testing.assert_array_equal(xy, obox.poly_to_xyxy(ref['boxes']))
        sc = dec['scores'][0, :n].cpu().numpy()
        rb, rs, rmask, _, r_iter = lanms_c.lanms(xy, sc, None, 0.2, merge_threshold=0.3, merge_method='union')
        m = int(out['counts'][0])
        assert m == int(rmask.sum()) and int(out['n_iter'][0]) == r_iter
        np.testing.assert_array_equal(out['boxes'][0, :m].cpu().numpy(), rb[rmask])
        np.testing.assert_array_equal(out['scores'][0, :m].cpu().numpy(), rs[rmask])
