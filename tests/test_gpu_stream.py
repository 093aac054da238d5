"""OCR stream on the GPU: the Dice kernel against the reference's values, and `BaseOCR.stream` over a frame
sequence with known repeats (teacher detector, so the boxes are meaningful with random weights)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def models():
    from ocr_amd.models.detection import EAST
    from ocr_amd.models.ocr import CRNN
    from ocr_amd.weights import random_crnn_params, random_east_params
    return EAST(random_east_params(5), name='east_stream'), CRNN(weights=random_crnn_params(97, seed=6), name='crnn_stream')


def test_dice_coefficient_matches_the_reference(models):
    east, _ = models
    gold = np.load(os.path.join(GOLD, 'stream_golden.npz'))
    maps = torch.from_numpy(gold['dice_maps']).cuda()
    for (a, b), want in zip(gold['dice_pairs'], gold['dice_values']):
        got = float(east.ctx.dice_coeff(maps[a], maps[b]).item())
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (a, b, got, want)
    assert float(east.ctx.dice_coeff(maps[5], maps[5]).item()) == 0.0            # 0 / 0 -> 0
    # channel 0 of two (h,w,6) detector maps, walked in place, against float64 NumPy
    rng = np.random.default_rng(1)
    m = rng.random((2, 256, 256, 6), dtype=np.float32)
    d = torch.from_numpy(m).cuda()
    x, y = m[0, :, :, 0].astype(np.float64), m[1, :, :, 0].astype(np.float64)
    want = 2 * (x * y).sum() / (x.sum() + y.sum())
    assert abs(float(east.ctx.dice_coeff(d[0, :, :, 0], d[1, :, :, 0]).item()) - want) < 1e-6
    # a view without a uniform stride takes the copy path
    got = float(east.ctx.dice_coeff(d[0, :100, :50, 0], d[1, :100, :50, 0]).item())
    xs, ys = x[:100, :50], y[:100, :50]
    assert abs(got - 2 * (xs * ys).sum() / (xs.sum() + ys.sum())) < 1e-6
    with pytest.raises(ValueError):
        east.ctx.dice_coeff(d[0, :, :, 0], d[1, :100, :, 0])


class TeacherDetector:
    """Answers like `EAST.infer` with the page's teacher maps instead of a forward pass."""

    def __init__(self, east, pages):
        self.east, self.pages = east, pages

    def infer(self, frame, output_on_device=False, **_):
        image, maps = self.pages[frame]
        boxes = self.east.decode_output(maps, inputs=np.zeros((1,) + image.shape, np.float32))
        # a trained detector's score map is close to 0 / 1; the teacher maps (0.9 inside words, noise outside) are
        # rounded for the frame comparison, which makes the Dice coefficient of a frame with itself 1
        maps = maps.copy()
        maps[..., 0] = maps[..., 0] > 0.5
        out = torch.from_numpy(maps).cuda() if output_on_device else maps
        return {'filename': None, 'boxes': boxes, 'image': image, 'output': out}


def test_stream_emits_every_text_once(models, request):
    from ocr_amd import synth
    from ocr_amd.models.ocr.streaming import keep_text_result
    from ocr_amd.utils.filters import RegionFilter, SizeFilter
    east, crnn = models
    images, maps, _ = synth.make_pages(77, 2, 256, 384, 24)
    blank = (np.full_like(images[0], 255), np.zeros_like(maps[0]))
    pages = {'A': (images[0], maps[0]), 'B': (images[1], maps[1]), '-': blank}
    det = TeacherDetector(east, pages)
    region = [0.0, 0.0, 1.0, 1.0]
    text_kw = dict(skip_single_word=False, skip_non_alpha=False, threshold=0., k=0, method='greedy')
    heard = []
    frames = ['A', 'A', 'A', 'A', '-', 'B', 'B', 'B', 'A', 'A']
    # a recogniser with random weights reads the same few symbols everywhere: tag every row's text with its place so
    # that texts differ between places and repeat between frames
    real_rows = crnn.recognise_rows

    def tagged_rows(image, rows, **kw):
        index, texts, scores = real_rows(image, rows, **kw)
        rows = np.asarray(rows).reshape(-1, 4)
        return index, ['%s x%03d y%03d' % (t, round(999 * rows[i][0]), round(999 * rows[i][1]))
                       for i, t in zip(index, texts)], scores

    crnn.recognise_rows = tagged_rows
    request.addfinalizer(lambda: crnn.__dict__.pop('recognise_rows', None))
    run = crnn.stream(frames, detector=det, region=region, callback=heard.append, **text_kw)
    by_frame = {}
    for f, r in run.emitted:
        by_frame.setdefault(f, []).append(r['text'])
    # what one frame holds, recognised outside the stream with the same static filters
    expect = {}
    for name in 'AB':
        res = crnn.infer({'image': pages[name][0], 'boxes': det.infer(name)['boxes']}, combine=True, method='greedy',
                         box_filters=[SizeFilter(min_h=0.025, min_w=0.1), RegionFilter(region=region, mode='center')])
        texts = []
        for r in res['ocr']:
            r = dict(r)
            if keep_text_result(r, set(), **{k: v for k, v in text_kw.items() if k != 'method'}):
                texts.append(r['text'])
        assert len(texts) >= 3
        expect[name] = texts
    # frame 0: first sighting; 1: second sighting -> emitted; 2: tracks start over; 3: seen twice again but already
    # emitted at these places; 4: no box (new frame, state reset); 5-7: page B likewise; 8-9: page A again is a new
    # frame, but its texts were emitted before
    assert sorted(by_frame) == [1, 6]
    assert len(set(expect['A'])) >= 3 and len(set(expect['B']) - set(expect['A'])) >= 3
    assert by_frame[1] == expect['A'] and by_frame[6] == [t for t in expect['B'] if t not in set(expect['A'])]
    assert heard == by_frame[1] + by_frame[6]
    assert run.state.n_duplicates == 1 and run.state.emitted_texts == set(heard)
    with pytest.raises(NotImplementedError):
        crnn.stream(0)


def test_long_runs_of_duplicates_are_thinned(models):
    """After `duplicate_tolerance` duplicates in a row only 3 of every 5 unproductive frames reach the recogniser."""
    from ocr_amd import synth
    east, crnn = models
    images, maps, _ = synth.make_pages(78, 1, 256, 384, 12)
    det = TeacherDetector(east, {'A': (images[0], maps[0])})
    calls = []
    real_infer = crnn.infer
    crnn.infer = lambda *a, **k: (calls.append(1), real_infer(*a, **k))[1]
    try:
        run = crnn.stream(['A'] * 16, detector=det, region=None, n_repeat=0, duplicate_tolerance=5, method='greedy',
                          skip_single_word=False, skip_non_alpha=False)
    finally:
        del crnn.infer
    # frames 0..5 reach the recogniser (n_duplicates <= 5 up to frame 5); from frame 6 on the frame is dropped while
    # n_skipped % 5 < 2.  n_skipped grows on every unproductive frame (everything was emitted at frame 0).
    skipped, n_skip, n_dup = [], 0, 0
    for f in range(16):
        if f == 0:
            n_skip = 0
            continue
        n_dup += 1
        if n_dup > 5 and n_skip % 5 < 2:
            n_skip += 1
            skipped.append(f)
            continue
        n_skip += 1
    assert len(calls) == 16 - len(skipped) and len(skipped) >= 3
    assert sorted({f for f, _ in run.emitted}) == [0]
