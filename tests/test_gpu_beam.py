"""GPU CTC prefix beam search (ocrhip_ctc_beam_search) against the host implementation of the same search
(utils/ctc_beam.py, itself checked exhaustively in test_ctc_beam_cpu.py): tokens bit-equal, scores within 1e-5, on
peaked and flat logits, ragged lengths, V = 97 and 6719; and the reference's default arguments (combine=True,
beam_search, 10 beams) through the batched pipeline against BaseOCR.infer's per-image path.  The reference's own
decoder (a TensorFlow kernel) is not available offline: parity of the search itself is unpinned beyond that."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def ctx():
    from ocr_amd.engine import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _logits(rng, B, T, V, kind):
    x = rng.normal(0, 1, (B, T, V)).astype(np.float32)
    if kind == 'peaked':                       # a trained recogniser: one class far ahead per step, many blanks, repeats
        lab = rng.integers(0, V, (B, T))
        lab[rng.uniform(size=(B, T)) < 0.5] = 0
        lab[:, 1::3] = lab[:, 0:-1:3][:, :lab[:, 1::3].shape[1]]
        np.put_along_axis(x, lab[..., None], 9.0, axis=2)
    elif kind == 'near':                       # two classes close together at every step: merging paths decides
        lab = rng.integers(0, V, (B, T, 2))
        np.put_along_axis(x, lab, 4.0 + rng.uniform(0, 0.3, (B, T, 2)).astype(np.float32), axis=2)
    return x


@pytest.mark.parametrize('V,T,kind,beams', [(97, 31, 'peaked', 10), (97, 31, 'flat', 10), (97, 63, 'near', 10), (97, 20, 'near', 3),
                                            (97, 17, 'flat', 16), (97, 12, 'peaked', 1), (6719, 15, 'peaked', 10), (6719, 9, 'near', 10),
                                            (5, 40, 'flat', 10), (2, 12, 'flat', 4)])
def test_beam_search_equals_the_host_search(ctx, V, T, kind, beams):
    from ocr_amd.utils.ctc_beam import ctc_beam_search
    rng = np.random.default_rng(V * 1000 + T)
    B = 12
    x = _logits(rng, B, T, V, kind)
    lengths = rng.integers(1, T + 1, B).astype(np.int32)
    lengths[0], lengths[1] = T, 0
    ref_t, ref_s = ctc_beam_search(x, lengths, num_beams=beams, blank=0)
    tok, sc = ctx.ctc_beam_search(torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda(), num_beams=beams, blank=0)
    np.testing.assert_array_equal(tok.cpu().numpy(), ref_t)
    np.testing.assert_allclose(sc.cpu().numpy(), ref_s, rtol=1e-5, atol=1e-6)
    # the flat layout the packed recogniser writes: sequences back to back, first rows given
    keep = lengths > 0
    flat = np.concatenate([x[b, :lengths[b]] for b in range(B) if keep[b]])
    L = lengths[keep]
    row0 = (np.cumsum(L) - L).astype(np.int32)
    tok2, sc2 = ctx.ctc_beam_search(torch.from_numpy(flat).cuda(), torch.from_numpy(L).cuda(), torch.from_numpy(row0).cuda(),
                                    t_stride=T, num_beams=beams, blank=0)
    np.testing.assert_array_equal(tok2.cpu().numpy(), ref_t[keep])
    np.testing.assert_allclose(sc2.cpu().numpy(), ref_s[keep], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        ctx.ctc_beam_search(torch.from_numpy(x).cuda(), num_beams=17)


def test_beam_search_ties_small_vocabularies_and_returning_prefixes(ctx):
    """Randomised sweep with what random float logits never show: exact ties (logits on a coarse grid: the two searches
    share one total order - value, then prefix, then label rank), vocabularies smaller than the beam, and prefixes that
    leave the kept set while an extension of theirs stays and come back later (the extension must still be recognised
    as their child: prefixes are compared by their labels, not by tree node)."""
    from ocr_amd.utils.ctc_beam import ctc_beam_search
    rng = np.random.default_rng(7)
    for case in range(120):
        V = int(rng.choice([2, 3, 5, 12, 64, 65, 97, 128, 129, 200]))
        T, K, B = int(rng.integers(1, 40)), int(rng.integers(1, 17)), 6
        x = rng.normal(0, 2, (B, T, V)).astype(np.float32)
        if case % 4 == 1:
            x = np.round(x * 2) / 2
        elif case % 4 == 2:
            x = np.round(x / 2).astype(np.float32)
        elif case % 4 == 3:
            lab = rng.integers(0, V, (B, T))
            lab[:, 1::2] = lab[:, 0:-1:2][:, :lab[:, 1::2].shape[1]]
            np.put_along_axis(x, lab[..., None], 8.0, axis=2)
        lengths = rng.integers(0, T + 1, B).astype(np.int32)
        lengths[0] = T
        ref_t, ref_s = ctc_beam_search(x, lengths, num_beams=K, blank=0)
        tok, sc = ctx.ctc_beam_search(torch.from_numpy(x).cuda(), torch.from_numpy(lengths).cuda(), num_beams=K, blank=0)
        np.testing.assert_array_equal(tok.cpu().numpy(), ref_t, err_msg='case %d V=%d T=%d K=%d' % (case, V, T, K))
        np.testing.assert_allclose(sc.cpu().numpy(), ref_s, rtol=1e-5, atol=1e-6, err_msg='case %d' % case)


def test_reference_default_arguments_through_the_batched_pipeline():
    """combine=True, method='beam_search', num_beams=10 (base_ocr.py:112-115,154-157): the batched pipeline returns,
    per page, the paragraphs BaseOCR.infer returns for that page alone - same grouping, same rows, same strings, scores
    equal to float32 summation order."""
    from ocr_amd import synth
    from ocr_amd.models.detection import EAST
    from ocr_amd.models.ocr import CRNN
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_crnn_params, random_east_params
    ep, cp = random_east_params(5), random_crnn_params(97, seed=6)
    images, maps, _ = synth.make_pages(33, 3, 256, 384, 24)
    pipe = OcrPipeline(ep, cp, combine=True, method='beam_search', num_beams=10)
    d_img, d_maps = torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda()
    got = pipe(d_img, maps_override=d_maps)
    streamed = list(pipe.run_batches(iter([(d_img, d_maps), (d_img[:1], d_maps[:1])])))
    east, crnn = EAST(ep, name='east_b'), CRNN(weights=cp, name='crnn_b')
    n_par = 0
    for b in range(3):
        boxes = east.decode_output(maps[b], inputs=np.zeros((1, 256, 384, 3), np.float32))
        ref = crnn.infer({'image': images[b], 'boxes': boxes})          # the reference's default arguments
        valid = np.nonzero(got[b]['valid'])[0]
        assert len(valid) == len(ref['ocr']) >= 2
        for p, r in zip(valid, ref['ocr']):
            np.testing.assert_array_equal(got[b]['boxes'][p], r['boxes'])
            np.testing.assert_array_equal(got[b]['rows'][p], r['rows'])
            assert got[b]['texts'][p] == r['text']
            np.testing.assert_allclose(got[b]['scores'][p], r['scores'], rtol=1e-4, atol=1e-5)
            n_par += 1
        for a, c in zip(streamed[0][b]['texts'], got[b]['texts']):
            assert a == c
    assert n_par >= 8 and streamed[1][0]['texts'] == got[0]['texts']
    # greedy through the same grouping: lines of a paragraph joined with ' \\n'
    g = OcrPipeline(None, None, ctx=pipe.ctx, combine=True)(d_img[:1], maps_override=d_maps[:1])[0]
    assert [t.count(' \n') + 1 for t, v in zip(g['texts'], g['valid']) if v] == [len(s) for s, v in zip(g['scores'], g['valid']) if v]
