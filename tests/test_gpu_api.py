"""The reference-shaped Python API (EAST, BaseOCR, ocr(), HIPRuntime, nms dispatcher) on the GPU,
checked against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def models():
    from ocr_amd.models.detection import EAST
    from ocr_amd.models.ocr import CRNN
    from ocr_amd.weights import random_crnn_params, random_east_params
    ep, cp = random_east_params(5), random_crnn_params(97, seed=6)
    return EAST(ep, name='east'), CRNN(weights=cp, name='crnn_en'), ep, cp


def test_east_infer_and_decode(models):
    from ocr_amd import synth
    from oracle import image as oimg, nets, pipeline as opipe
    east, _, ep, _ = models
    images, maps, _ = synth.make_pages(31, 1, 256, 384, 20)
    res = east.infer(images[0])
    assert set(res) >= {'filename', 'boxes', 'image', 'output'}
    assert res['output'].shape == (64, 96, 6) and res['boxes']['format'] in ('xyxy', 'poly')
    ref = nets.east_forward(ep, oimg.prepare_east_input(images[0])[None])[0].numpy()
    np.testing.assert_allclose(res['output'], ref, rtol=1e-4, atol=1e-5)
    # decode_output on given maps == reference-faithful CPU decode + LANMS
    got = east.decode_output(maps[0], inputs=np.zeros((1, 256, 384, 3), np.float32))
    _, boxes, _ = opipe.detect_page(images[0], None, maps_override=maps[0], run_network=False)
    assert got['format'] == 'xyxy'
    np.testing.assert_array_equal(got['boxes'], boxes)
    polys = east.decode_output(maps[0], inputs=np.zeros((1, 256, 384, 3), np.float32), nms_threshold=1.)
    assert polys['format'] == 'poly' and polys['boxes'].shape[1:] == (4, 2)


def test_east_input_not_multiple_of_32(models):
    from oracle import image as oimg, nets
    east, _, ep, _ = models
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (250, 390, 3), dtype=np.uint8)
    x = east.get_input(img)
    assert tuple(x.shape) == (256, 416, 3)
    ref_in = oimg.prepare_east_input(img)
    np.testing.assert_allclose(x.cpu().numpy(), ref_in, rtol=0, atol=2e-5)
    # bit-exact against the NumPy restatement of the same filter (float32 taps, width first, then height)
    np.testing.assert_array_equal(x.cpu().numpy(), oimg.prepare_east_input(img, resize='numpy'))
    # any target size, float input, batch of two: a down-scaling (wide filter support) and an up-scaling axis
    f = (rng.integers(0, 256, (2, 75, 130, 3)) / 255).astype(np.float32)
    y = east.ctx.image_resize_norm(torch.from_numpy(f).cuda(), 32, 160).cpu().numpy()
    for b in range(2):
        np.testing.assert_array_equal(y[b], oimg.normalize_mean_std(oimg.aa_resize_numpy(f[b], 32, 160)))
        np.testing.assert_allclose(y[b], oimg.normalize_mean_std(oimg.aa_resize_torch(f[b], 32, 160)), rtol=0, atol=2e-6)
    out = east.infer(img)['output']
    # the network on this page, at the tolerance of every other map comparison (1e-4 relative): the oracle network on
    # the SAME input the GPU network read (the NumPy restatement of the filter, bit-identical to the GPU's resize)
    ref = nets.east_forward(ep, oimg.prepare_east_input(img, resize='numpy')[None])[0].numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
    # against the oracle network on torch's resize of the page the maps differ by what the two INPUTS differ: torch
    # rounds the filter taps differently (inputs up to 2e-5 apart, above), and 20 layers carry that to ~2e-4 of a map
    # value - an input difference of the resize restatement, not a network one
    ref_t = nets.east_forward(ep, ref_in[None])[0].numpy()
    np.testing.assert_allclose(out, ref_t, rtol=2e-4, atol=2e-5)


def test_base_ocr_infer_and_ocr_function(models):
    from ocr_amd import synth
    from ocr_amd.models.ocr import ocr
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
    from oracle import ctc as octc, pipeline as opipe
    east, crnn, ep, cp = models
    images, maps, _ = synth.make_pages(32, 1, 256, 384, 20)
    boxes = east.decode_output(maps[0], inputs=np.zeros((1, 256, 384, 3), np.float32))
    res = crnn.infer({'image': images[0], 'boxes': boxes}, combine=False, method='greedy')
    assert set(res) == {'filename', 'boxes', 'ocr', 'image'}
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    index, texts, scores, logits = opipe.recognise_rows(images[0], boxes['boxes'], cp, tok, return_logits=True)
    assert len(res['ocr']) == len(index) >= 10
    octc.assert_texts_match_or_tie([r['text'] for r in res['ocr']], texts, logits, tok)
    for r, i, s in zip(res['ocr'], index, scores):
        assert r['format'] == 'xyxy' and r['rows'].shape == (1, 4)
        np.testing.assert_array_equal(r['boxes'], boxes['boxes'][i])
        assert abs(float(r['scores'][0]) - float(s)) < 1e-3
    # get_input for one row == the oracle's prepared crop
    from oracle import boxes as obox, image as oimg
    inp = crnn.get_input({'filename': images[0], 'boxes': {'boxes': boxes['boxes'][0:1], 'format': 'xyxy'}})
    ref = oimg.prepare_crnn_input(obox.crop_box(images[0], boxes['boxes'][0])[1], resize='numpy')
    np.testing.assert_array_equal(inp.cpu().numpy(), ref)
    # the public function: detector by name, random-weight detection -> just the structure
    out = ocr(images[0], model='crnn_en', detector='east')
    assert isinstance(out, list) and len(out) == 1 and 'ocr' in out[0]
    # reference defaults: words grouped into lines / paragraphs, beam search; on these well separated
    # logits-free checks only the structure and the grouping are compared
    from ocr_amd.utils.combination import combine_boxes
    full = crnn.infer({'image': images[0], 'boxes': boxes})
    cb, _, rows = combine_boxes(boxes['boxes'])
    assert len(full['ocr']) == len(cb)
    for r, bx, rw in zip(full['ocr'], cb, rows):
        np.testing.assert_array_equal(r['boxes'], bx)
        assert r['rows'].shape == np.asarray(rw).shape and r['text'].count(' \n') == len(rw) - 1
    # beam search: same rows as greedy, scores are log-probabilities / length (random weights give nearly
    # flat distributions, so the strings themselves are not comparable with greedy here; the search is
    # checked exhaustively in tests/test_ctc_beam_cpu.py)
    g = crnn.infer({'image': images[0], 'boxes': boxes}, combine=False, method='greedy')
    bsr = crnn.infer({'image': images[0], 'boxes': boxes}, combine=False, method='beam_search')
    assert len(g['ocr']) == len(bsr['ocr'])
    for x, y in zip(g['ocr'], bsr['ocr']):
        np.testing.assert_array_equal(x['boxes'], y['boxes'])
    assert all(-10 < float(r['scores'][0]) <= 0 for r in bsr['ocr'])
    # dezoom_factor: every row's pixel box scaled about its centre before the crop (base_ocr.py:178-181 ->
    # converter.py:202-232) == recognising the boxes the oracle's dezoom gives
    from ocr_amd.utils.image_sizes import dezoom_absolute, relative_to_absolute
    dz = crnn.infer({'image': images[0], 'boxes': boxes}, combine=False, method='greedy', dezoom_factor=1.5)
    H, W = images[0].shape[:2]
    grown = dezoom_absolute(relative_to_absolute(boxes['boxes'], H, W), 1.5, H, W)
    ref_texts, ref_logits = [], []
    for x0, y0, x1, y1 in grown:
        inp = oimg.prepare_crnn_input(images[0][y0:y1, x0:x1], resize='numpy')
        lg = __import__('oracle.nets', fromlist=['x']).crnn_forward(cp, inp[None])[0].numpy()
        ref_logits.append(lg)
        ref_texts.append(tok.ctc_decode(lg[None])[0][0])
    assert len(dz['ocr']) == len(ref_texts) and [r['text'] for r in dz['ocr']] != [r['text'] for r in g['ocr']]
    octc.assert_texts_match_or_tie([r['text'] for r in dz['ocr']], ref_texts, ref_logits, tok)
    for r, bx in zip(dz['ocr'], boxes['boxes']):
        np.testing.assert_array_equal(r['boxes'], bx)                      # the reported boxes are the detected ones


def test_runtime_object_and_nms_dispatch(models):
    from ocr_amd.runtimes import HIPRuntime, build_runtime
    from ocr_amd.utils.boxes import nms
    from oracle import nets
    east, crnn, ep, cp = models
    rt = build_runtime('hip', 'random:crnn:97:6')
    assert isinstance(rt, HIPRuntime) and rt.path == 'random:crnn:97:6' and rt.layers == []
    assert build_runtime('hip', 'random:crnn:97:6').engine is rt.engine          # class-level engine cache
    x = np.random.default_rng(1).uniform(-1, 1, (1, 64, 128, 1)).astype(np.float32)
    out = rt(x, method='beam_search', num_beams=10, num_sentences=1, length_power=.25)   # extra kwargs ignored
    assert isinstance(out, np.ndarray) and out[0].shape == (31, 97) and len(out) == 1
    ref = nets.crnn_forward(cp, x).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()))
    chain = np.array([[0, 0, 0.2, 0.2], [0.1, 0.1, 0.3, 0.3], [0.2, 0.2, 0.4, 0.4], [0.3, 0.3, 0.5, 0.5]], 'float32')
    boxes, scores, valid = nms(chain, nms_threshold=0.1, merge_threshold=0.1, source='xyxy', method='lanms')
    np.testing.assert_allclose(boxes[valid], [[0, 0, 0.3, 0.3], [0.2, 0.2, 0.5, 0.5]], atol=1e-6)
    assert boxes.shape == (4, 4) and valid.dtype == bool


@pytest.mark.parametrize('method', ['tensorflow', 'nms', 'standard', 'fast', 'padded', 'padded_nms'])
def test_standard_nms_methods(method):
    """The other methods of the reference's registry (non_max_suppression.py:80-189): its known answer
    (tests/test_utils_boxes.py:559-584: 4 chained boxes, threshold 0.1 -> keep [0, 2]) and random sets against the
    oracle's restatement of the matrix iteration."""
    from ocr_amd.utils.boxes import nms
    from oracle import nms as onms
    chain = np.array([[0, 0, 0.2, 0.2], [0.1, 0.1, 0.3, 0.3], [0.2, 0.2, 0.4, 0.4], [0.3, 0.3, 0.5, 0.5]], 'float32')
    boxes, _, valid = nms(chain, nms_threshold=0.1, source='xyxy', method=method)
    np.testing.assert_array_equal(boxes[valid], chain[[0, 2]])
    with pytest.raises(ValueError):
        nms(chain, nms_threshold=0.1, source='xyxy', method='no_such_method')
    rng = np.random.default_rng(17)
    for trial in range(6):
        n = int(rng.integers(2, 400))
        xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
        b = np.concatenate([xy, xy + rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)], axis=1)
        sc = (rng.permutation(n).astype(np.float32) + 1) / n
        thr = float(rng.choice([0.1, 0.25, 0.5]))
        max_out = None if trial % 2 else 7
        rb, rs, rv, order = onms.fast_nms(b, sc, max_output_size=max_out, nms_threshold=thr)
        got_b, got_s, got_v = nms(b, sc, max_out, thr, method=method, source='xyxy')
        if method == 'tensorflow':                        # input order, boxes untouched
            np.testing.assert_array_equal(got_b, b)
            np.testing.assert_array_equal(np.nonzero(got_v)[0], np.sort(order[rv]))
        else:                                             # score order, dropped boxes zeroed
            np.testing.assert_array_equal(got_b, rb)
            np.testing.assert_array_equal(got_s, rs)
            np.testing.assert_array_equal(got_v, rv)


def test_checkpoint_files_round_trip_through_the_runtime(tmp_path):
    """`HIPRuntime(<file>.pth)`: the torch checkpoints of the original projects (DataParallel `module.` prefixes,
    `num_batches_tracked` counters) loaded from disk give bit for bit the outputs of the same parameters handed over
    in memory, for the detector and for the recogniser (VGG and ResNet extractors)."""
    from ocr_amd.runtimes import HIPRuntime
    from ocr_amd.weights import random_crnn_params, random_east_params
    rng = np.random.default_rng(5)
    cases = [('east', random_east_params(8), rng.integers(0, 256, (1, 96, 128, 3), dtype=np.uint8)),
             ('crnn_vgg', random_crnn_params(97, seed=9), rng.uniform(-1, 1, (2, 64, 128, 1)).astype(np.float32)),
             ('crnn_resnet', random_crnn_params(211, hidden=512, seed=10, backbone='resnet'),
              rng.uniform(-1, 1, (2, 64, 96, 1)).astype(np.float32))]
    for name, params, x in cases:
        saved = {'module.' + k: v for k, v in params.items()}
        bn = [k for k in saved if k.endswith('running_mean')]
        for k in bn:
            saved[k.replace('running_mean', 'num_batches_tracked')] = torch.tensor(77)
        path = str(tmp_path / (name + '.pth'))
        torch.save(saved, path)
        from_file = HIPRuntime(path)
        in_memory = HIPRuntime('state_dict:' + name, state_dict=params, reload=True)
        assert from_file.engine.kind == ('east' if name == 'east' else 'crnn')
        a, b = from_file(x), in_memory(x)
        assert isinstance(a, np.ndarray) and np.isfinite(a).all() and float(np.abs(a).max()) > 0
        np.testing.assert_array_equal(a, b)
        assert HIPRuntime(path).engine is from_file.engine               # cached per path
        from_file.engine.ctx.close()
        in_memory.engine.ctx.close()
        HIPRuntime._engines.pop(path, None)
        HIPRuntime._engines.pop('state_dict:' + name, None)
