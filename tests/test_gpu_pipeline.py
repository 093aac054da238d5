"""End-to-end GPU pipeline against the reference-faithful CPU pipeline on small synthetic pages:
identical box counts, boxes at IoU >= 0.99 (bit-exact here, since both decode the same maps), and
identical strings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def test_pipeline_matches_cpu_reference_path():
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from oracle import boxes as obox, ctc as octc, nets, pipeline as opipe
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS

    east_params = nets.init_east_params(seed=3)
    crnn_params = nets.init_crnn_params(vocab_size=97, seed=4)
    images, maps, words = synth.make_pages(5, 2, 256, 384, 24)
    pipe = OcrPipeline(east_params, crnn_params)
    d_img = torch.from_numpy(images).cuda()
    d_maps = torch.from_numpy(maps).cuda()
    got = pipe(d_img, maps_override=d_maps)
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    assert tok.vocab_size == 97
    n_words = 0
    for b in range(2):
        ref = opipe.ocr_page(images[b], east_params, crnn_params, tok, maps_override=maps[b], run_network=False)
        assert len(got[b]['boxes']) == len(ref['boxes']) >= 10
        iou = obox.compute_iou(got[b]['boxes'], ref['boxes'], as_matrix=False)
        assert float(iou.min()) >= 0.99
        np.testing.assert_array_equal(got[b]['boxes'], ref['boxes'])
        assert list(np.nonzero(got[b]['valid'])[0]) == list(ref['index'])
        got_texts = [got[b]['texts'][i] for i in ref['index']]
        # identical strings; a string may differ only where the oracle's own logits have an argmax tie (two largest
        # logits of a step within 1e-4 of the logit scale) and taking the runner-up there reproduces it
        octc.assert_texts_match_or_tie(got_texts, ref['texts'], ref['logits'], tok)
        np.testing.assert_allclose([got[b]['scores'][i] for i in ref['index']], ref['scores'], rtol=1e-3, atol=1e-3)
        n_words += len(ref['texts'])
    assert n_words >= 20
    # the network output itself (not used by decode above) matches the oracle network
    maps_gpu, _, _ = pipe.detect(d_img[:1])
    x = np.stack([__import__('oracle.image', fromlist=['x']).prepare_east_input(images[0])])
    ref_maps = nets.east_forward(east_params, x).numpy()
    np.testing.assert_allclose(maps_gpu.cpu().numpy(), ref_maps, rtol=1e-4, atol=1e-5)


def test_blank_pages_and_streaming_batches():
    """A page without any detection next to a page with text, a batch of blank pages only, and the software-pipelined
    stream of batches: same results as one call per batch (base_ocr.py:182-184 skips what has no crop)."""
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_crnn_params, random_east_params

    pipe = OcrPipeline(random_east_params(3), random_crnn_params(97, seed=4))
    images, maps, _ = synth.make_pages(9, 3, 256, 384, 16)
    maps = maps.copy()
    maps[1, ..., 0] = 0.01                                   # page 1: nothing above the score threshold
    d_img, d_maps = torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda()
    got = pipe(d_img, maps_override=d_maps)
    assert len(got) == 3 and len(got[1]['boxes']) == 0 and got[1]['texts'] == [] and len(got[0]['boxes']) >= 5
    blank = pipe(d_img[1:2], maps_override=d_maps[1:2])
    assert len(blank) == 1 and len(blank[0]['boxes']) == 0
    # streaming: 4 batches through run_batches == the same batches one by one
    batches = [(d_img[i:i + 2], d_maps[i:i + 2]) for i in (0, 1, 0, 1)]
    streamed = list(pipe.run_batches(iter(batches)))
    assert len(streamed) == 4
    for (im, mp), out in zip(batches, streamed):
        single = pipe(im, maps_override=mp)
        for a, b in zip(out, single):
            np.testing.assert_array_equal(a['boxes'], b['boxes'])
            assert a['texts'] == b['texts']
            np.testing.assert_array_equal(a['scores'], b['scores'])


def test_lstm_timeout_in_the_stream_is_recognised_again_in_process():
    """A persistent BiLSTM launch that does not get its hand-overs (a co-tenant on the GPU; here the test hook) must not
    cost the batch: the pipeline runs the recogniser of every batch that was in flight again with one launch per time
    step, in the same process, counts it, and the stream's results equal the undisturbed ones."""
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_crnn_params, random_east_params

    pipe = OcrPipeline(random_east_params(3), random_crnn_params(97, seed=4))
    images, maps, _ = synth.make_pages(9, 2, 256, 384, 16)
    d_img, d_maps = torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda()
    want = pipe(d_img, maps_override=d_maps)
    assert pipe.lstm_fallbacks == 0

    def batches():
        for k in range(4):
            # the recogniser of batch 1 is enqueued while batch 2 is fed: the hook is on for exactly that call
            pipe.ctx.set_option('lstm_debug_drop_handover', 1 if k == 2 else 0)
            yield d_img, d_maps
    try:
        pipe.ctx.set_option('lstm_persistent', 1)
        streamed = list(pipe.run_batches(batches()))
    finally:
        pipe.ctx.set_option('lstm_debug_drop_handover', 0)
        pipe.ctx.set_option('lstm_persistent', 2)
    assert len(streamed) == 4 and 1 <= pipe.lstm_fallbacks <= 2
    for out in streamed:
        for a, b in zip(out, want):
            np.testing.assert_array_equal(a['boxes'], b['boxes'])
            assert a['texts'] == b['texts']
            # the per-step kernel and the persistent one compute the same sums
            np.testing.assert_allclose(a['scores'], b['scores'], rtol=1e-5, atol=1e-6)
