"""Every BASELINE configuration at its full size on the GPU.

configs[1] (32 pages 512x512, detection) and configs[2] (4096 word crops, recognition, bfloat16 variant) are checked
against the CPU oracle run live (seconds); configs[3] (1024x1024 pages) and configs[4] (2048x2048 pages, ~500 words,
6719-class zh_sim_g2 head) against outputs the oracle produced once in the build container
(tests/golden/make_config_golden.py -> config4_page.npz, config5_pages.npz; the inputs are re-created from seeds).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from conftest import GOLDEN_DIR  # noqa: E402
from oracle import boxes as obox, ctc as octc, east_decode, image as oimg, lanms_c, nets  # noqa: E402


def _close(got, ref, rtol=1e-4, atol=1e-5):
    err = np.abs(np.asarray(got) - np.asarray(ref))
    bad = err > atol + rtol * np.abs(ref)
    assert not bad.any(), 'max err %.3e, %d / %d out of tolerance' % (err.max(), int(bad.sum()), bad.size)


# ------------------------------------------------------------------------------------------- configs[3], configs[4]

@pytest.mark.parametrize('name', ['config4_page', 'config5_pages'])
def test_full_pipeline_matches_the_oracle_fixture(name):
    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.utils.text import CharTokenizer, get_characters
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    S, pages, V = int(g['size']), int(g['pages']), int(g['vocab_size'])
    chars = get_characters(str(g['charset']))
    tok = octc.CharTokenizer(chars)
    assert tok.vocab_size == V
    east, crnn = nets.init_east_params(seed=int(g['east_seed'])), nets.init_crnn_params(vocab_size=V, seed=int(g['crnn_seed']))
    images, maps, _ = synth.make_pages(int(g['page_seed']), pages, S, S, int(g['words']))
    pipe = OcrPipeline(east, crnn, tokenizer=CharTokenizer(chars))
    try:
        d_img, d_maps = torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda()
        # detector: the network's own maps of page 0 on the sampled map pixels, 1e-4 relative
        net_maps = pipe.ctx.east_forward(d_img[:1]).cpu().numpy().reshape(-1, 6)
        _close(net_maps[g['maps_pixels']], g['maps_values'])
        np.testing.assert_allclose(net_maps.astype(np.float64).mean(axis=0), g['maps_channel_means'], rtol=1e-4, atol=1e-6)
        # decode + LANMS + crop + recogniser + CTC on the teacher maps
        got = pipe(d_img, maps_override=d_maps)
        n_strings = 0
        for b in range(pages):
            p = 'page%d_' % b
            np.testing.assert_array_equal(got[b]['boxes'], g[p + 'boxes'])
            # the reference's own trigonometry (NumPy float32 sin / cos): identical count, every box at IoU >= 0.99
            ref32 = g[p + 'boxes_numpy32']
            assert len(ref32) == len(got[b]['boxes']) >= 0.9 * int(g['words'])
            iou = obox.compute_iou(got[b]['boxes'], ref32, as_matrix=True)
            assert float(iou.max(axis=1).min()) >= 0.99 and float(iou.max(axis=0).min()) >= 0.99
            assert list(np.nonzero(got[b]['valid'])[0]) == list(g[p + 'index'])
            texts = [got[b]['texts'][i] for i in g[p + 'index']]
            row0 = np.concatenate([[0], np.cumsum(g[p + 'steps'])])
            top2 = [(g[p + 'top2_class'][row0[k]:row0[k + 1]], g[p + 'top2_logit'][row0[k]:row0[k + 1]], g[p + 'logit_absmax'][k])
                    for k in range(len(texts))]
            flips = octc.assert_texts_match_or_tie(texts, [str(t) for t in g[p + 'texts']], top2, tok)
            assert flips <= max(1, len(texts) // 50)             # ties are rare events, not a way out
            np.testing.assert_allclose([got[b]['scores'][i] for i in g[p + 'index']], g[p + 'scores'], rtol=1e-3, atol=1e-3)
            n_strings += len(texts)
        assert n_strings >= 0.9 * pages * int(g['words'])
    finally:
        pipe.ctx.close()


# ------------------------------------------------------------------------------------------------------- configs[1]

def test_config2_detection_batch_of_32_pages_512():
    """EAST-VGG16 detection only, batch 32 of 512x512 pages, float32: the network maps of the first and the last page
    of the batch against the oracle network, decode + LANMS + NMS of all 32 pages bit-checked against the oracle."""
    from ocr_amd import synth
    from ocr_amd.engine import HipContext
    east = nets.init_east_params(seed=1)
    images, maps, _ = synth.make_pages(1234 + 2, 32, 512, 512, 30)
    ctx = HipContext(0)
    try:
        ctx.east_load_weights(east)
        net_maps = ctx.east_forward(torch.from_numpy(images).cuda())
        for b in (0, 31):
            ref = nets.east_forward(east, oimg.prepare_east_input(images[b])[None])[0].numpy()
            _close(net_maps[b].cpu().numpy(), ref)
        dec = ctx.east_decode(torch.from_numpy(maps).cuda(), (512, 512), 0.35, with_pix=True)
        out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], merge_threshold=0.3, nms_threshold=0.2)
        torch.cuda.synchronize()
        cnt, kept_n = dec['counts'].cpu().numpy(), out['counts'].cpu().numpy()
        xy, sc, pix = dec['xyxy'].cpu().numpy(), dec['scores'].cpu().numpy(), dec['pix'].cpu().numpy()
        kept, kept_sc, n_iter = out['boxes'].cpu().numpy(), out['scores'].cpu().numpy(), out['n_iter'].cpu().numpy()
        for b in range(32):
            ref = east_decode.decode_map(maps[b], (512, 512), trig='rounded64', return_index=True)
            n = int(cnt[b])
            assert n == len(ref['boxes']) > 500
            np.testing.assert_array_equal(pix[b, :n], ref['index'])
            np.testing.assert_array_equal(xy[b, :n], obox.poly_to_xyxy(ref['boxes']))
            np.testing.assert_array_equal(sc[b, :n], ref['scores'])
            rb, rs, rmask, _, r_iter = lanms_c.lanms(xy[b, :n], sc[b, :n], None, 0.2, merge_threshold=0.3, merge_method='union')
            m = int(kept_n[b])
            assert m == int(rmask.sum()) and 20 <= m <= 40 and int(n_iter[b]) == r_iter
            np.testing.assert_array_equal(kept[b, :m], rb[rmask])
            np.testing.assert_array_equal(kept_sc[b, :m], rs[rmask])
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------------- configs[2]

def _shelf_pack(crops, size=2048):
    """Word crops laid out on square pages: -> pages (B,size,size,3) uint8, per page the relative xyxy boxes."""
    pages, boxes = [np.full((size, size, 3), 230, np.uint8)], [[]]
    x = y = shelf = 0
    for c in crops:
        h, w = c.shape[:2]
        if x + w > size:
            x, y, shelf = 0, y + shelf, 0
        if y + h > size:
            pages.append(np.full((size, size, 3), 230, np.uint8))
            boxes.append([])
            x = y = shelf = 0
        pages[-1][y:y + h, x:x + w] = c
        boxes[-1].append([x / size, y / size, (x + w) / size, (y + h) / size])
        x, shelf = x + w, max(shelf, h)
    return np.stack(pages), [np.array(b, dtype=np.float32) for b in boxes]


def test_config3_recognition_of_4096_word_crops_f32_and_bf16():
    """CRNN English recognition only, 4096 synthetic word crops (source height 12..96, aspect 1..8, so padded widths
    64..512), CTC greedy: float32 logits within 1e-4 and bfloat16-operand logits within 1e-2 of the float32 oracle on
    every 8th crop; greedy tokens and scores of ALL crops bit-equal to the oracle's decoding of the same GPU logits,
    in both precisions; float32 strings equal to the oracle's strings apart from proven ties."""
    from ocr_amd import synth
    from ocr_amd._lib import BF16, F32
    from ocr_amd.engine import HipContext
    from ocr_amd.utils import image_sizes as S
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
    crnn = nets.init_crnn_params(vocab_size=97, seed=2)
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    crops = synth.make_crops(1234 + 3, 4096)
    pages, boxes = _shelf_pack(crops)
    H = W = pages.shape[1]
    plans = [S.plan_crops(b, H, W) for b in boxes]
    assert sum(len(p['index']) for p in plans) == 4096
    rects = np.concatenate([p['rects'] for p in plans])
    inter = np.concatenate([p['inter'] for p in plans])
    out_w = np.concatenate([p['out_w'] for p in plans])
    box_img = np.concatenate([np.full(len(p['index']), i, np.int32) for i, p in enumerate(plans)])
    order = np.argsort(out_w, kind='stable')
    rects, inter, out_w, box_img = rects[order], inter[order], out_w[order], box_img[order]
    assert out_w.min() == 64 and out_w.max() >= 512
    off = np.concatenate([[0], np.cumsum(64 * out_w.astype(np.int64))])
    ctx = HipContext(0)
    try:
        ctx.crnn_load_weights(crnn)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        packed = ctx.crop_resize_pack(dev(pages), dev(rects), dev(box_img), dev(inter), dev(out_w), dev(off[:-1]), int(off[-1]))
        h_packed = packed.cpu().numpy()
        T = ctx.seq_len(out_w)
        row0 = np.concatenate([[0], np.cumsum(T)])
        # the oracle network on every 8th crop, batched per width (its input is the packed plane the GPU network reads;
        # that plane itself is checked against the oracle's image preparation for a few crops)
        sample = np.arange(0, 4096, 8)
        ref = {}
        for w in np.unique(out_w[sample]):
            ks = sample[out_w[sample] == w]
            x = np.stack([h_packed[off[k]:off[k + 1]].reshape(64, w, 1) for k in ks])
            for k, lg in zip(ks, nets.crnn_forward(crnn, x).numpy()):
                ref[int(k)] = lg
        for j in sample[::64]:
            crop = pages[int(box_img[j])][rects[j, 1]:rects[j, 3], rects[j, 0]:rects[j, 2]]
            np.testing.assert_array_equal(h_packed[off[j]:off[j + 1]].reshape(64, out_w[j], 1),
                                          oimg.prepare_crnn_input(crop, resize='numpy'))
        for dtype, tol in ((F32, 1e-4), (BF16, 1e-2)):
            res = ctx.crnn_forward_packed(packed, out_w, want_logits=True, compute_dtype=dtype)
            torch.cuda.synchronize()
            ctx.check_errors()
            logits, tokens, scores = res['logits'].cpu().numpy(), res['tokens'].cpu().numpy(), res['scores'].cpu().numpy()
            worst = 0.0
            for k, lg in ref.items():
                got = logits[row0[k]:row0[k + 1]]
                worst = max(worst, float(np.abs(got - lg).max() / np.abs(lg).max()))
            assert worst <= tol, worst
            # greedy decoding of all 4096 sequences == the oracle's decoding of the same logits
            padded = np.zeros((4096, int(T.max()), 97), np.float32)
            for k in range(4096):
                padded[k, :T[k]] = logits[row0[k]:row0[k + 1]]
            rt, rs = octc.ctc_greedy(padded, T.astype(np.int32), 0)
            np.testing.assert_array_equal(tokens, rt)
            np.testing.assert_array_equal(scores, rs)
            if dtype == F32:
                ks = sorted(ref)
                texts = [tok.decode(tokens[k]) for k in ks]
                ref_texts = [tok.ctc_decode(ref[k][None])[0][0] for k in ks]
                octc.assert_texts_match_or_tie(texts, ref_texts, [ref[k] for k in ks], tok)
    finally:
        ctx.close()
