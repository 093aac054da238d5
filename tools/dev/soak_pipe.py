"""Development aid: the batched GPU pipeline against the CPU oracle pipeline over random page sizes, word counts and
batch sizes (teacher maps feed decode on both sides; prints mismatches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd import synth
from ocr_amd.pipeline import OcrPipeline
from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
from oracle import ctc as octc, nets, pipeline as opipe

east_params = nets.init_east_params(seed=3)
crnn_params = nets.init_crnn_params(vocab_size=97, seed=4)
tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
pipe = OcrPipeline(east_params, crnn_params)
rng = np.random.default_rng(2024)
bad = 0
n_words = 0
for case in range(14):
    H = int(rng.choice([96, 128, 160, 256, 320])); W = int(rng.choice([96, 128, 224, 384, 512]))
    B = int(rng.integers(1, 4)); nw = int(rng.choice([0, 1, 3, 12, 30]))
    images, maps, _ = synth.make_pages(100 + case, B, H, W, nw)
    got = pipe(torch.from_numpy(images).cuda(), maps_override=torch.from_numpy(maps).cuda())
    for b in range(B):
        ref = opipe.ocr_page(images[b], east_params, crnn_params, tok, maps_override=maps[b], run_network=False)
        try:
            assert len(got[b]['boxes']) == len(ref['boxes'])
            np.testing.assert_array_equal(got[b]['boxes'], ref['boxes'])
            assert list(np.nonzero(got[b]['valid'])[0]) == list(ref['index'])
            octc.assert_texts_match_or_tie([got[b]['texts'][i] for i in ref['index']], ref['texts'], ref['logits'], tok)
            np.testing.assert_allclose([got[b]['scores'][i] for i in ref['index']], ref['scores'], rtol=1e-3, atol=1e-3)
            n_words += len(ref['texts'])
        except AssertionError as e:
            bad += 1
            print('MISMATCH case', case, 'page', b, (H, W, B, nw), str(e)[:300].replace('\n', ' | '), flush=True)
    print('case', case, (H, W, B, nw), 'ok so far, words', n_words, flush=True)
print('cases 14 bad', bad, 'words', n_words)
