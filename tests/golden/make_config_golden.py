"""Full-size expected outputs of the CPU oracle for the BASELINE configurations the GPU box cannot afford to
recompute on its host cores in every test run.

    python tests/golden/make_config_golden.py            # ~3 minutes on 8 cores

Writes (inputs are re-created in the tests from the seeds stored with the outputs):

  tests/golden/config4_page.npz   configs[3]: 1 page 1024x1024, ~220 words, english_g2 head (97 classes)
  tests/golden/config5_pages.npz  configs[4]: 2 pages 2048x2048, ~500 words, zh_sim_g2 head (6719 classes)

Per file: the detector maps of page 0 on a sample of map pixels (random-weight EAST forward through
oracle/nets.py), and per page the kept boxes after decode + LANMS + NMS of the page's teacher maps with both
trigonometry modes of oracle/east_decode.py ('rounded64' = the GPU's arithmetic, 'numpy32' = the reference's own),
the recognised strings and scores, and for every time step of every text row the two largest logits and their
classes (what a tie check needs; the full logits of config 5 would be 0.5 GB).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ocr_amd import synth  # noqa: E402
from ocr_amd.utils.text import get_characters  # noqa: E402
from oracle import ctc as octc, image as oimg, nets, pipeline as opipe  # noqa: E402

CONFIGS = {
    'config4_page': dict(page_seed=1234 + 4, pages=1, size=1024, words=220, east_seed=1, crnn_seed=2, charset='english_g2'),
    'config5_pages': dict(page_seed=1234 + 5, pages=2, size=2048, words=500, east_seed=1, crnn_seed=2, charset='zh_sim_g2'),
}
MAP_SAMPLES = 8192


def make(name, cfg):
    t0 = time.time()
    tok = octc.CharTokenizer(get_characters(cfg['charset']))
    east = nets.init_east_params(seed=cfg['east_seed'])
    crnn = nets.init_crnn_params(vocab_size=tok.vocab_size, seed=cfg['crnn_seed'])
    S = cfg['size']
    images, maps, _ = synth.make_pages(cfg['page_seed'], cfg['pages'], S, S, cfg['words'])
    out = {k: np.array(v) for k, v in cfg.items()}
    out['vocab_size'] = np.array(tok.vocab_size)
    # detector maps of page 0 on a seeded sample of map pixels + the per-channel means of the whole map
    ref_maps = nets.east_forward(east, oimg.prepare_east_input(images[0])[None])[0].numpy()
    flat = ref_maps.reshape(-1, 6)
    pick = np.sort(np.random.default_rng(7).choice(len(flat), MAP_SAMPLES, replace=False))
    out['maps_pixels'] = pick.astype(np.int64)
    out['maps_values'] = flat[pick]
    out['maps_channel_means'] = flat.astype(np.float64).mean(axis=0)
    print('%s: EAST forward %.1f s' % (name, time.time() - t0))
    for b in range(cfg['pages']):
        res = opipe.ocr_page(images[b], east, crnn, tok, maps_override=maps[b], run_network=False, trig='rounded64')
        _, boxes32, _ = opipe.detect_page(images[b], east, maps_override=maps[b], run_network=False, trig='numpy32')
        order = [np.argsort(-lg, axis=1, kind='stable')[:, :2] for lg in res['logits']]
        top = [np.take_along_axis(lg, o, axis=1) for lg, o in zip(res['logits'], order)]
        p = 'page%d_' % b
        out[p + 'boxes'] = res['boxes']
        out[p + 'boxes_numpy32'] = boxes32
        out[p + 'index'] = np.asarray(res['index'], dtype=np.int32)
        out[p + 'texts'] = np.array(res['texts'])                     # unicode array, no pickle
        out[p + 'scores'] = np.asarray(res['scores'], dtype=np.float32)
        out[p + 'steps'] = np.array([len(lg) for lg in res['logits']], dtype=np.int32)
        out[p + 'top2_class'] = np.concatenate(order).astype(np.int32)
        out[p + 'top2_logit'] = np.concatenate(top).astype(np.float32)
        out[p + 'logit_absmax'] = np.array([np.abs(lg).max() for lg in res['logits']], dtype=np.float32)
        print('%s: page %d, %d boxes, %d strings, %.1f s' % (name, b, len(res['boxes']), len(res['texts']), time.time() - t0))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    for name, cfg in CONFIGS.items():
        if len(sys.argv) == 1 or name in sys.argv[1:]:
            make(name, cfg)
