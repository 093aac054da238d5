"""Generates tests/golden/*.npz by running the reference's own NumPy functions.

Run from anywhere inside the build container (the reference is mounted at /root/reference):

    python tests/golden/make_golden.py

Only the parts of the reference that import without Keras are used (they dispatch NumPy inputs to
NumPy): `restore_polys_from_map`, `convert_box_format`, `compute_iou`, `crop_box`,
`get_output_size` and the 'east' / 'easyocr' normalisers.  The outputs are small arrays (inputs +
expected outputs); no reference source is copied.  The GPU box never sees the reference, only
these files.
"""
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def synth_map(rng, h, w, n_words, max_angle):
    """A (h,w,6) raw EAST-like sigmoid map with a few rotated 'words' and unequal scores."""
    out = np.zeros((h, w, 6), dtype=np.float32)
    out[..., 0] = rng.uniform(0.0, 0.3, (h, w))
    out[..., 1:5] = rng.uniform(0.0, 0.02, (h, w, 4))
    out[..., 5] = 0.5
    for _ in range(n_words):
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        hh, hw = rng.integers(1, 4), rng.integers(2, 10)
        y0, y1 = max(0, cy - hh), min(h, cy + hh + 1)
        x0, x1 = max(0, cx - hw), min(w, cx + hw + 1)
        ang = rng.uniform(-max_angle, max_angle)
        ys, xs = np.mgrid[y0:y1, x0:x1]
        out[y0:y1, x0:x1, 0] = rng.uniform(0.36, 0.999, (y1 - y0, x1 - x0))
        # distances (in input pixels / 512) to the four edges of the word box
        out[y0:y1, x0:x1, 1] = ((ys - y0) * 4 + rng.uniform(2, 6, ys.shape)) / 512
        out[y0:y1, x0:x1, 2] = ((y1 - ys) * 4 + rng.uniform(2, 6, ys.shape)) / 512
        out[y0:y1, x0:x1, 3] = ((xs - x0) * 4 + rng.uniform(2, 6, ys.shape)) / 512
        out[y0:y1, x0:x1, 4] = ((x1 - xs) * 4 + rng.uniform(2, 6, ys.shape)) / 512
        out[y0:y1, x0:x1, 5] = 0.5 + ang / np.pi + rng.uniform(-0.01, 0.01, ys.shape)
    return out.astype(np.float32)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from models.detection.east import restore_polys_from_map
    from utils.image.bounding_box import compute_iou, convert_box_format, crop_box
    from utils.image.image_processing import get_output_size
    from utils.image import get_image_normalization_fn

    rng = np.random.default_rng(20261111)
    gold = {}

    # ---- RBOX decode (east.py:102-187) + poly -> xyxy (converter.py:151-166)
    cases = [(64, 64, 12, 0.0), (64, 64, 14, 0.35), (48, 80, 10, 0.2), (40, 40, 30, 0.6)]
    for k, (h, w, n_words, max_angle) in enumerate(cases):
        m = synth_map(rng, h, w, n_words, max_angle)
        thr = 0.35
        res = restore_polys_from_map(
            score_map=m[..., :1], geo_map=m[..., 1:5] * 512, theta_map=(m[..., 5:6] - 0.5) * np.pi,
            normalize=True, threshold=thr,
            input_shape=np.array([h * 4, w * 4]), output_shape=np.array([h, w]))
        xyxy = convert_box_format(res['boxes'], source='poly', target='xyxy')
        gold['decode%d_map' % k] = m
        gold['decode%d_thr' % k] = np.float32(thr)
        gold['decode%d_polys' % k] = res['boxes']
        gold['decode%d_scores' % k] = res['scores']
        gold['decode%d_xyxy' % k] = np.asarray(xyxy)
        print('decode case', k, 'boxes', len(res['boxes']), res['boxes'].dtype, np.asarray(xyxy).dtype)

    # ---- IoU matrices (metrics.py:18-46)
    pts = rng.uniform(0, 1, (200, 2, 2)).astype(np.float32)
    boxes = np.concatenate([pts.min(1), pts.min(1) + rng.uniform(0.01, 0.2, (200, 2)).astype(np.float32)], 1)
    boxes[::17] = 0                                         # zero boxes: 0/0 -> 0
    gold['iou_boxes'] = boxes
    gold['iou_matrix'] = np.asarray(compute_iou(boxes, as_matrix=True, source='xyxy'))
    gold['iou_pairs'] = np.asarray(compute_iou(boxes[:100], boxes[100:], source='xyxy'))

    # ---- relative -> absolute + crop (converter.py:180-200, processing.py:111-132)
    image = np.arange(37 * 53 * 3, dtype=np.int64).reshape(37, 53, 3)
    rel = np.array([[0.1, 0.2, 0.5, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.02, 0.1, 0.3, 0.5],
                    [0.5, 0.5, 1.03, 1.01], [-0.2, 0.1, 0.4, 0.6], [0.3, 0.3, 0.3, 0.6],
                    [0.9999, 0.0001, 0.99999, 0.5]], dtype=np.float32)
    gold['crop_image_shape'] = np.array(image.shape)
    gold['crop_rel'] = rel
    abs_list, shapes, sums = [], [], []
    for b in rel:
        pos, crop = crop_box(image, {'boxes': b[None], 'format': 'xyxy'})
        abs_list.append([int(np.asarray(c).reshape(-1)[0]) for c in pos])
        shapes.append(crop.shape)
        sums.append(int(crop.sum()))
    gold['crop_abs'] = np.array(abs_list, dtype=np.int32)
    gold['crop_shapes'] = np.array(shapes, dtype=np.int32)
    gold['crop_sums'] = np.array(sums, dtype=np.int64)

    # ---- size rules (image_processing.py:144-213)
    hs, ws, outs = [], [], []
    for h in range(8, 200):
        for w in (h // 2 + 1, h, 2 * h + 3, 5 * h + 1, 640, 1000):
            o = get_output_size(np.zeros((h, w, 1), np.float32), size=(64, None), multiples=64,
                                preserve_aspect_ratio=True)
            hs.append(h), ws.append(w), outs.append(np.asarray(o))
    gold['crnn_size_h'] = np.array(hs, dtype=np.int32)
    gold['crnn_size_w'] = np.array(ws, dtype=np.int32)
    gold['crnn_size_out'] = np.array(outs, dtype=np.int32)
    hw_in, hw_out = [], []
    for h, w in [(480, 640), (512, 512), (1, 1), (31, 33), (1024, 1000), (1080, 1920), (2048, 2047)]:
        o = get_output_size(np.zeros((h, w, 3), np.float32), size=None, multiples=32)
        hw_in.append((h, w)), hw_out.append(np.asarray(o))
    gold['east_size_in'] = np.array(hw_in, dtype=np.int32)
    gold['east_size_out'] = np.array(hw_out, dtype=np.int32)

    # ---- normalisations (image_normalization.py:57-58)
    x3 = rng.uniform(0, 1, (5, 7, 3)).astype(np.float32)
    x1 = rng.uniform(0, 1, (5, 7, 1)).astype(np.float32)
    gold['norm_x3'] = x3
    gold['norm_east'] = np.asarray(get_image_normalization_fn('east')(x3))
    gold['norm_x1'] = x1
    gold['norm_easyocr'] = np.asarray(get_image_normalization_fn('easyocr')(x1))

    np.savez_compressed(os.path.join(OUT, 'reference_vectors.npz'), **gold)
    print('wrote', os.path.join(OUT, 'reference_vectors.npz'))


if __name__ == '__main__':
    main()
