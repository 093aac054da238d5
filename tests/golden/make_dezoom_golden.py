"""Generates tests/golden/dezoom_golden.npz with the reference's own `convert_box_format(..., dezoom_factor=f)`
(utils/image/bounding_box/converter.py:74-178,202-232) on seeded relative boxes, in the two forms the OCR path uses:
relative xyxy -> relative xyxy, and relative xyxy -> absolute int32 xyxy as `crop_box` receives it
(processing.py:111-114: normalize_mode='absolute', then the dezoom in pixel units).

    python tests/golden/make_dezoom_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
FACTORS = (0.5, 1.3, 2.0)
IMAGE_HW = (480, 640)


def sample_boxes():
    rng = np.random.default_rng(2024)
    xy = rng.uniform(0, 0.9, (80, 2)).astype(np.float32)
    wh = rng.uniform(0.01, 0.5, (80, 2)).astype(np.float32)
    return np.concatenate([xy, np.minimum(xy + wh, 1.0).astype(np.float32)], axis=1)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from utils.image.bounding_box import convert_box_format
    boxes = sample_boxes()
    gold = {'boxes': boxes, 'factors': np.array(FACTORS), 'image_hw': np.array(IMAGE_HW)}
    for f in FACTORS:
        gold['rel_%g' % f] = np.asarray(convert_box_format(boxes, target='xyxy', source='xyxy', dezoom_factor=f))
        gold['abs_%g' % f] = np.asarray(convert_box_format(boxes, target='xyxy', source='xyxy', dezoom_factor=f,
                                                           normalize_mode='absolute', image_h=IMAGE_HW[0], image_w=IMAGE_HW[1]))
    np.savez(os.path.join(OUT, 'dezoom_golden.npz'), **gold)
    print({k: (v.shape, v.dtype) for k, v in gold.items()})


if __name__ == '__main__':
    main()
