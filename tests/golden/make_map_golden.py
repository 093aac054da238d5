"""Generates tests/golden/map_golden.json with the reference's own `to_json` / `dump_json`
(utils/generic_utils.py:72-104, utils/file_utils.py:333-339): the `map.json` text the reference writes for a fixed
result dictionary.  The dictionary itself is rebuilt by `sample_infos()` below, which the test imports too.

    python tests/golden/make_map_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def sample_infos(k=0):
    """What `BaseOCR.infer` hands to its callbacks (base_ocr.py:242-253): the result without the image."""
    rng = np.random.default_rng(7 + k)
    boxes = rng.uniform(0, 1, (3, 4)).astype(np.float32)
    return {
        'filename': 'scans/page %d.jpg' % k,
        'boxes': {'boxes': boxes, 'format': 'xyxy'},
        'ocr': [
            {'boxes': boxes[0], 'rows': boxes[:1], 'text': 'Im', 'format': 'xyxy', 'scores': np.array([-0.03125], np.float32)},
            {'boxes': boxes[1], 'rows': boxes[1:], 'text': 'Im Sionstal = \nKoelnmesse', 'format': 'xyxy',
             'scores': rng.uniform(-2, 0, 2).astype(np.float32)},
        ],
    }


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from utils.file_utils import dump_json
    from utils.generic_utils import to_json
    data = {}
    for k in range(2):
        infos = sample_infos(k)
        data[infos['filename']] = to_json(infos)
    dump_json(os.path.join(OUT, 'map_golden.json'), data, indent=4)
    print('wrote', os.path.join(OUT, 'map_golden.json'))


if __name__ == '__main__':
    main()
