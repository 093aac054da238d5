"""Development aid: histogram of the recogniser's padded crop widths for the bench's synthetic pages."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd import synth  # noqa: E402
from ocr_amd.pipeline import OcrPipeline  # noqa: E402
from ocr_amd.utils import image_sizes as S  # noqa: E402
from ocr_amd.weights import random_crnn_params, random_east_params  # noqa: E402

pipe = OcrPipeline(random_east_params(1), random_crnn_params(97, seed=2))
images, maps, _ = synth.make_pages(1234 + 4, 16, 1024, 1024, 220)
st = pipe._stage_detect(torch.from_numpy(images).cuda(), torch.from_numpy(maps).cuda())
st['boxes_ready'].synchronize()
counts = st['h_counts'].numpy()
boxes = st['h_boxes'].numpy()
w = np.concatenate([S.plan_crops(boxes[b, :int(counts[b])].copy(), 1024, 1024)['out_w'] for b in range(16)])
u, c = np.unique(w, return_counts=True)
print('crops', len(w), 'mean width', w.mean())
for a, b in zip(u, c):
    print(a, b)
