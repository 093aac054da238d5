"""Development aid: detector maps with the F(4x4,3x3) kernel (conv3x3_winograd = 2) against F(2x2,3x3) and the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_amd.engine import HipContext  # noqa: E402
from oracle import image as oimg, nets  # noqa: E402

ctx = HipContext(0)
params = nets.init_east_params(seed=1)
ctx.east_load_weights(params)
rng = np.random.default_rng(77)
for shape in ((1, 96, 160, 3), (2, 128, 256, 3)):
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = nets.east_forward(params, np.stack([oimg.prepare_east_input(i) for i in img])).numpy()
    out = {}
    for opt in (1, 2):
        ctx.set_option('conv3x3_winograd', opt)
        out[opt] = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
    for opt in (1, 2):
        d = np.abs(out[opt] - ref)
        print(shape, 'option', opt, 'max abs err', d.max(), 'worst ratio to (1e-5 + 1e-4 |ref|)', (d / (1e-5 + 1e-4 * np.abs(ref))).max(), flush=True)
