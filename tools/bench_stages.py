"""Per-kernel time, TFLOP/s and GB/s (algorithmic) of one pipeline step on bench.py's workload (development aid)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_amd import synth  # noqa: E402
from ocr_amd.pipeline import OcrPipeline  # noqa: E402
from ocr_amd.weights import random_crnn_params, random_east_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--steps', type=int, default=4)
ap.add_argument('--opt', action='append', default=[], help='name=value passed to ocrhip_set_option')
args = ap.parse_args()
pipe = OcrPipeline(random_east_params(1), random_crnn_params(97, seed=2), device=0)
for o in args.opt:
    k, v = o.split('=')
    pipe.ctx.set_option(k, int(v))
images, maps, _ = synth.make_pages(1238, args.batch, 1024, 1024, 220)
d_img = torch.from_numpy(images).cuda()
d_maps = torch.from_numpy(maps).cuda()
for _ in pipe.run_batches((d_img, d_maps) for _ in range(2)):
    pass
pipe.ctx.profile_enable(True)
for _ in pipe.run_batches((d_img, d_maps) for _ in range(args.steps)):
    pass
torch.cuda.synchronize()
stats = pipe.ctx.profile_collect()
tot = sum(s['total_ms'] for s in stats) / args.steps
print('kernel time per step %.2f ms' % tot)
for s in stats:
    ms = s['total_ms'] / args.steps
    print('  %-46s %5.1f launches %8.3f ms %7.1f TFLOP/s %7.1f GB/s' % (
        s['name'], s['launches'] / args.steps, ms, s['flops'] / args.steps / ms / 1e9 if ms else 0,
        s['bytes'] / args.steps / ms / 1e6 if ms else 0))
