"""Per-kernel timing of the two network forwards alone (development aid).

    python tools/bench_nets.py [--batch 4] [--size 1024] [--crops 1024] [--wpad 192]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_amd.engine import HipContext  # noqa: E402
from ocr_amd.weights import random_crnn_params, random_east_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--size', type=int, default=1024)
ap.add_argument('--crops', type=int, default=1024)
ap.add_argument('--wpad', type=int, default=192)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--backbone', default='vgg', choices=['vgg', 'resnet'])
ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
ap.add_argument('--opt', action='append', default=[], help='name=value passed to ocrhip_set_option')
args = ap.parse_args()

from ocr_amd._lib import BF16, F32  # noqa: E402
DT = BF16 if args.dtype == 'bf16' else F32
ctx = HipContext(0)
for o in args.opt:
    k, v = o.split('=')
    ctx.set_option(k, int(v))
ctx.east_load_weights(random_east_params(1))
ctx.crnn_load_weights(random_crnn_params(97, seed=2, backbone=args.backbone))
img = torch.randint(0, 256, (args.batch, args.size, args.size, 3), dtype=torch.uint8, device='cuda')
crops = torch.rand((args.crops, 64, args.wpad), device='cuda') * 2 - 1
ctx.east_forward(img, compute_dtype=DT)
ctx.crnn_forward(crops, want_logits=False, want_tokens=True, compute_dtype=DT)
torch.cuda.synchronize()
for name, fn in (('east', lambda: ctx.east_forward(img, compute_dtype=DT)),
                 ('crnn', lambda: ctx.crnn_forward(crops, want_logits=False, want_tokens=True, compute_dtype=DT))):
    ctx.profile_enable(True)
    for _ in range(args.reps):
        fn()
    stats = ctx.profile_collect()
    ctx.profile_enable(False)
    tot = sum(s['total_ms'] for s in stats) / args.reps
    fl = sum(s['flops'] for s in stats) / args.reps
    print('%s: %.3f ms per forward, %.1f TFLOP/s overall' % (name, tot, fl / tot / 1e9))
    for s in stats:
        ms = s['total_ms'] / args.reps
        print('   %-40s %4d launches %9.3f ms  %7.1f TFLOP/s  %7.1f GB/s' % (
            s['name'], s['launches'] // args.reps, ms, s['flops'] / args.reps / ms / 1e9 if ms else 0,
            s['bytes'] / args.reps / ms / 1e6 if ms else 0))
