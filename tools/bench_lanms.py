"""Timing of decode + LANMS alone on synthetic pages (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_amd import synth
from ocr_amd.engine import HipContext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = HipContext(0)
_, maps, _ = synth.make_pages(1238, B, 1024, 1024, 220)
d_maps = torch.from_numpy(maps).cuda()
dec = ctx.east_decode(d_maps, (1024, 1024), 0.35)
print('candidates', dec['counts'].cpu().numpy())
for label, kw in (('full', {}), ('max_iter=1', {'max_iter': 1}), ('max_iter=2', {'max_iter': 2}),
                  ('no nms', {'nms_threshold': 0.5, 'merge_threshold': 0.3})):
    args = dict(merge_threshold=0.3, nms_threshold=0.2)
    args.update(kw)
    out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], **args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], **args)
    torch.cuda.synchronize()
    print('%-12s %.3f ms  n_iter %s kept %s' % (label, (time.perf_counter() - t0) / 5 * 1e3,
                                               out['n_iter'].cpu().numpy(), out['counts'].cpu().numpy()))
