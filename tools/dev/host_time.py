"""Development aid: wall and CPU time of the pipeline's host stages per batch of 16 pages (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ocr_amd import synth
from ocr_amd.pipeline import OcrPipeline
from ocr_amd.weights import random_east_params, random_crnn_params
pipe = OcrPipeline(random_east_params(1), random_crnn_params(97, seed=2), device=0)
images, maps, _ = synth.make_pages(1238, 16, 1024, 1024, 220)
d_img = torch.from_numpy(images).cuda(); d_maps = torch.from_numpy(maps).cuda()
for _ in pipe.run_batches((d_img, d_maps) for _ in range(3)):
    pass
acc = {}
waits = [0.0]
_sync = torch.cuda.Event.synchronize
def _timed_sync(self):
    t = time.perf_counter()
    _sync(self)
    waits[0] += (time.perf_counter() - t) * 1e3
torch.cuda.Event.synchronize = _timed_sync
for name in ('_stage_detect', '_stage_recognise', '_stage_strings'):
    f = getattr(pipe, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            t, c = time.perf_counter(), waits[0]
            r = f(*a, **k)
            acc.setdefault(name, []).append(((time.perf_counter() - t) * 1e3, (time.perf_counter() - t) * 1e3 - (waits[0] - c)))
            return r
        return g
    setattr(pipe, name, wrap())
t0 = time.perf_counter(); c0 = time.process_time()
n = 12
for _ in pipe.run_batches((d_img, d_maps) for _ in range(n)):
    pass
torch.cuda.synchronize()
print('per batch: wall %.1f ms, process CPU %.1f ms' % ((time.perf_counter() - t0) * 1e3 / n, (time.process_time() - c0) * 1e3 / n))
for k, v in acc.items():
    print('%-18s wall %s' % (k, ' '.join('%.1f' % x[0] for x in v[2:8])))
    print('%-18s work %s' % ('', ' '.join('%.1f' % x[1] for x in v[2:8])))
