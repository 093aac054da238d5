"""Developer aid: host-side timing of the pipeline stages and of the yields of run_batches (needs a GPU)."""
import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from ocr_amd import synth
from ocr_amd.pipeline import OcrPipeline
from ocr_amd.weights import random_east_params, random_crnn_params
pipe = OcrPipeline(random_east_params(1), random_crnn_params(97, seed=2), device=0)
images, maps, _ = synth.make_pages(1238, 8, 1024, 1024, 220)
d_img = torch.from_numpy(images).cuda(); d_maps = torch.from_numpy(maps).cuda()
def run(n):
    ts = []
    t0 = time.perf_counter()
    for out in pipe.run_batches((d_img, d_maps) for _ in range(n)):
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return ts, time.perf_counter() - t0
run(2)
torch.cuda.synchronize()
for n in (5, 10):
    ts, tot = run(n)
    print(n, 'total %.1f ms' % (tot * 1e3), 'yield times', ['%.1f' % (t * 1e3) for t in ts])
# stage timing on host
import types
for name in ('_stage_detect', '_stage_recognise', '_stage_strings'):
    f = getattr(pipe, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); acc[name].append((time.perf_counter() - t) * 1e3); return r
        return g
    setattr(pipe, name, wrap())
acc = {'_stage_detect': [], '_stage_recognise': [], '_stage_strings': []}
ts, tot = run(6)
for k, v in acc.items(): print(k, ['%.1f' % x for x in v])
