"""Development aid: time of ocrhip_ctc_beam_search alone over batch shapes (prints, asserts nothing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd.engine import HipContext  # noqa: E402

ctx = HipContext(0)
g = torch.Generator(device='cuda').manual_seed(5)
for B, T, V, K, scale in ((10512, 32, 97, 10, 1.0), (10512, 32, 97, 10, 8.0), (1440, 256, 97, 10, 1.0), (1440, 256, 97, 10, 8.0),
                          (1440, 256, 97, 1, 1.0), (1440, 256, 97, 4, 1.0), (1440, 256, 97, 16, 1.0), (256, 64, 6719, 10, 4.0)):
    logits = (torch.randn((B, T, V), device='cuda', generator=g) * scale).contiguous()
    ctx.ctc_beam_search(logits, num_beams=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        tok, sc = ctx.ctc_beam_search(logits, num_beams=K)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    ctx.profile_enable(True)
    ctx.ctc_beam_search(logits, num_beams=K)
    st = ctx.profile_collect()
    ctx.profile_enable(False)
    print('   ' + ', '.join('%s %.3f ms' % (x['name'], x['total_ms']) for x in st))
    print('B=%d T=%d V=%d K=%d scale=%.0f: %.3f ms  (%.2f us per sequence-step, %.1f us per step of a wave)' % (
        B, T, V, K, scale, ms, ms * 1e3 / (B * T), ms * 1e3 / T), flush=True)
ctx.close()
