"""Development aid: shape sweep of both network forwards, float32 and bfloat16, against the oracle (prints worst errors)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd.engine import HipContext
from ocr_amd._lib import BF16, F32
from oracle import image as oimg, nets

def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))

ctx = HipContext(0)
rng = np.random.default_rng(99)
params = nets.init_east_params(seed=3)
ctx.east_load_weights(params)
worst = {F32: 0.0, BF16: 0.0}
for shape in ((1, 32, 32), (1, 32, 64), (3, 64, 32), (1, 96, 416), (2, 160, 96), (1, 32, 1056), (5, 224, 32), (2, 288, 288)):
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    x = np.stack([oimg.prepare_east_input(i) for i in img])
    ref = nets.east_forward(params, x).numpy()
    for dt in (F32, BF16):
        got = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=dt).cpu().numpy()
        e = rel(got, ref)
        worst[dt] = max(worst[dt], e)
        flag = '' if e < (1e-4 if dt == F32 else 1e-2) else '   <<<<<< OVER'
        print('east %s %s: %.2e%s' % (shape, 'f32' if dt == F32 else 'bf16', e, flag), flush=True)
cp = nets.init_crnn_params(vocab_size=97, seed=4)
ctx.crnn_load_weights(cp)
for n, W in ((1, 64), (1, 128), (2, 320), (5, 64), (17, 192), (33, 128), (65, 64), (3, 1088), (1, 2048), (130, 128)):
    x = rng.uniform(-1, 1, (n, 64, W, 1)).astype(np.float32)
    ref = nets.crnn_forward(cp, x).numpy()
    for dt in (F32, BF16):
        out = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, want_tokens=True, compute_dtype=dt)
        torch.cuda.synchronize(); ctx.check_errors()
        e = rel(out['logits'].cpu().numpy(), ref)
        worst[dt] = max(worst[dt], e)
        flag = '' if e < (1e-4 if dt == F32 else 1e-2) else '   <<<<<< OVER'
        print('crnn n=%d W=%d %s: %.2e%s' % (n, W, 'f32' if dt == F32 else 'bf16', e, flag), flush=True)
# ragged packed batches
for widths in ([64], [64, 64, 128], [128] * 7 + [64] * 9 + [320], [64] * 40 + [1088], list(rng.choice([64, 128, 192, 256], 97))):
    widths = np.array(sorted(widths), dtype=np.int32)
    crops = [rng.uniform(-1, 1, (64, int(w))).astype(np.float32) for w in widths]
    packed = torch.from_numpy(np.concatenate([c.ravel() for c in crops])).cuda()
    refs = [nets.crnn_forward(cp, c[None, :, :, None]).numpy()[0] for c in crops]
    for dt in (F32, BF16):
        out = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=dt)
        torch.cuda.synchronize(); ctx.check_errors()
        lg = out['logits'].cpu().numpy()
        row0 = out['row0'].cpu().numpy() if 'row0' in out else None
        e, r = 0.0, 0
        for i, rf in enumerate(refs):
            T = rf.shape[0]
            g = lg[r:r + T] if lg.ndim == 2 else lg[i, :T]
            r += T
            e = max(e, rel(g, rf))
        worst[dt] = max(worst[dt], e)
        flag = '' if e < (1e-4 if dt == F32 else 1e-2) else '   <<<<<< OVER'
        print('packed %d crops (max W %d) %s: %.2e%s' % (len(widths), widths.max(), 'f32' if dt == F32 else 'bf16', e, flag), flush=True)
print('worst f32 %.2e bf16 %.2e' % (worst[F32], worst[BF16]))
ctx.close()
