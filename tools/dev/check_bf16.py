"""Development aid: per-tap error of the bfloat16-tensor path against the float32 oracle (prints, asserts nothing)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd.engine import HipContext
from ocr_amd._lib import BF16
from oracle import image as oimg, nets


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(1e-30, np.linalg.norm(b.ravel())))


ctx = HipContext(0)
params = nets.init_east_params(seed=1)
ctx.east_load_weights(params)
rng = np.random.default_rng(123)
for shape in ((2, 96, 160, 3), (1, 256, 320, 3)):
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    x = np.stack([oimg.prepare_east_input(i) for i in img])
    ref, taps, merged = nets.east_forward(params, x, return_taps=True)
    for opt in (1, 0):
        ctx.set_option('bf16_tensors', opt)
        got = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=BF16).cpu().numpy()
        torch.cuda.synchronize()
        line = ['east %s tensors16=%d' % (shape, opt)]
        for i in range(5):
            line.append('pool%d %.2e' % (i + 1, rel(ctx.east_activation('pool%d' % (i + 1)), taps[i].numpy())))
        line.append('merge7 %.2e' % rel(ctx.east_activation('merge7'), merged.numpy()))
        line.append('maps %.2e max %.2e' % (rel(got, ref.numpy()), np.abs(got - ref.numpy()).max()))
        print(' | '.join(line), flush=True)
ctx.set_option('bf16_tensors', 1)

for V, cases in ((97, [(4, 160), (3, 128), (33, 192), (2, 64), (1, 48), (2, 1040)]), (6719, [(3, 128)])):
    cp = nets.init_crnn_params(vocab_size=V, seed=2)
    ctx.crnn_load_weights(cp)
    for n, W in cases:
        x = rng.uniform(-1, 1, (n, 64, W, 1)).astype(np.float32)
        f = nets.crnn_features(cp, x).numpy()
        ref = nets.crnn_forward(cp, x).numpy()
        for opt in (1, 0):
            ctx.set_option('bf16_tensors', opt)
            out = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, want_tokens=True, compute_dtype=BF16)
            torch.cuda.synchronize()
            ctx.check_errors()
            lg = out['logits'].cpu().numpy()
            ft = ctx.crnn_activation('features').reshape(f.shape)
            s0 = ctx.crnn_activation('seq0')
            print('crnn V=%d n=%d W=%d tensors16=%d | features %.2e | seq0 finite %s | logits %.2e max/scale %.2e' % (
                V, n, W, opt, rel(ft, f), np.isfinite(s0).all(), rel(lg, ref), np.abs(lg - ref).max() / np.abs(ref).max()), flush=True)
        ctx.set_option('bf16_tensors', 1)
cp = nets.init_crnn_params(vocab_size=97, seed=2)
ctx.crnn_load_weights(cp)
widths = np.array([64] * 3 + [128] * 35 + [192] * 2 + [320], dtype=np.int32)
packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
b = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
for opt in (1, 0):
    ctx.set_option('bf16_tensors', opt)
    a = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
    ctx.check_errors()
    print('packed tensors16=%d rel vs f32 GPU %.2e' % (opt, rel(a, b)), flush=True)
ctx.close()
