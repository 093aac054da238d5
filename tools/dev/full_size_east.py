import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ocr_amd.engine import HipContext
from ocr_amd._lib import BF16, F32
from oracle import image as oimg, nets
ctx = HipContext(0)
params = nets.init_east_params(seed=3)
ctx.east_load_weights(params)
rng = np.random.default_rng(1)
for shape in ((2, 512, 512), (1, 1024, 1024), (1, 736, 1280)):
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    x = np.stack([oimg.prepare_east_input(i) for i in img])
    ref = nets.east_forward(params, x).numpy()
    for dt, name, tol in ((F32, 'f32', 1e-4), (BF16, 'bf16', 1e-2)):
        got = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=dt).cpu().numpy()
        l2 = float(np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel()))
        mx = float(np.abs(got - ref).max())
        print(shape, name, 'rel L2 %.2e max abs %.2e' % (l2, mx), 'OK' if l2 <= tol else 'OVER', flush=True)
ctx.close()
