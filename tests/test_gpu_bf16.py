"""The bfloat16-tensor path (`compute_dtype = BF16`, option bf16_tensors = 1, the default): bfloat16 activations in
HBM, tile kernels with float32 accumulation, bfloat16 recurrent product with float32 cell state.  The reference computes
in float32; the bar for this variant is 1e-2 against the float32 oracle (relative error of a tensor,
||got - ref|| / ||ref||), checked tap by tap so that a wrong layer shows where it is."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import ctc as octc, image as oimg, nets  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from ocr_amd.engine import HipContext
    c = HipContext(0)
    yield c
    c.close()


def rel(a, b):
    return float(np.linalg.norm((np.asarray(a) - np.asarray(b)).ravel()) / max(1e-30, np.linalg.norm(np.asarray(b).ravel())))


# (2, 96, 160): partial tiles everywhere, conv5 maps of 6 x 10; (1, 256, 320): several tiles per map and both pooled
# tile shapes; (3, 64, 96): three images, maps smaller than a tile
@pytest.mark.parametrize('shape', [(2, 96, 160, 3), (1, 256, 320, 3), (3, 64, 96, 3)])
def test_detector_taps_and_maps(ctx, shape):
    from ocr_amd._lib import BF16
    params = nets.init_east_params(seed=1)
    ctx.east_load_weights(params)
    rng = np.random.default_rng(shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    x = np.stack([oimg.prepare_east_input(i) for i in img])
    ref, taps, merged = nets.east_forward(params, x, return_taps=True)
    for in_t in (torch.from_numpy(img).cuda(), torch.from_numpy(x).cuda()):
        got = ctx.east_forward(in_t, compute_dtype=BF16).cpu().numpy()
        for i in range(5):
            assert rel(ctx.east_activation('pool%d' % (i + 1)), taps[i].numpy()) <= 1e-2, 'pool%d' % (i + 1)
        assert rel(ctx.east_activation('merge7'), merged.numpy()) <= 1e-2
        assert rel(got, ref.numpy()) <= 1e-2 and np.abs(got - ref.numpy()).max() <= 5e-2
    # the float32-tensor variant (operands rounded on their way into LDS) is the same arithmetic up to where the
    # roundings happen: the two agree far inside the tolerance either has against the oracle
    ctx.set_option('bf16_tensors', 0)
    try:
        old = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=BF16).cpu().numpy()
    finally:
        ctx.set_option('bf16_tensors', 1)
    assert rel(got, old) <= 1e-2 and np.abs(got - old).max() > 0
    # the first two layers as two launches (float32 first layer writing a bfloat16 tensor) instead of one: both forms
    # against the oracle, and close to each other at the first tap
    ctx.set_option('bf16_fuse_first', 0)
    try:
        two = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=BF16).cpu().numpy()
        pool1_two = ctx.east_activation('pool1')
    finally:
        ctx.set_option('bf16_fuse_first', 1)
    assert rel(pool1_two, taps[0].numpy()) <= 1e-2 and rel(two, ref.numpy()) <= 1e-2
    ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=BF16)
    assert rel(ctx.east_activation('pool1'), pool1_two) <= 1e-2


@pytest.mark.parametrize('n,Wpad,V', [(4, 160, 97), (33, 192, 97), (2, 64, 97), (1, 48, 97), (2, 1040, 97), (3, 128, 6719)])
def test_recogniser_features_and_logits(ctx, n, Wpad, V):
    from ocr_amd._lib import BF16
    cp = nets.init_crnn_params(vocab_size=V, seed=2)
    ctx.crnn_load_weights(cp)
    rng = np.random.default_rng(n * 7 + Wpad)
    x = rng.uniform(-1, 1, (n, 64, Wpad, 1)).astype(np.float32)
    f = nets.crnn_features(cp, x).numpy()
    ref = nets.crnn_forward(cp, x).numpy()
    out = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, want_tokens=True, compute_dtype=BF16)
    torch.cuda.synchronize()
    ctx.check_errors()
    lg = out['logits'].cpu().numpy()
    assert rel(ctx.crnn_activation('features').reshape(f.shape), f) <= 1e-2
    assert rel(lg, ref) <= 1e-2 and np.abs(lg - ref).max() <= 5e-2 * max(1.0, float(np.abs(ref).max()))
    # greedy decoding on the GPU == the oracle's decoding of the same logits, bit for bit
    rt, rs = octc.ctc_greedy(lg, None, 0)
    np.testing.assert_array_equal(out['tokens'].cpu().numpy(), rt)
    np.testing.assert_array_equal(out['scores'].cpu().numpy(), rs)


def test_ragged_batch_and_the_per_step_fallback(ctx):
    """One packed call over several widths == the float32 path to 1e-2; `lstm_persistent = 0` (one float32 launch per
    time step, the path taken after a hand-over time-out or on a shared GPU) stays within the same bar; a sequence's
    logits do not depend on what else is in the batch beyond float32 summation order."""
    from ocr_amd._lib import BF16
    cp = nets.init_crnn_params(vocab_size=97, seed=2)
    ctx.crnn_load_weights(cp)
    rng = np.random.default_rng(5)
    widths = np.array([64] * 3 + [128] * 35 + [192] * 2 + [320], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    f32 = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
    a = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=BF16)
    ctx.check_errors()
    assert rel(a['logits'].cpu().numpy(), f32) <= 1e-2
    T = widths // 4 - 1
    row0 = np.concatenate([[0], np.cumsum(T)])
    lg = a['logits'].cpu().numpy()
    padded = np.zeros((len(widths), int(T.max()), 97), np.float32)
    for i in range(len(widths)):
        padded[i, :T[i]] = lg[row0[i]:row0[i + 1]]
    rt, rs = octc.ctc_greedy(padded, T.astype(np.int32), 0)
    np.testing.assert_array_equal(a['tokens'].cpu().numpy(), rt)
    np.testing.assert_array_equal(a['scores'].cpu().numpy(), rs)
    ctx.set_option('lstm_persistent', 0)
    try:
        b = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
    finally:
        ctx.set_option('lstm_persistent', 2)
    assert rel(b, f32) <= 1e-2
    # 32 hidden units per workgroup of the recurrence instead of 64: the same sums in the same order
    ctx.set_option('lstm16_units', 32)
    try:
        u32 = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
        ctx.check_errors()
    finally:
        ctx.set_option('lstm16_units', 64)
    np.testing.assert_array_equal(u32, lg)
    # the widest crop alone
    k = len(widths) - 1
    alone = packed[64 * int(widths[:k].sum()):].contiguous()
    c = ctx.crnn_forward_packed(alone, widths[k:], want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
    assert rel(c, lg[row0[k]:]) <= 2e-3
