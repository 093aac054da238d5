"""GPU parity of the two networks (f32 MFMA path) against the torch-CPU float32 oracle, through the
C ABI.  Tolerance: 1e-4 relative on the maps / logits (different summation order, same precision)."""
import numpy as np
import pytest

LSTM_DEFAULT = 2          # the library's default `lstm_persistent`

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import image as oimg, nets  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from ocr_amd.engine import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _close(got, ref, rtol=1e-4, atol=2e-5):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    bad = err > tol
    assert not bad.any(), 'max err %.3e (ref scale %.3e), %d / %d out of tolerance' % (
        err.max(), np.abs(ref).max(), int(bad.sum()), bad.size)


@pytest.mark.parametrize('B,H,W', [(1, 64, 96), (2, 128, 64)])
def test_east_forward_matches_oracle(ctx, B, H, W):
    params = nets.init_east_params(seed=1)
    assert nets.count_params(params) == 15098726          # "15.099 M" in the reference notebook
    ctx.east_load_weights(params)
    rng = np.random.default_rng(B * 100 + H)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    x = np.stack([oimg.prepare_east_input(i) for i in img])
    ref, taps, merged = nets.east_forward(params, x, return_taps=True)
    for in_t in (torch.from_numpy(img).cuda(), torch.from_numpy(x).cuda()):
        maps = ctx.east_forward(in_t)
        torch.cuda.synchronize()
        for i in range(5):
            # activations grow with depth under random weights: compare relative to their scale
            t = taps[i].numpy()
            _close(ctx.east_activation('pool%d' % (i + 1)), t, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(t).max())))
        m = merged.numpy()
        _close(ctx.east_activation('merge7'), m, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(m).max())))
        _close(maps.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


# V = 6719 is the simplified-Chinese head, len(zh_sim_g2 characters) + 1 (crnn_arch.py:153-160)
@pytest.mark.parametrize('n,Wpad,V', [(3, 128, 97), (2, 64, 97), (33, 192, 211), (5, 208, 97), (1, 48, 97), (2, 1040, 97),
                                      (3, 128, 6719)])
def test_crnn_forward_matches_oracle(ctx, n, Wpad, V):
    params = nets.init_crnn_params(vocab_size=V, seed=2)
    ctx.crnn_load_weights(params)
    assert ctx.vocab_size == V
    rng = np.random.default_rng(n + Wpad)
    x = rng.uniform(-1, 1, (n, 64, Wpad, 1)).astype(np.float32)
    feats = nets.crnn_features(params, x)
    ref = nets.crnn_forward(params, x).numpy()
    out = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, want_tokens=True)
    torch.cuda.synchronize()
    f = feats.numpy()
    _close(ctx.crnn_activation('features').reshape(f.shape), f, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(f).max())))
    got = out['logits'].cpu().numpy()
    _close(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    # greedy decode of the GPU logits, on the GPU, equals the oracle decode of the same logits
    from oracle import ctc as octc
    rt, rs = octc.ctc_greedy(got, None, 0)
    np.testing.assert_array_equal(out['tokens'].cpu().numpy(), rt)
    np.testing.assert_array_equal(out['scores'].cpu().numpy(), rs)


def test_crnn_forward_packed_ragged_matches_per_row_oracle(ctx):
    """All widths in one call == the reference's one-forward-per-row + pad_batch + ctc_decode(lengths)."""
    from oracle import ctc as octc
    V = 97
    params = nets.init_crnn_params(vocab_size=V, seed=7)
    ctx.crnn_load_weights(params)
    rng = np.random.default_rng(3)
    widths = np.array([64] * 3 + [128] * 35 + [192] * 2 + [320], dtype=np.int32)
    crops = [rng.uniform(-1, 1, (64, int(w))).astype(np.float32) for w in widths]
    packed = torch.from_numpy(np.concatenate([c.reshape(-1) for c in crops])).cuda()
    out = ctx.crnn_forward_packed(packed, widths, want_logits=True)
    torch.cuda.synchronize()
    logits = out['logits'].cpu().numpy()
    T = widths // 4 - 1
    np.testing.assert_array_equal(out['lengths'].cpu().numpy(), T)
    row0 = np.concatenate([[0], np.cumsum(T)])
    padded = np.zeros((len(widths), int(T.max()), V), np.float32)
    for i, c in enumerate(crops):
        ref = nets.crnn_forward(params, c[None, :, :, None])[0].numpy()
        got = logits[row0[i]:row0[i + 1]]
        _close(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
        padded[i, :T[i]] = got
    rt, rs = octc.ctc_greedy(padded, T.astype(np.int32), 0)
    np.testing.assert_array_equal(out['tokens'].cpu().numpy(), rt)
    np.testing.assert_array_equal(out['scores'].cpu().numpy(), rs)
    with pytest.raises(ValueError):
        ctx.crnn_forward_packed(packed, widths[::-1].copy())          # widths must be non-decreasing


@pytest.mark.parametrize('n,Wpad', [(2, 128), (3, 64), (1, 208), (2, 16)])
def test_crnn_resnet_backbone_matches_oracle(ctx, n, Wpad):
    """`CRNNCTC(backbone='resnet')` (crnn_arch.py:367-425,504-505): 512 features, T = W/4 + 1."""
    V = 97
    params = nets.init_crnn_params(vocab_size=V, seed=5, backbone='resnet')
    ctx.crnn_load_weights(params)
    try:
        assert int(ctx.seq_len(Wpad)) == Wpad // 4 + 1
        rng = np.random.default_rng(n * 1000 + Wpad)
        x = rng.uniform(-1, 1, (n, 64, Wpad, 1)).astype(np.float32)
        f = nets.crnn_features(params, x).numpy()
        ref = nets.crnn_forward(params, x).numpy()
        out = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, want_tokens=True)
        torch.cuda.synchronize()
        assert f.shape == (n, Wpad // 4 + 1, 512)
        _close(ctx.crnn_activation('features').reshape(f.shape), f, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(f).max())))
        got = out['logits'].cpu().numpy()
        _close(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
        from oracle import ctc as octc
        rt, rs = octc.ctc_greedy(got, None, 0)
        np.testing.assert_array_equal(out['tokens'].cpu().numpy(), rt)
        np.testing.assert_array_equal(out['scores'].cpu().numpy(), rs)
        # ragged call: same rows, two widths
        widths = np.array([Wpad] * n + [Wpad + 32], dtype=np.int32)
        extra = rng.uniform(-1, 1, (64, Wpad + 32)).astype(np.float32)
        packed = torch.from_numpy(np.concatenate([x.reshape(-1), extra.reshape(-1)])).cuda()
        res = ctx.crnn_forward_packed(packed, widths, want_logits=True)
        torch.cuda.synchronize()
        T = widths // 4 + 1
        np.testing.assert_array_equal(res['lengths'].cpu().numpy(), T)
        flat = res['logits'].cpu().numpy()
        _close(flat[:n * T[0]].reshape(ref.shape), ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
        ref_extra = nets.crnn_forward(params, extra[None, :, :, None])[0].numpy()
        _close(flat[n * T[0]:], ref_extra, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref_extra).max())))
    finally:
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=V, seed=2))      # leave the VGG recogniser loaded


@pytest.mark.parametrize('hidden,backbone', [(512, 'resnet'), (128, 'vgg'), (512, 'vgg')])
def test_crnn_other_lstm_widths_match_oracle(ctx, hidden, backbone):
    """`lstm_size` other than 256 (crnn_arch.py:485-486; the gen-1 EasyOCR recognisers are ResNet + 512): the width is
    read off the checkpoint; ragged call over two widths against the per-row oracle."""
    V = 97
    params = nets.init_crnn_params(vocab_size=V, hidden=hidden, seed=11, backbone=backbone)
    ctx.crnn_load_weights(params)
    try:
        rng = np.random.default_rng(hidden)
        widths = np.array([96] * 3 + [160] * 2, dtype=np.int32)
        crops = [rng.uniform(-1, 1, (64, int(w))).astype(np.float32) for w in widths]
        packed = torch.from_numpy(np.concatenate([c.reshape(-1) for c in crops])).cuda()
        res = ctx.crnn_forward_packed(packed, widths, want_logits=True)
        torch.cuda.synchronize()
        ctx.check_errors()
        T = ctx.seq_len(widths)
        row0 = np.concatenate([[0], np.cumsum(T)])
        flat = res['logits'].cpu().numpy()
        for i, c in enumerate(crops):
            ref = nets.crnn_forward(params, c[None, :, :, None])[0].numpy()
            _close(flat[row0[i]:row0[i + 1]], ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    finally:
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=V, seed=2))


@pytest.mark.parametrize('V', [97, 128, 129, 6719])
def test_greedy_tokens_without_logits_equal_the_decoding_of_the_logits(ctx, V):
    """Tokens without a logits request (the class layer hands per-block maxima to the CTC kernel, no logits in
    memory) == tokens and scores decoded from the logits the same call writes when asked, bit for bit; uniform and
    ragged calls, vocabularies of one block, exactly one block, one block + 1 and 53 blocks of 128 classes."""
    from oracle import ctc as octc
    params = nets.init_crnn_params(vocab_size=V, seed=9)
    rng = np.random.default_rng(V)
    # duplicate class rows: exact ties inside a block and across blocks, the first class must win
    w, b = params['Prediction.weight'], params['Prediction.bias']
    for dst, src in ((5, 3), (V - 1, 3), (V // 2, 1)):
        w[dst], b[dst] = w[src], b[src]
    ctx.crnn_load_weights(params)
    try:
        def blocks(w):          # coarse random patterns: the argmax changes along the sequence (white noise gives one class)
            return np.kron(rng.uniform(-1, 1, (4, w // 16)), np.ones((16, 16))).astype(np.float32)
        x = torch.from_numpy(np.stack([blocks(160) for _ in range(9)])[..., None]).cuda()
        full = ctx.crnn_forward(x, want_logits=True, want_tokens=True)
        fused = ctx.crnn_forward(x, want_logits=False, want_tokens=True)
        torch.cuda.synchronize()
        rt, rs = octc.ctc_greedy(full['logits'].cpu().numpy(), None, 0)
        for out in (full, fused):
            np.testing.assert_array_equal(out['tokens'].cpu().numpy(), rt)
            np.testing.assert_array_equal(out['scores'].cpu().numpy(), rs)
        widths = np.array([64] * 5 + [128] * 300 + [320] * 2, dtype=np.int32)
        packed = torch.from_numpy(np.concatenate([blocks(int(w)).reshape(-1) for w in widths])).cuda()
        a = ctx.crnn_forward_packed(packed, widths, want_logits=True)
        f = ctx.crnn_forward_packed(packed, widths, want_logits=False)
        ctx.set_option('fuse_ctc', 0)
        u = ctx.crnn_forward_packed(packed, widths, want_logits=False)
        torch.cuda.synchronize()
        ctx.check_errors()
        for out in (f, u):
            np.testing.assert_array_equal(out['tokens'].cpu().numpy(), a['tokens'].cpu().numpy())
            np.testing.assert_array_equal(out['scores'].cpu().numpy(), a['scores'].cpu().numpy())
        assert len(np.unique(a['tokens'].cpu().numpy())) >= 4          # a real mixture of classes and padding
    finally:
        ctx.set_option('fuse_ctc', 1)
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=2))


@pytest.mark.parametrize('kernel', [1, 3])
def test_lstm_handover_timeout_is_an_error_not_output(ctx, kernel):
    """A persistent recurrence whose workgroups cannot hand over (here: one of them withholds its first hand-over
    through the test hook) must fail the call loudly, and the handle must work again afterwards."""
    from ocr_amd._lib import OcrHipError
    params = nets.init_crnn_params(vocab_size=97, seed=2)
    ctx.crnn_load_weights(params)
    x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (40, 64, 128, 1)).astype(np.float32)).cuda()
    good = ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy()
    ctx.check_errors()
    ctx.set_option('lstm_persistent', kernel)          # workgroup-wide passes / independent waves
    ctx.set_option('lstm_debug_drop_handover', 1)
    try:
        ctx.crnn_forward(x, want_logits=True)
        with pytest.raises(OcrHipError, match='lstm hand-over timed out'):
            ctx.check_errors()
        ctx.check_errors()                                   # reported once, then re-armed
        # without an explicit check the next forward call on the handle reports it
        ctx.crnn_forward(x, want_logits=True)
        torch.cuda.synchronize()
        with pytest.raises(OcrHipError, match='lstm hand-over timed out'):
            ctx.crnn_forward(x, want_logits=True)
    finally:
        ctx.set_option('lstm_debug_drop_handover', 0)
        ctx.set_option('lstm_persistent', LSTM_DEFAULT)
    again = ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy()
    ctx.check_errors()
    np.testing.assert_array_equal(again, good)


def test_lstm_timeout_of_one_pipelined_call_fails_every_call_in_flight_and_no_later_one(ctx):
    """Two packed forwards enqueued back to back, the first of which loses a hand-over.  The status of a call travels
    in a slot of its own and the error word is cleared in stream order at the start of every call: the second call's
    kernels must NOT leave early on the first call's word (its output is complete), the check after both must report
    the error (one code cannot say which of the pipelined calls it was: all of them count as failed), and calls made
    after the report are clean."""
    from ocr_amd._lib import OcrHipError
    params = nets.init_crnn_params(vocab_size=97, seed=2)
    ctx.crnn_load_weights(params)
    rng = np.random.default_rng(11)
    widths = np.array([64] * 20 + [128] * 30, dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    good = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
    ctx.check_errors()
    ctx.set_option('lstm_persistent', 1)
    try:
        ctx.set_option('lstm_debug_drop_handover', 1)
        first = ctx.crnn_forward_packed(packed, widths, want_logits=True)        # enqueued; times out on the GPU
        ctx.set_option('lstm_debug_drop_handover', 0)
        try:
            second = ctx.crnn_forward_packed(packed, widths, want_logits=True)   # enqueued right behind it
            entry_reported = False
        except OcrHipError as e:                                                 # the first call's status had already arrived
            assert 'lstm hand-over timed out' in str(e)
            second, entry_reported = None, True
        if not entry_reported:
            with pytest.raises(OcrHipError, match='every recogniser call enqueued before this check'):
                ctx.check_errors()
            # the second call ran on clean error words: complete, and equal to the healthy result
            np.testing.assert_array_equal(second['logits'].cpu().numpy(), good)
        del first                                                                # incomplete: whatever the workspace held
        ctx.check_errors()                                                       # reported once
        third = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
        ctx.check_errors()
        np.testing.assert_array_equal(third, good)
    finally:
        ctx.set_option('lstm_debug_drop_handover', 0)
        ctx.set_option('lstm_persistent', LSTM_DEFAULT)


@pytest.mark.parametrize('shape', [(2, 128, 224, 3), (1, 96, 160, 3), (1, 224, 256, 3)])
def test_direct_3x3_path_matches_oracle_too(ctx, shape):
    """`conv3x3_winograd`: 2 (default) Winograd F(4x4,3x3) where the maps fill its 16x32 tiles and F(2x2,3x3) elsewhere,
    1 F(2x2,3x3) everywhere, 0 the 3x3 layers as a direct implicit GEMM.  All three agree with the oracle to the
    stated tolerance (1e-4 relative) and with each other."""
    params = nets.init_east_params(seed=1)
    ctx.east_load_weights(params)
    rng = np.random.default_rng(77)
    # (128, 224): 128x224 and 64x112 maps take the 4x4 kernel, the second with a half-empty last tile column;
    # (96, 160): 96x160 and 48x80 maps, the second with partial tiles in both directions;
    # (224, 256): maps of 28 and 14 rows end in a half tile (8 rows) with 4 and 6 valid rows, pooled and unpooled
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = nets.east_forward(params, np.stack([oimg.prepare_east_input(i) for i in img])).numpy()
    cp = nets.init_crnn_params(vocab_size=97, seed=2)
    ctx.crnn_load_weights(cp)
    # 3 crops (an odd count) of width 128: the 8-row maps of the last two 3x3 layers are 32 columns wide and go through
    # the 4x4 kernel two images per tile, the third one alone
    x = rng.uniform(-1, 1, (3, 64, 128, 1)).astype(np.float32)
    cref = nets.crnn_forward(cp, x).numpy()
    maps = {}
    try:
        for opt in (0, 1, 2):
            ctx.set_option('conv3x3_winograd', opt)
            maps[opt] = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
            got = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True)['logits'].cpu().numpy()
            _close(got, cref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(cref).max())))
        ctx.set_option('wino4_split', 1)          # deep layers as transform kernel + multiply kernel instead of fused
        split = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
        ctx.set_option('wino4_wave_tiles', 0)     # ... against the fused kernel with the multiply kernel's division of the work
        split_ref = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
        ctx.set_option('wino4_split', 0)
        ctx.set_option('wino4_wave_tiles', 2)
        # the other division of the fused 4x4 kernel's work: 3x3 components per wave on full tiles instead of whole
        # (block, channel) tiles per wave on half tiles
        ctx.set_option('wino4_wave_tiles', 0)
        parts = (ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy(),
                 ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True)['logits'].cpu().numpy())
        # whole tiles per wave, but every layer in workgroups of 64 output channels (the default gives the layers with a
        # multiple of 128 output channels workgroups of 128)
        ctx.set_option('wino4_wave_tiles', 1)
        narrow = (ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy(),
                  ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True)['logits'].cpu().numpy())
    finally:
        ctx.set_option('conv3x3_winograd', 2)
        ctx.set_option('wino4_split', 0)
        ctx.set_option('wino4_wave_tiles', 2)
    _close(split, ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(split_ref, parts[0])          # the multiply kernel: same products in the same order
    np.testing.assert_array_equal(narrow[0], maps[2])
    np.testing.assert_array_equal(narrow[1], got)
    _close(parts[0], ref, rtol=1e-4, atol=1e-5)                  # another order of the inverse transform's additions
    _close(parts[1], cref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(cref).max())))
    assert not np.array_equal(parts[0], maps[2])
    for opt in (0, 1, 2):
        _close(maps[opt], ref, rtol=1e-4, atol=1e-5)
    _close(maps[1], maps[0], rtol=1e-4, atol=1e-5)
    _close(maps[2], maps[0], rtol=1e-4, atol=1e-5)
    assert not np.array_equal(maps[2], maps[1])          # the 4x4 kernel did run
    with pytest.raises(ValueError):
        ctx.set_option('no_such_option', 1)
    with pytest.raises(ValueError):
        ctx.set_option('conv3x3_winograd', 3)


@pytest.mark.parametrize('backbone', ['vgg', 'resnet'])
def test_crops_tiled_side_by_side_match_the_oracle_and_the_image_by_image_launch(ctx, backbone):
    """`wino4_side_by_side` (default 1): the fused F(4x4,3x3) launches tile a batch of crops whose map width is not a
    multiple of 32 columns as one wide image with zero columns between the crops.  Widths 80 / 192 / 320 give maps of 40
    (2x2-pooled layer: two zero columns), 20, 48 and 80 columns; 7, 5 and 3 crops give wide images that end inside a
    tile.  Every crop must match its own oracle forward (so no crop sees its neighbours beyond rounding), and the
    image-by-image launch to the same tolerance; the two are different computations."""
    V = 97
    params = nets.init_crnn_params(vocab_size=V, seed=21, backbone=backbone)
    ctx.crnn_load_weights(params)
    try:
        rng = np.random.default_rng(31)
        widths = np.array([80] * 3 + [192] * 7 + [320] * 5, dtype=np.int32)
        crops = [rng.uniform(-1, 1, (64, int(w))).astype(np.float32) for w in widths]
        packed = torch.from_numpy(np.concatenate([c.reshape(-1) for c in crops])).cuda()
        side = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
        ctx.set_option('wino4_side_by_side', 0)
        single = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
        ctx.check_errors()
        T = ctx.seq_len(widths)
        row0 = np.concatenate([[0], np.cumsum(T)])
        for i, c in enumerate(crops):
            ref = nets.crnn_forward(params, c[None, :, :, None])[0].numpy()
            tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
            _close(side[row0[i]:row0[i + 1]], ref, rtol=1e-4, atol=tol)
            _close(single[row0[i]:row0[i + 1]], ref, rtol=1e-4, atol=tol)
        assert not np.array_equal(side, single)          # the side-by-side launch did run
    finally:
        ctx.set_option('wino4_side_by_side', 1)
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=V, seed=2))


def test_lstm_per_step_launches_match_the_persistent_layer(ctx):
    """`lstm_persistent = 0` (one launch per time step) against the default single persistent launch per layer:
    same cell arithmetic, different k order inside the matrix cores."""
    V = 97
    params = nets.init_crnn_params(vocab_size=V, seed=11)
    ctx.crnn_load_weights(params)
    rng = np.random.default_rng(5)
    widths = np.array([64] * 5 + [96] * 40 + [128] * 150 + [256] * 3 + [512], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    a = ctx.crnn_forward_packed(packed, widths, want_logits=True)
    ctx.set_option('lstm_persistent', 0)
    try:
        b = ctx.crnn_forward_packed(packed, widths, want_logits=True)
    finally:
        ctx.set_option('lstm_persistent', LSTM_DEFAULT)
    torch.cuda.synchronize()
    la, lb = a['logits'].cpu().numpy(), b['logits'].cpu().numpy()
    _close(la, lb, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(lb).max())))
    np.testing.assert_array_equal(a['lengths'].cpu().numpy(), b['lengths'].cpu().numpy())


def test_lstm_independent_waves_equal_the_pass_kernel_bit_for_bit(ctx):
    """`lstm_persistent = 2 / 3`: waves own blocks of 16 sequences with all gate columns, h straight from memory, one
    hand-over counter per block, no barrier - against `lstm_persistent = 1` (workgroup-wide passes of 112 sequences
    through LDS).  Same k order in every accumulator and the same cell arithmetic: identical logits.  Ragged batch with
    partial blocks (5, 40, 150, 3, 1 sequences per length) and a uniform batch whose count is not a multiple of 16."""
    V = 97
    params = nets.init_crnn_params(vocab_size=V, seed=11)
    ctx.crnn_load_weights(params)
    rng = np.random.default_rng(6)
    widths = np.array([64] * 5 + [96] * 40 + [128] * 150 + [256] * 3 + [512], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.uniform(-1, 1, (37, 64, 160, 1)).astype(np.float32)).cuda()
    out = {}
    try:
        for opt in (1, 3):          # 3: independent waves whatever the batch size (2 takes them from 8 blocks per workgroup on)
            ctx.set_option('lstm_persistent', opt)
            out[opt] = (ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy(),
                        ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy())
            ctx.check_errors()
        # a batch large enough for the default to pick the wave kernel by itself: 16 groups x 9 blocks
        big = torch.from_numpy(rng.uniform(-1, 1, (2200, 64, 64, 1)).astype(np.float32)).cuda()
        for opt in (1, 2):
            ctx.set_option('lstm_persistent', opt)
            out['big', opt] = ctx.crnn_forward(big, want_logits=True)['logits'].cpu().numpy()
            ctx.check_errors()
    finally:
        ctx.set_option('lstm_persistent', LSTM_DEFAULT)
    np.testing.assert_array_equal(out[3][0], out[1][0])
    np.testing.assert_array_equal(out[3][1], out[1][1])
    np.testing.assert_array_equal(out['big', 2], out['big', 1])
    ref = nets.crnn_forward(params, x.cpu().numpy()[:3]).numpy()
    _close(out[3][1][:3], ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))


def test_flat_gemm_matches_the_convolution_kernel(ctx):
    """`flat_gemm = 0` runs the Dense layers, the LSTM input projections and the 2x2 convolutions on the implicit-GEMM
    convolution kernel instead of the 256 x 128 flat GEMM.  Both are float32 fma chains; the 2x2 layers sum taps and
    channels in a different order (tap-major here, channel-chunk-major there), so the logits agree to float32
    rounding, not bit for bit (VGG and ResNet extractors, ragged widths)."""
    rng = np.random.default_rng(9)
    widths = np.array([64] * 3 + [96] * 7 + [128] * 5 + [320], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    try:
        for backbone in ('vgg', 'resnet'):
            ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=13, backbone=backbone))
            a = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
            ctx.set_option('flat_gemm', 0)
            try:
                b = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
            finally:
                ctx.set_option('flat_gemm', 1)
            assert np.isfinite(a).all()
            _close(a, b, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(b).max())))
    finally:
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=2))


def test_mean_over_feature_rows_in_the_gemm_epilogue_is_bit_identical(ctx):
    """`fuse_mean_rows = 1` (default): the extractor's last 2x2 convolution sums its 3 output rows in its epilogue, in row
    order, and divides by 3 - the arithmetic of the separate kernel behind `fuse_mean_rows = 0`, on the same convolution
    sums: identical features and logits (VGG and ResNet extractors, uniform and ragged batches, a batch whose positions
    do not fill the last 85-position tile)."""
    rng = np.random.default_rng(21)
    widths = np.array([64] * 3 + [96] * 7 + [128] * 5 + [320], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.uniform(-1, 1, (7, 64, 192, 1)).astype(np.float32)).cuda()
    try:
        for backbone in ('vgg', 'resnet'):
            ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=13, backbone=backbone))
            a = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
            au = ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy()
            fa = ctx.crnn_activation('features')
            ctx.set_option('fuse_mean_rows', 0)
            try:
                b = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
                bu = ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy()
                fb = ctx.crnn_activation('features')
            finally:
                ctx.set_option('fuse_mean_rows', 1)
            assert np.isfinite(a).all() and np.abs(a).max() > 0
            np.testing.assert_array_equal(fa, fb)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(au, bu)
    finally:
        ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=2))


def test_width_groups_on_side_streams_give_identical_results(ctx):
    """Many small groups of equal-width crops (lines of grouped text) run their feature extractors on the caller's stream
    and three side streams with buffers of their own (`crnn_side_streams = 1`, default); everything on one stream
    (`= 0`) must give the same bits, in both precisions, also when calls follow each other without a host wait."""
    from ocr_amd._lib import BF16, F32
    rng = np.random.default_rng(33)
    widths = np.sort(rng.choice(np.arange(64, 1500, 64), 40)).astype(np.int32)          # ~20 distinct widths, 1-4 crops each
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    ctx.crnn_load_weights(nets.init_crnn_params(vocab_size=97, seed=2))
    for dt in (F32, BF16):
        outs = [ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=dt)['logits'] for _ in range(3)]
        torch.cuda.synchronize()
        ctx.check_errors()
        ctx.set_option('crnn_side_streams', 0)
        try:
            ref = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=dt)['logits'].cpu().numpy()
        finally:
            ctx.set_option('crnn_side_streams', 1)
        assert np.isfinite(ref).all() and np.abs(ref).max() > 0
        for o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), ref)


def test_bf16_variant_within_1e2_of_the_float32_oracle(ctx):
    """`compute_dtype = BF16`: bfloat16 operands, float32 accumulation.  Tolerance: relative error of the whole tensor
    ||got - ref|| / ||ref|| <= 1e-2 against the float32 oracle, and no single element further than 5e-2 of the tensor's
    scale (20 layers of 2^-9 operand rounding under random weights leave a few 1e-2 outliers); the float32 path is held
    to 1e-4 element-wise by the other tests."""
    def rel(a, b):
        return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    from ocr_amd._lib import BF16
    params = nets.init_east_params(seed=1)
    ctx.east_load_weights(params)
    rng = np.random.default_rng(123)
    img = rng.integers(0, 256, (2, 96, 160, 3), dtype=np.uint8)
    ref = nets.east_forward(params, np.stack([oimg.prepare_east_input(i) for i in img])).numpy()
    got = ctx.east_forward(torch.from_numpy(img).cuda(), compute_dtype=BF16).cpu().numpy()
    f32 = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
    assert rel(got, ref) <= 1e-2 and np.abs(got - ref).max() <= 5e-2, (rel(got, ref), np.abs(got - ref).max())
    assert np.abs(got - f32).max() > 0                                            # it really is another arithmetic
    cp = nets.init_crnn_params(vocab_size=97, seed=2)
    ctx.crnn_load_weights(cp)
    x = rng.uniform(-1, 1, (4, 64, 160, 1)).astype(np.float32)
    cref = nets.crnn_forward(cp, x).numpy()
    cgot = ctx.crnn_forward(torch.from_numpy(x).cuda(), want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
    scale = float(np.abs(cref).max())
    assert rel(cgot, cref) <= 1e-2 and np.abs(cgot - cref).max() <= 5e-2 * max(1.0, scale), (rel(cgot, cref), np.abs(cgot - cref).max(), scale)
    widths = np.array([64, 64, 128, 160], dtype=np.int32)
    packed = torch.from_numpy(rng.uniform(-1, 1, int(64 * widths.sum())).astype(np.float32)).cuda()
    a = ctx.crnn_forward_packed(packed, widths, want_logits=True, compute_dtype=BF16)['logits'].cpu().numpy()
    b = ctx.crnn_forward_packed(packed, widths, want_logits=True)['logits'].cpu().numpy()
    assert rel(a, b) <= 1e-2, rel(a, b)


def test_first_layer_on_the_matrix_cores_equals_the_vector_kernel(ctx):
    """The detector's 3 -> 64 layer runs on the matrix cores by default; `first_layer_mfma = 0` runs it on the vector
    ALU.  Both sum the same 27 exact products tap by tap (the matrix instruction two per step); on gfx950 everything
    downstream is bit-identical, which this test pins (it is not guaranteed by construction) - on images large
    enough that several workgroups share a compute unit (more than 256 tiles of 8 x 64 pixels), uint8 and float input,
    widths that end in a partial tile."""
    params = nets.init_east_params(seed=1)
    ctx.east_load_weights(params)
    rng = np.random.default_rng(5)
    cases = [rng.integers(0, 256, (2, 512, 576, 3), dtype=np.uint8), rng.integers(0, 256, (1, 1024, 1024, 3), dtype=np.uint8),
             rng.normal(0, 1, (1, 544, 480, 3)).astype(np.float32), rng.integers(0, 256, (3, 96, 160, 3), dtype=np.uint8),
             # large values of mixed sign per tap: a different order of the 27 additions would show as cancellation error
             (rng.choice([-1e4, 1e4, 3.3e3, -7.7e3, 1e-3], (2, 96, 160, 3)) * rng.uniform(0.5, 1.0, (2, 96, 160, 3))).astype(np.float32)]
    try:
        for img in cases:
            got = {}
            for opt in (1, 0, 1):
                ctx.set_option('first_layer_mfma', opt)
                maps = ctx.east_forward(torch.from_numpy(img).cuda()).cpu().numpy()
                got.setdefault(opt, []).append((ctx.east_activation('pool1'), maps))
            for pool1, maps in got[1]:
                np.testing.assert_array_equal(pool1, got[0][0][0])
                np.testing.assert_array_equal(maps, got[0][0][1])
        # the recogniser's pooled 1 -> 32 first layer likewise: widths with full and partial tiles of 64 pooled columns
        cp = nets.init_crnn_params(vocab_size=97, seed=2)
        ctx.crnn_load_weights(cp)
        for n, wpad in ((300, 64), (40, 192), (17, 320), (3, 1040)):
            x = torch.from_numpy(rng.uniform(-1, 1, (n, 64, wpad, 1)).astype(np.float32)).cuda()
            got = {}
            for opt in (1, 0):
                ctx.set_option('first_layer_mfma', opt)
                got[opt] = ctx.crnn_forward(x, want_logits=True)['logits'].cpu().numpy()
            np.testing.assert_array_equal(got[1], got[0])
    finally:
        ctx.set_option('first_layer_mfma', 1)
