"""ocrhip_east_decode_static (staging area in a fixed device array, no allocation in the call) against
ocrhip_east_decode through the C ABI: identical bits, on two calls in a row, the second on a smaller input."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _decode(ctx, fn, d_maps, in_hw, cap):
    """Calls one of the two decode entry points on caller-allocated outputs (pre-filled, so stale slots compare equal)."""
    B, mh, mw, _ = d_maps.shape
    dev = d_maps.device
    out = {
        'xyxy': torch.full((B, cap, 4), -7.0, dtype=torch.float32, device=dev),
        'polys': torch.full((B, cap, 8), -7.0, dtype=torch.float32, device=dev),
        'scores': torch.full((B, cap), -7.0, dtype=torch.float32, device=dev),
        'pix': torch.full((B, cap), -7, dtype=torch.int32, device=dev),
        'counts': torch.full((B,), -7, dtype=torch.int32, device=dev),
    }
    torch.cuda.synchronize()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = fn(ctx._h, p(d_maps), B, mh, mw, int(in_hw[0]), int(in_hw[1]), 0.35, cap, p(out['xyxy']), p(out['polys']),
            p(out['scores']), p(out['pix']), p(out['counts']), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, {k: v.cpu().numpy() for k, v in out.items()}


def test_static_decode_equals_decode_on_two_calls_in_a_row():
    from ocr_amd import synth
    from ocr_amd.engine import HipContext
    ctx = HipContext(0)
    try:
        lib = ctx.lib
        _, big, _ = synth.make_pages(41, 3, 512, 768, 90)         # 3 maps of 128 x 192
        _, small, _ = synth.make_pages(42, 1, 128, 160, 6)        # 1 map of 32 x 40 (a ragged last chunk of 256 pixels)
        for maps, hw in ((big, (512, 768)), (small, (128, 160))):
            d = torch.from_numpy(maps).cuda()
            cap = maps.shape[1] * maps.shape[2]
            rc0, ref = _decode(ctx, lib.ocrhip_east_decode, d, hw, cap)
            rc1, got = _decode(ctx, lib.ocrhip_east_decode_static, d, hw, cap)
            assert rc0 == 0 and rc1 == 0, lib.ocrhip_last_error()
            assert int(ref['counts'].sum()) > 0
            for k in ref:
                np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
        # beyond the fixed staging area: a clear error, nothing launched
        huge = torch.zeros((2, 2048, 2304, 6), dtype=torch.float32, device='cuda')       # 2 * 18432 chunks > 32768
        rc, out = _decode(ctx, lib.ocrhip_east_decode_static, huge, (8192, 9216), 16)
        assert rc == -2 and b'static staging area' in lib.ocrhip_last_error()
        assert int(out['counts'][0]) == -7
    finally:
        ctx.close()


def test_static_decode_takes_no_device_memory():
    """The property the entry point exists for: on a FRESH handle (its grow-only staging buffer has never been grown)
    the first call, on a map larger than any earlier one of the process, leaves the device's free memory where it was.
    Outputs are allocated by the caller beforehand; `torch.cuda.mem_get_info` reads the driver's figure, which sees the
    library's hipMalloc calls as well as torch's."""
    from ocr_amd import synth
    from ocr_amd.engine import HipContext
    _, maps, _ = synth.make_pages(43, 6, 1024, 1024, 150)           # 6 maps of 256 x 256: 1536 chunks of staging
    d = torch.from_numpy(maps).cuda()
    B, mh, mw, _ = d.shape
    cap = mh * mw
    outs = [torch.empty((B, cap, 4), dtype=torch.float32, device='cuda'), torch.empty((B, cap, 8), dtype=torch.float32, device='cuda'),
            torch.empty((B, cap), dtype=torch.float32, device='cuda'), torch.empty((B, cap), dtype=torch.int32, device='cuda'),
            torch.empty((B,), dtype=torch.int32, device='cuda')]
    ctx = HipContext(0)
    try:
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        rc = ctx.lib.ocrhip_east_decode_static(ctx._h, p(d), B, mh, mw, 1024, 1024, 0.35, cap, *[p(t) for t in outs], stream)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert rc == 0, ctx.lib.ocrhip_last_error()
        assert free1 == free0, 'the static entry point changed the free device memory by %d bytes' % (free0 - free1)
        assert int(outs[4].sum()) > 0
    finally:
        ctx.close()
