"""Thin host layer over the C ABI: torch tensors in, torch tensors out.

PyTorch is used for device memory and streams only; every computation below happens inside
libocrhip.so.  One `HipContext` per (device, host thread).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import F32, BF16, U8, MERGE_UNION, MERGE_AVERAGE, MERGE_WEIGHTED, TensorDesc, check

_MERGE = {'union': MERGE_UNION, 'average': MERGE_AVERAGE, 'weighted': MERGE_WEIGHTED}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _uniform_stride(t):
    """Element stride of a tensor whose elements lie at base + k * stride, k = 0..numel-1 in row-major order (a
    contiguous tensor: 1; `m[..., 0]` of a contiguous (h,w,6) map: 6); None when there is no such stride."""
    if t.numel() <= 1:
        return 1
    dims = [(n, st) for n, st in zip(t.shape, t.stride()) if n > 1]
    step = dims[-1][1]
    expect = step
    for n, st in reversed(dims):
        if st != expect:
            return None
        expect = st * n
    return step if step >= 1 else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _descs(state_dict):
    """state dict (torch tensors or arrays) -> (ctypes array of TensorDesc, keep-alive list)."""
    keep, items = [], []
    for name, value in state_dict.items():
        if name.startswith('module.'):
            name = name[len('module.'):]
        if name.endswith('num_batches_tracked'):
            continue
        arr = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if arr.ndim > 4:
            continue
        bname = name.encode('utf-8')
        keep.append((arr, bname))
        d = TensorDesc()
        d.name = bname
        d.data = arr.ctypes.data
        d.ndim = arr.ndim
        for i in range(arr.ndim):
            d.shape[i] = arr.shape[i]
        items.append(d)
    array = (TensorDesc * len(items))(*items)
    return array, keep


class HipContext:
    """Owns one ocrhip handle (weights + workspaces) on one GPU."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError('ocr_amd needs a ROCm GPU (torch.cuda.is_available() is False)')
        self.lib = _lib.load()
        self.device = torch.device('cuda', device)
        h = ctypes.c_void_p()
        check(self.lib.ocrhip_create(int(device), ctypes.byref(h)))
        self._h = h
        self.east_loaded = False
        self.crnn_loaded = False
        self.options = {}              # what set_option was last called with, per name

    def close(self):
        if getattr(self, '_h', None):
            self.lib.ocrhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ stage timers
    def profile_enable(self, enable=True):
        check(self.lib.ocrhip_profile_enable(self._h, 1 if enable else 0))

    def profile_only(self, name=None):
        """Times only the launches recorded under `name` (None: all of them again)."""
        check(self.lib.ocrhip_profile_only(self._h, name.encode() if name else None))

    def profile_collect(self, cap=64):
        """-> list of dict(name, launches, total_ms, flops, bytes) sorted by total time (synchronises)."""
        arr = (_lib.KernelStat * cap)()
        n = ctypes.c_int(0)
        check(self.lib.ocrhip_profile_collect(self._h, arr, cap, ctypes.byref(n)))
        return [{'name': arr[i].name.decode(), 'launches': arr[i].launches, 'total_ms': arr[i].total_ms,
                 'flops': arr[i].flops, 'bytes': arr[i].bytes} for i in range(min(cap, n.value))]

    # ------------------------------------------------------------------ EAST
    def east_load_weights(self, state_dict):
        arr, keep = _descs(state_dict)
        check(self.lib.ocrhip_east_load_weights(self._h, arr, len(arr)))
        self.east_loaded = True

    def set_option(self, name, value):
        """'conv3x3_winograd': 2 (default) Winograd F(4x4,3x3) where the map fills its tiles and F(2x2,3x3) elsewhere,
        1 F(2x2,3x3) for all wide 3x3 convolutions, 0 direct implicit GEMM.
        'lstm_persistent': 2 (default) one persistent launch per BiLSTM layer (independent waves on blocks of 16
        sequences for large batches, workgroup-wide passes otherwise), 1 passes always, 3 independent waves always,
        0 one launch per time step.
        'lstm16_units': 64 (default) or 32 hidden units per workgroup of the bfloat16 recurrence.
        'crnn_side_streams': 1 (default) many small width groups of a packed batch run on side streams, 0 on one stream.
        'fuse_mean_rows': 1 (default) the mean over the feature rows in the last convolution's epilogue, 0 a kernel of its own.
        'bf16_fuse_first': 1 (default) the bfloat16 detector's first two layers in one launch, 0 two launches.
        'wino4_split': 0 (default) every F(4x4,3x3) layer fused, 1 layers with >= 512 input channels as transform kernel +
        multiply kernel.
        'flat_gemm': 1 (default) Dense layers / LSTM input projections / 2x2 convolutions on the 256 x 128 flat GEMM,
        0 on the implicit-GEMM convolution kernel."""
        check(self.lib.ocrhip_set_option(self._h, name.encode(), int(value)))
        self.options[name] = int(value)

    def image_resize_norm(self, images, out_h, out_w):
        """images: (B,H,W,3) uint8 raw or float32 in [0,1], on the GPU -> (B,out_h,out_w,3) float32 normalised
        network input (`/255` for uint8, antialiased bilinear stretch, `(x-0.5)/0.5`)."""
        assert images.is_cuda and images.dim() == 4 and images.shape[3] == 3 and images.is_contiguous()
        if images.dtype == torch.uint8:
            in_dtype = U8
        elif images.dtype == torch.float32:
            in_dtype = F32
        else:
            raise ValueError('image_resize_norm: images must be uint8 or float32')
        B, H, W, _ = images.shape
        out = torch.empty((B, int(out_h), int(out_w), 3), dtype=torch.float32, device=images.device)
        check(self.lib.ocrhip_image_resize_norm(self._h, _ptr(images), in_dtype, B, H, W, int(out_h), int(out_w),
                                                _ptr(out), _stream()))
        return out

    def dice_coeff(self, x, y):
        """Dice coefficient 2*sum(x*y) / (sum(x)+sum(y)) of two float32 GPU tensors of one shape (0 when both are all
        zero) -> float32 GPU scalar.  Views that step over whole elements (`maps[..., 0]` of a contiguous map) are
        walked in place."""
        assert x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32
        if tuple(x.shape) != tuple(y.shape):
            raise ValueError('dice_coeff: shapes differ: {} vs {}'.format(tuple(x.shape), tuple(y.shape)))
        stride = _uniform_stride(x)
        if stride is None or _uniform_stride(y) != stride:
            x, y, stride = x.contiguous(), y.contiguous(), 1
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
        check(self.lib.ocrhip_dice_coeff(self._h, _ptr(x), _ptr(y), x.numel(), stride, _ptr(out), _stream()))
        return out[0]

    def east_forward(self, images, compute_dtype=F32):
        """images: (B,H,W,3) uint8 raw or float32 normalised, on the GPU -> (B,H/4,W/4,6) float32."""
        assert images.is_cuda and images.dim() == 4 and images.shape[3] == 3 and images.is_contiguous()
        if images.dtype == torch.uint8:
            in_dtype = U8
        elif images.dtype == torch.float32:
            in_dtype = F32
        else:
            raise ValueError('east_forward: images must be uint8 or float32')
        B, H, W, _ = images.shape
        maps = torch.empty((B, H // 4, W // 4, 6), dtype=torch.float32, device=images.device)
        check(self.lib.ocrhip_east_forward(self._h, _ptr(images), B, H, W, in_dtype, compute_dtype, _ptr(maps), _stream()))
        return maps

    def _activation(self, fn, name):
        shape = (ctypes.c_int64 * 4)()
        check(fn(self._h, name.encode(), None, 0, shape))
        out = np.empty(tuple(int(s) for s in shape), dtype=np.float32)
        check(fn(self._h, name.encode(), out.ctypes.data, out.size, shape))
        return out

    def east_activation(self, name):
        return self._activation(self.lib.ocrhip_east_get_activation, name)

    def east_decode(self, maps, input_hw, threshold=0.35, cap=None, with_polys=False, with_pix=False):
        """maps (B,mh,mw,6) -> dict(xyxy (B,cap,4), scores (B,cap), counts (B,), [polys (B,cap,4,2)], [pix])."""
        assert maps.is_cuda and maps.dtype == torch.float32 and maps.is_contiguous()
        B, mh, mw, _ = maps.shape
        if cap is None:
            cap = mh * mw
        dev = maps.device
        out = {
            'xyxy': torch.empty((B, cap, 4), dtype=torch.float32, device=dev),
            'scores': torch.empty((B, cap), dtype=torch.float32, device=dev),
            'counts': torch.empty((B,), dtype=torch.int32, device=dev),
        }
        polys = torch.empty((B, cap, 4, 2), dtype=torch.float32, device=dev) if with_polys else None
        pix = torch.empty((B, cap), dtype=torch.int32, device=dev) if with_pix else None
        check(self.lib.ocrhip_east_decode(self._h, _ptr(maps), B, mh, mw, int(input_hw[0]), int(input_hw[1]),
                                          float(threshold), int(cap), _ptr(out['xyxy']), _ptr(polys),
                                          _ptr(out['scores']), _ptr(pix), _ptr(out['counts']), _stream()))
        if with_polys:
            out['polys'] = polys
        if with_pix:
            out['pix'] = pix
        return out

    def lanms(self, xyxy, scores, counts, merge_threshold=0.3, nms_threshold=0.25, merge_method='union',
              max_iter=0, max_out=0, full=False):
        """xyxy (B,cap,4), scores (B,cap) or None, counts (B,) int32 ->
        dict(boxes (B,cap,4) rank order, scores, src, counts, n_iter, [merged, keep in input order])."""
        assert xyxy.is_cuda and xyxy.dtype == torch.float32 and xyxy.is_contiguous()
        B, cap, _ = xyxy.shape
        dev = xyxy.device
        out = {
            'boxes': torch.empty((B, cap, 4), dtype=torch.float32, device=dev),
            'scores': torch.empty((B, cap), dtype=torch.float32, device=dev),
            'src': torch.empty((B, cap), dtype=torch.int32, device=dev),
            'counts': torch.empty((B,), dtype=torch.int32, device=dev),
            'n_iter': torch.empty((B,), dtype=torch.int32, device=dev),
        }
        merged = torch.empty((B, cap, 4), dtype=torch.float32, device=dev) if full else None
        keep = torch.empty((B, cap), dtype=torch.uint8, device=dev) if full else None
        check(self.lib.ocrhip_lanms(self._h, _ptr(xyxy), _ptr(scores), _ptr(counts), B, cap, float(merge_threshold),
                                    float(nms_threshold), _MERGE[merge_method], int(max_iter or 0), int(max_out or 0),
                                    _ptr(merged), _ptr(keep), _ptr(out['boxes']), _ptr(out['scores']), _ptr(out['src']),
                                    _ptr(out['counts']), _ptr(out['n_iter']), _stream()))
        if full:
            out['merged'], out['keep'] = merged, keep
        return out

    # ------------------------------------------------------------------ crop / pack
    def crop_resize_pack(self, images, rects, box_img, inter_hw, out_w, dst_off, total_elems, out_h=64,
                         out_dtype=torch.float32):
        """images (B,H,W,3) uint8; rects (n,4) int32; box_img (n,) int32; inter_hw (n,2) int32; out_w (n,) int32;
        dst_off (n,) int64 -> flat packed buffer of `total_elems` elements."""
        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        B, H, W, _ = images.shape
        n = int(rects.shape[0])
        packed = torch.empty((int(total_elems),), dtype=out_dtype, device=images.device)
        code = F32 if out_dtype == torch.float32 else BF16
        check(self.lib.ocrhip_crop_resize_pack(self._h, _ptr(images), B, H, W, _ptr(rects), _ptr(box_img), _ptr(inter_hw),
                                               _ptr(out_w), _ptr(dst_off), n, int(out_h), _ptr(packed), code, _stream()))
        return packed

    def crop_rotated_pack(self, images, geom, box_img, inter_hw, out_w, dst_off, total_elems, out_h=64,
                          out_dtype=torch.float32):
        """`crop_resize_pack` for oriented rectangles: geom (n,8) float32 = origin x, y, step along the text (x, y),
        step across the text (x, y), crop width, crop height (see `utils.image_sizes.plan_crops_rotated`)."""
        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        assert geom.dtype == torch.float32 and geom.is_contiguous() and geom.shape[1] == 8
        B, H, W, _ = images.shape
        n = int(geom.shape[0])
        packed = torch.empty((int(total_elems),), dtype=out_dtype, device=images.device)
        code = F32 if out_dtype == torch.float32 else BF16
        check(self.lib.ocrhip_crop_rotated_pack(self._h, _ptr(images), B, H, W, _ptr(geom), _ptr(box_img), _ptr(inter_hw),
                                                _ptr(out_w), _ptr(dst_off), n, int(out_h), _ptr(packed), code, _stream()))
        return packed

    def rotated_nms(self, polys, scores, counts, iou_threshold, max_out=2048):
        """Greedy NMS with rotated IoU on decoded quadrilaterals: polys (B,cap,4,2) or (B,cap,8) float32, scores
        (B,cap), counts (B,) int32, all on the GPU -> dict(polys (B,max_out,4,2), scores, index (B,max_out) int32,
        counts (B,) int32), kept quadrilaterals in descending score order."""
        assert polys.is_cuda and polys.dtype == torch.float32 and polys.is_contiguous()
        assert scores.dtype == torch.float32 and scores.is_contiguous() and counts.dtype == torch.int32
        B, cap = int(scores.shape[0]), int(scores.shape[1])
        assert polys.numel() == B * cap * 8
        dev = polys.device
        out = {'polys': torch.zeros((B, int(max_out), 4, 2), dtype=torch.float32, device=dev),
               'scores': torch.zeros((B, int(max_out)), dtype=torch.float32, device=dev),
               'index': torch.full((B, int(max_out)), -1, dtype=torch.int32, device=dev),
               'counts': torch.zeros((B,), dtype=torch.int32, device=dev)}
        check(self.lib.ocrhip_rotated_nms(self._h, _ptr(polys), _ptr(scores), _ptr(counts), B, cap, float(iou_threshold),
                                          int(max_out), _ptr(out['polys']), _ptr(out['scores']), _ptr(out['index']),
                                          _ptr(out['counts']), _stream()))
        return out

    # ------------------------------------------------------------------ CRNN
    def crnn_load_weights(self, state_dict):
        arr, keep = _descs(state_dict)
        check(self.lib.ocrhip_crnn_load_weights(self._h, arr, len(arr)))
        self.crnn_loaded = True

    @property
    def vocab_size(self):
        return int(self.lib.ocrhip_crnn_vocab_size(self._h))

    def seq_len(self, wpad):
        """Time steps of the loaded recogniser for padded crop width(s) `wpad` (VGG: w/4 - 1, ResNet: w/4 + 1)."""
        delta = int(self.lib.ocrhip_crnn_seq_len(self._h, 8)) - 2
        if delta not in (-1, 1):
            raise RuntimeError('seq_len: recogniser weights are not loaded')
        return np.asarray(wpad) // 4 + delta

    def crnn_forward(self, crops, want_logits=True, want_tokens=False, compute_dtype=F32):
        """crops (n,64,Wpad,1) or (n,64,Wpad) float32 -> dict(logits (n,T,V), tokens (n,T), scores (n,))."""
        assert crops.is_cuda and crops.dtype == torch.float32 and crops.is_contiguous()
        n, hh, Wpad = int(crops.shape[0]), int(crops.shape[1]), int(crops.shape[2])
        if hh != 64:
            raise ValueError('crnn_forward: crops must be 64 pixels high')
        T, V = int(self.seq_len(Wpad)), self.vocab_size
        dev = crops.device
        logits = torch.empty((n, T, V), dtype=torch.float32, device=dev) if want_logits else None
        tokens = torch.empty((n, T), dtype=torch.int32, device=dev) if want_tokens else None
        scores = torch.empty((n,), dtype=torch.float32, device=dev) if want_tokens else None
        check(self.lib.ocrhip_crnn_forward(self._h, _ptr(crops), n, Wpad, compute_dtype, _ptr(logits), _ptr(tokens),
                                           _ptr(scores), _stream()))
        return {'logits': logits, 'tokens': tokens, 'scores': scores}

    def crnn_forward_packed(self, packed, out_w, want_logits=False, compute_dtype=F32):
        """packed: flat float32 GPU buffer of (64, out_w[i]) planes back to back; out_w: host int32 array,
        non-decreasing -> dict(tokens (n,Tmax) int32, scores (n,), lengths (n,), [logits (sum T, V)])."""
        assert packed.is_cuda and packed.dtype == torch.float32 and packed.is_contiguous()
        out_w = np.ascontiguousarray(out_w, dtype=np.int32)
        n = int(out_w.shape[0])
        if int(packed.numel()) < 64 * int(out_w.astype(np.int64).sum()):
            raise ValueError('crnn_forward_packed: packed buffer is smaller than the crops it should hold')
        t_all = self.seq_len(out_w)
        tmax, rows = int(t_all.max()), int(t_all.astype(np.int64).sum())
        dev = packed.device
        tokens = torch.empty((n, tmax), dtype=torch.int32, device=dev)
        scores = torch.empty((n,), dtype=torch.float32, device=dev)
        lengths = torch.empty((n,), dtype=torch.int32, device=dev)
        logits = torch.empty((rows, self.vocab_size), dtype=torch.float32, device=dev) if want_logits else None
        check(self.lib.ocrhip_crnn_forward_packed(self._h, _ptr(packed), ctypes.c_void_p(out_w.ctypes.data), n,
                                                  compute_dtype, tmax, _ptr(logits), _ptr(tokens), _ptr(scores),
                                                  _ptr(lengths), _stream()))
        return {'tokens': tokens, 'scores': scores, 'lengths': lengths, 'logits': logits}

    def check_errors(self, synchronize=True):
        """Raises OcrHipError when a persistent BiLSTM launch of an earlier call timed out (its outputs are invalid).
        synchronize=False: the caller has already waited for the outputs it is about to use."""
        check(self.lib.ocrhip_check_errors(self._h, _stream(), 1 if synchronize else 0))

    def crnn_activation(self, name):
        return self._activation(self.lib.ocrhip_crnn_get_activation, name)

    MAX_GPU_BEAMS = 16

    def ctc_beam_search(self, logits, lengths=None, row0=None, t_stride=None, num_beams=10, blank=0):
        """CTC prefix beam search on the GPU.  logits (B,T,V) float32 padded [+ lengths (B,) int32], or a flat
        (sum T, V) matrix with lengths and row0 (B,) int32 (first row of each sequence) and t_stride >= max length
        -> (tokens (B,T) int32 padded with -1, scores (B,) float32 = log p / length)."""
        assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
        if row0 is None:
            B, T, V = logits.shape
        else:
            if lengths is None or t_stride is None:
                raise ValueError('ctc_beam_search: a flat logits matrix needs lengths, row0 and t_stride')
            B, T, V = int(lengths.shape[0]), int(t_stride), int(logits.shape[1])
        tokens = torch.empty((B, T), dtype=torch.int32, device=logits.device)
        scores = torch.empty((B,), dtype=torch.float32, device=logits.device)
        check(self.lib.ocrhip_ctc_beam_search(self._h, _ptr(logits), B, T, V, _ptr(lengths), _ptr(row0), int(blank),
                                              int(num_beams), _ptr(tokens), _ptr(scores), _stream()))
        return tokens, scores

    def ctc_greedy(self, logits, lengths=None, blank=0):
        assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
        B, T, V = logits.shape
        tokens = torch.empty((B, T), dtype=torch.int32, device=logits.device)
        scores = torch.empty((B,), dtype=torch.float32, device=logits.device)
        check(self.lib.ocrhip_ctc_greedy(self._h, _ptr(logits), B, T, V, _ptr(lengths), int(blank), _ptr(tokens),
                                         _ptr(scores), _stream()))
        return tokens, scores
