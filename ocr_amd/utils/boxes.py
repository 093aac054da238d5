"""Box utilities with the reference's names and signatures, backed by the HIP kernels.

`nms` mirrors utils/image/bounding_box/non_max_suppression.py:25-78: a dispatcher over a method
registry (`nms.dispatch('name')`, utils/wrappers.py:109-137) whose methods take
`(boxes (N,4) xyxy, scores (N,)|None, max_output_size, nms_threshold, **kw)` and return
`(boxes, scores, valid_mask)`.
"""
import numpy as np
import torch

_nms_methods = {}
_default_ctx = [None]


def default_context(device=0):
    from ..engine import HipContext
    if _default_ctx[0] is None:
        _default_ctx[0] = HipContext(device)
    return _default_ctx[0]


def dezoom_box(x, y, w, h, factor, image_h=None, image_w=None):
    """Scales boxes about their centres by `factor` and clips them to the image
    (utils/image/bounding_box/converter.py:202-232, unrotated boxes).  Float coordinates are relative (clipped to
    [0, 1]); integer coordinates are pixels: the arithmetic runs in float32, the new corner and size are truncated
    back to the integer type, then the size is clipped to `image_h` / `image_w`."""
    if factor == 1.:
        return x, y, w, h
    dtype = x.dtype
    relative = dtype.kind == 'f'
    if not relative:
        if image_h is None or image_w is None:
            raise ValueError('dezoom of pixel boxes needs image_h and image_w')
        x, y, w, h = [c.astype(np.float32) for c in (x, y, w, h)]
    new_h, new_w = h * factor, w * factor
    new_x = np.maximum((x + w / 2.) - new_w / 2., 0)
    new_y = np.maximum((y + h / 2.) - new_h / 2., 0)
    if relative:
        return new_x, new_y, np.minimum(new_w, 1. - new_x), np.minimum(new_h, 1. - new_y)
    new_x, new_y, new_w, new_h = [c.astype(dtype) for c in (new_x, new_y, new_w, new_h)]
    return new_x, new_y, np.minimum(new_w, image_w - new_x), np.minimum(new_h, image_h - new_y)


def convert_box_format(boxes, target=None, source=None, dezoom_factor=1., normalize_mode=None, image_h=None,
                       image_w=None, **_):
    """poly / xyxy / xywh conversions with the optional steps of the reference in its order
    (utils/image/bounding_box/converter.py:74-178), NumPy: `normalize_mode='absolute'` turns relative float boxes into
    int32 pixels (truncated float32 product, :180-200), then `dezoom_factor` scales the boxes about their centres."""
    if isinstance(boxes, dict):
        out = boxes.copy()
        out['boxes'] = convert_box_format(boxes['boxes'], target=target, source=boxes.get('format', source),
                                          dezoom_factor=dezoom_factor, normalize_mode=normalize_mode, image_h=image_h,
                                          image_w=image_w)
        out['format'] = target
        return out
    if source is None:
        raise ValueError('You must provide the `source` format')
    if target is None:
        target = source
    boxes = np.asarray(boxes)
    if boxes.shape[0] == 0:
        return boxes
    if boxes.ndim == 1 or (source == 'poly' and boxes.ndim == 2):
        boxes = boxes[None]
    if normalize_mode == 'absolute' and boxes.dtype.kind == 'f':
        if image_h is None or image_w is None:
            raise ValueError('normalize_mode="absolute" needs image_h and image_w')
        factors = np.array([image_w, image_h] * (1 if source == 'poly' else 2), dtype=boxes.dtype)
        boxes = (boxes * factors).astype(np.int32)
    elif normalize_mode not in (None, 'absolute'):
        raise NotImplementedError('normalize_mode {!r} is not available in this build'.format(normalize_mode))
    if source == target and dezoom_factor == 1.:
        return boxes
    if source == 'xywh':
        x, y, w, h = [boxes[..., k] for k in range(4)]
    elif source == 'xyxy':
        x, y = boxes[..., 0], boxes[..., 1]
        w, h = boxes[..., 2] - x, boxes[..., 3] - y
    elif source == 'poly':
        xy_min, xy_max = boxes.min(axis=-2), boxes.max(axis=-2)
        x, y = xy_min[..., 0], xy_min[..., 1]
        wh = xy_max - xy_min
        w, h = wh[..., 0], wh[..., 1]
    else:
        raise ValueError('Invalid `source` format : {}'.format(source))
    if dezoom_factor != 1.:
        x, y, w, h = dezoom_box(x, y, w, h, dezoom_factor, image_h=image_h, image_w=image_w)
    if target == 'xywh':
        return np.stack([x, y, w, h], axis=-1)
    if target == 'xyxy':
        return np.stack([x, y, x + w, y + h], axis=-1)
    raise ValueError('Invalid `target` format : {}'.format(target))


def nms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, *, method=None, source=None, **kwargs):
    if isinstance(boxes, dict):
        scores = boxes.get('scores', scores)
        boxes = convert_box_format(boxes, target='xyxy', source=source)['boxes']
    elif source not in (None, 'xyxy'):
        boxes = convert_box_format(boxes, target='xyxy', source=source)
    if nms_threshold >= 1. or len(boxes) == 0:
        return boxes, scores, np.ones((len(boxes),), dtype=bool)
    if not callable(method):
        if method is None:
            method = 'lanms'
        if method not in _nms_methods:
            raise ValueError('Unknown NMS method !\n  Accepted : {}\n  Got : {}'.format(tuple(_nms_methods.keys()), method))
        method = _nms_methods[method]
    return method(boxes, scores, max_output_size, nms_threshold, **kwargs)


def _dispatch(keys):
    if isinstance(keys, str):
        keys = (keys,)

    def wrapper(fn):
        _nms_methods.update({k: fn for k in keys})
        return fn
    return wrapper


nms.dispatch = _dispatch
nms.methods = _nms_methods


@nms.dispatch(('lanms', 'locality_aware_nms'))
def lanms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, merge_threshold=0.3, max_iter=None,
          merge_method='union', tile_size=512, ctx=None, **_):
    """utils/image/bounding_box/locality_aware_nms.py:22-57 on the GPU.

    Returns (boxes, scores, valid_mask) in score-descending order, like the reference (which pads to
    `tile_size` only in graph mode; no padding here)."""
    ctx = ctx or default_context()
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(1, -1, 4)
    n = b.shape[1]
    dev = ctx.device
    d_scores = None
    if scores is not None:
        s = np.ascontiguousarray(scores, dtype=np.float32).reshape(1, n)
        d_scores = torch.from_numpy(s).to(dev)
    out = ctx.lanms(torch.from_numpy(b).to(dev), d_scores, torch.tensor([n], dtype=torch.int32, device=dev),
                    merge_threshold=merge_threshold, nms_threshold=nms_threshold, merge_method=merge_method,
                    max_iter=max_iter or 0, max_out=max_output_size or 0, full=True)
    merged = out['merged'][0].cpu().numpy()
    keep = out['keep'][0].cpu().numpy().astype(bool)
    if scores is None:
        return merged, None, keep
    order = np.argsort(-np.asarray(scores, dtype=np.float64), kind='stable')
    return merged[order], np.asarray(scores)[order], keep[order]


def _greedy_nms_on_gpu(boxes, scores, nms_threshold, ctx):
    """Suppression only: the library's LANMS entry point with a merge threshold no IoU reaches, which leaves its
    second stage, the reference's sorted-NMS fixed point (non_max_suppression.py:117-138,191-220).
    -> (boxes with the suppressed ones zeroed, keep mask, rank order), the first two in INPUT order."""
    ctx = ctx or default_context()
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(1, -1, 4)
    n = b.shape[1]
    dev = ctx.device
    d_scores = None if scores is None else torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32).reshape(1, n)).to(dev)
    out = ctx.lanms(torch.from_numpy(b).to(dev), d_scores, torch.tensor([n], dtype=torch.int32, device=dev),
                    merge_threshold=2.0, nms_threshold=nms_threshold, full=True)
    order = np.arange(n) if scores is None else np.argsort(-np.asarray(scores, dtype=np.float64), kind='stable')
    keep = out['keep'][0].cpu().numpy().astype(bool)
    return b[0] * keep[:, None].astype(np.float32), keep, order


def _limit(valid_ranked, max_output_size):
    """`_get_valid_mask` of the reference (non_max_suppression.py:255-259): the first `max_output_size` survivors in
    rank order stay valid; the others keep their coordinates but leave the mask."""
    if max_output_size is None:
        return valid_ranked
    return valid_ranked & (np.cumsum(valid_ranked.astype(np.int32)) <= max_output_size)


@nms.dispatch(('nms', 'standard', 'fast', 'padded_nms', 'padded'))
def fast_nms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, tile_size=512, ctx=None, **_):
    """`fast_nms` / `padded_nms` of the reference (non_max_suppression.py:117-189): boxes sorted by score
    (descending), a box is dropped when a kept better-ranked box overlaps it with IoU >= nms_threshold.  Returns
    (sorted boxes with the dropped ones zeroed, sorted scores, valid mask); `tile_size` only shapes the reference's
    tensor program (its two methods select the same boxes) and is ignored here."""
    kept_boxes, keep, order = _greedy_nms_on_gpu(boxes, scores, nms_threshold, ctx)
    return kept_boxes[order], None if scores is None else np.asarray(scores)[order], _limit(keep[order], max_output_size)


@nms.dispatch('tensorflow')
def tensorflow_nms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, ctx=None, **_):
    """The reference's wrapper of `tf.image.non_max_suppression` (non_max_suppression.py:80-102): boxes and scores
    come back unchanged in input order with the mask of the selected ones (scores default to 1).  TensorFlow drops a
    box when IoU > threshold, the kernel here when IoU >= threshold: the two differ only on an exact tie."""
    boxes = np.asarray(boxes)
    if scores is None:
        scores = np.ones((len(boxes),), dtype=np.float32)
    _, keep, order = _greedy_nms_on_gpu(boxes, scores, nms_threshold, ctx)
    valid = np.zeros(len(boxes), dtype=bool)
    valid[order] = _limit(keep[order], max_output_size)
    return boxes, scores, valid
