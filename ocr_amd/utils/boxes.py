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


def convert_box_format(boxes, target=None, source=None, **_):
    """poly / xyxy / xywh conversions (utils/image/bounding_box/converter.py:74-178), NumPy."""
    if isinstance(boxes, dict):
        out = boxes.copy()
        out['boxes'] = convert_box_format(boxes['boxes'], target=target, source=boxes.get('format', source))
        out['format'] = target
        return out
    if source is None:
        raise ValueError('You must provide the `source` format')
    if target is None:
        target = source
    boxes = np.asarray(boxes)
    if boxes.shape[0] == 0 or source == target:
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
