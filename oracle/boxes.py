"""Box arithmetic of the reference, restated in NumPy (test infrastructure, see oracle/__init__.py).

Follows `utils/image/bounding_box/metrics.py:18-46` (IoU), `converter.py:74-200` (format
conversion / normalisation) and `processing.py:111-132` (crop).
"""
import numpy as np


def divide_no_nan(a, b):
    """`x / y` with 0 where `y == 0` (reference: utils/keras/ops/numpy.py:58)."""
    a = np.asarray(a)
    b = np.asarray(b)
    out = np.zeros(np.broadcast(a, b).shape, dtype=np.result_type(a, b))
    np.divide(a, b, out=out, where=(b != 0))
    return out


def compute_iou(boxes1, boxes2=None, as_matrix=None):
    """IoU of xyxy boxes, operation order of metrics.py:30-46.

    union = area_1 - inter + area_2 (the subtraction sits in the middle), 0/0 -> 0.
    """
    if as_matrix is None:
        as_matrix = boxes2 is None
    if boxes2 is None:
        boxes2 = boxes1
    boxes1 = np.asarray(boxes1)
    boxes2 = np.asarray(boxes2)
    if boxes1.dtype.kind != 'f':
        boxes1 = boxes1.astype(np.float32)
    if boxes2.dtype.kind != 'f':
        boxes2 = boxes2.astype(np.float32)
    if len(boxes1) == 0 or len(boxes2) == 0:
        return np.zeros((0, 0), dtype=np.float32)
    b1 = [boxes1[..., k] for k in range(4)]
    b2 = [boxes2[..., k] for k in range(4)]
    if as_matrix:
        b1 = [b[..., None] for b in b1]
        b2 = [b[..., None, :] for b in b2]
    xmin_1, ymin_1, xmax_1, ymax_1 = b1
    xmin_2, ymin_2, xmax_2, ymax_2 = b2
    areas_1 = (ymax_1 - ymin_1) * (xmax_1 - xmin_1)
    areas_2 = (ymax_2 - ymin_2) * (xmax_2 - xmin_2)
    xmin, ymin = np.maximum(xmin_1, xmin_2), np.maximum(ymin_1, ymin_2)
    xmax, ymax = np.minimum(xmax_1, xmax_2), np.minimum(ymax_1, ymax_2)
    zero = np.zeros((), dtype=xmax.dtype)
    inter_w, inter_h = np.maximum(zero, xmax - xmin), np.maximum(zero, ymax - ymin)
    inter = inter_w * inter_h
    union = areas_1 - inter + areas_2
    return divide_no_nan(inter, union)


def poly_to_xyxy(polys):
    """(N,4,2) corner polygons -> (N,4) xyxy, converter.py:151-154,165-166.

    The far corner is rebuilt as `min + (max - min)`, which is not always `max` in float32.
    """
    polys = np.asarray(polys)
    if polys.shape[0] == 0:
        return np.zeros((0, 4), dtype=polys.dtype)
    xy_min, xy_max = polys.min(axis=-2), polys.max(axis=-2)
    wh = xy_max - xy_min
    return np.stack([xy_min[..., 0], xy_min[..., 1],
                     xy_min[..., 0] + wh[..., 0], xy_min[..., 1] + wh[..., 1]], axis=-1)


def convert_xyxy_xywh(boxes, source, target):
    """xywh <-> xyxy (converter.py:146-166)."""
    boxes = np.asarray(boxes)
    if source == target:
        return boxes
    if source == 'xywh':
        x, y, w, h = [boxes[..., k] for k in range(4)]
    elif source == 'xyxy':
        x, y = boxes[..., 0], boxes[..., 1]
        w, h = boxes[..., 2] - x, boxes[..., 3] - y
    else:
        raise ValueError('Invalid `source` format : {}'.format(source))
    if target == 'xywh':
        res = [x, y, w, h]
    elif target == 'xyxy':
        res = [x, y, x + w, y + h]
    else:
        raise ValueError('Invalid `target` format : {}'.format(target))
    return np.stack(res, axis=-1)


def to_absolute(boxes_rel, image_h, image_w):
    """Relative float xyxy -> absolute int32, converter.py:186-196.

    `int32(trunc(float_box * factors))`, the factors taking the box dtype.
    """
    boxes_rel = np.asarray(boxes_rel)
    factors = np.array([image_w, image_h, image_w, image_h], dtype=boxes_rel.dtype)
    return (boxes_rel * factors).astype(np.int32)


def to_relative(boxes_abs, image_h, image_w):
    """Absolute int xyxy -> relative float32, converter.py:193-194."""
    factors = np.array([image_w, image_h, image_w, image_h], dtype=np.float32)
    return np.asarray(boxes_abs).astype(np.float32) / factors


def crop_box(image, box_rel_xyxy):
    """`image[y0:y1, x0:x1]` with plain NumPy slice semantics (processing.py:128-132).

    A negative start index therefore wraps around exactly as in the reference and usually
    yields an empty crop, which `BaseOCR.infer` skips (base_ocr.py:182-184).
    """
    h, w = image.shape[0], image.shape[1]
    box = np.asarray(box_rel_xyxy).reshape(-1)
    if box.dtype.kind == 'f':
        x0, y0, x1, y1 = [int(v) for v in to_absolute(box, h, w)]
    else:
        x0, y0, x1, y1 = [int(v) for v in box]
    return (x0, y0, x1, y1), image[y0:y1, x0:x1]


def slice_bounds(start, stop, size):
    """Resolved (lo, hi) of the Python slice `[start:stop]` on an axis of length `size`."""
    lo, hi, _ = slice(start, stop).indices(size)
    return lo, max(lo, hi)
