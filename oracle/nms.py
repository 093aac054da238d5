"""Locality-aware NMS and the matrix NMS of the reference, restated in NumPy (test infrastructure).

Follows `utils/image/bounding_box/locality_aware_nms.py:22-126` and
`non_max_suppression.py:104-138,191-220,247-259`.  The dense N x N formulation is kept on
purpose: this file is the literal statement of the algorithm; `oracle/lanms_c.c` is the same
algorithm without the N x N temporaries for sizes where NumPy would need gigabytes.
"""
import numpy as np

from .boxes import compute_iou, divide_no_nan


def sort_boxes(boxes, scores):
    """Score-descending order, non_max_suppression.py:247-253.

    The reference flips an ascending argsort whose tie order is backend-defined.  Ties are
    resolved here towards the lower original index (stable descending sort); inputs used for
    parity either have no ties or are compared after this canonical ordering.
    """
    if scores is None:
        return boxes, None, np.arange(len(boxes))
    order = np.argsort(-np.asarray(scores, dtype=np.float64), kind='stable')
    return boxes[order], scores[order], order


def _merge_boxes(boxes, box_slice, mask, merge_method):
    """locality_aware_nms.py:99-126."""
    mask3 = mask[..., None]
    merged = box_slice[None] * mask3.astype(box_slice.dtype)
    if merge_method == 'union':
        union_xy_min = np.min(np.where(mask3, merged[..., :2], np.float32(np.inf)), axis=1)
        union_xy_max = np.max(merged[..., 2:], axis=1)
        return np.concatenate([np.minimum(boxes[:, :2], union_xy_min),
                               np.maximum(boxes[:, 2:], union_xy_max)], axis=1)
    elif merge_method == 'average':
        sum_coords = boxes + np.sum(merged, axis=1)
        count = (1 + np.count_nonzero(mask3, axis=1)).astype(sum_coords.dtype)
        return divide_no_nan(sum_coords, count)
    raise ValueError(merge_method)


def self_merging_step(boxes, tri, merge_threshold, merge_method, weights=None):
    """One iteration of `_self_merging_body` (locality_aware_nms.py:74-97).

    merge_method 'weighted' is NOT in the reference (an opt-in extension: the score-weighted merge of the original
    EAST LANMS inside the reference's iteration): box i becomes the mean of itself and the boxes it absorbs weighted by
    `weights`, and takes over their weights; a box that absorbs nothing is unchanged.  Returns the new weights as a
    third value in that mode."""
    iou = compute_iou(boxes, as_matrix=True)
    iou = iou * (tri & (iou >= merge_threshold)).astype(iou.dtype)
    can_suppress_others = (np.max(iou, axis=0) < merge_threshold).astype(iou.dtype)[:, None]
    merging_mask = iou * can_suppress_others >= merge_threshold
    suppressed_box = np.any(merging_mask, axis=0)[:, None]
    cond = bool(np.any(iou * suppressed_box.astype(iou.dtype) >= merge_threshold))
    if merge_method == 'weighted':
        w = weights.astype(np.float64)
        m = merging_mask.astype(np.float64)
        sw = m @ w
        num = w[:, None] * boxes.astype(np.float64) + m @ (w[:, None] * boxes.astype(np.float64))
        den = np.where(sw > 0, w + sw, 1.0)
        merged = np.where((sw > 0)[:, None], num / den[:, None], boxes.astype(np.float64)).astype(np.float32)
        keep = np.logical_not(suppressed_box)
        return merged * keep.astype(merged.dtype), cond, ((w + sw) * keep[:, 0]).astype(np.float32)
    merged = _merge_boxes(boxes, boxes, merging_mask, merge_method)
    merged = merged * np.logical_not(suppressed_box).astype(merged.dtype)
    return merged, cond


def self_merging(boxes, merge_threshold, merge_method='union', max_iter=None, weights=None):
    n = len(boxes)
    if max_iter is None:
        max_iter = n
    idx = np.arange(n)
    tri = idx[None, :] > idx[:, None]
    cond, it = True, 0
    if merge_method == 'weighted':
        weights = np.ones(n, np.float32) if weights is None else np.asarray(weights, dtype=np.float32)
    while cond and it < max_iter:
        if merge_method == 'weighted':
            boxes, cond, weights = self_merging_step(boxes, tri, merge_threshold, merge_method, weights)
        else:
            boxes, cond = self_merging_step(boxes, tri, merge_threshold, merge_method)
        it += 1
    return boxes, it


def self_suppression(boxes, iou_threshold):
    """Matrix NMS on rank-ordered boxes, non_max_suppression.py:191-220."""
    n = len(boxes)
    idx = np.arange(n)
    tri = idx[None, :] > idx[:, None]
    iou = compute_iou(boxes, as_matrix=True)
    iou = iou * (tri & (iou >= iou_threshold)).astype(iou.dtype)
    iou_sum = np.sum(iou)
    cond = True
    while cond:
        can_suppress_others = (np.max(iou, axis=0) < iou_threshold).astype(iou.dtype)[:, None]
        iou_after = iou * (np.max(can_suppress_others * iou, axis=0) < iou_threshold
                           ).astype(iou.dtype)[:, None]
        iou_sum_new = np.sum(iou_after)
        cond = bool(iou_sum - iou_sum_new > iou_threshold)
        iou, iou_sum = iou_after, iou_sum_new
    suppressed_box = np.any(iou > 0, axis=0)[:, None]
    return boxes * np.logical_not(suppressed_box).astype(boxes.dtype)


def get_valid_mask(boxes, max_output_size=None):
    mask = np.any(boxes > 0, axis=1)
    if max_output_size is not None:
        mask = mask & (np.cumsum(mask.astype(np.int32)) <= max_output_size)
    return mask


def fast_nms(boxes, scores=None, max_output_size=None, nms_threshold=0.25):
    """non_max_suppression.py:117-138.  Returns (boxes, scores, valid_mask, order)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    boxes, scores, order = sort_boxes(boxes, scores)
    if len(boxes) == 0:
        return boxes, scores, np.zeros((0,), dtype=bool), order
    selected = self_suppression(boxes, np.float32(nms_threshold))
    return selected, scores, get_valid_mask(selected, max_output_size), order


def lanms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, *,
          merge_threshold=0.3, max_iter=None, merge_method='union'):
    """locality_aware_nms.py:22-57.  Returns (boxes, scores, valid_mask, order, n_iter).

    `order[r]` is the input index of the box at output position r.
    """
    boxes = np.asarray(boxes, dtype=np.float32)
    if scores is not None:
        scores = np.asarray(scores, dtype=np.float32)
    boxes, scores, order = sort_boxes(boxes, scores)
    if len(boxes) == 0:
        return boxes, scores, np.zeros((0,), dtype=bool), order, 0
    nms_threshold = np.float32(nms_threshold)
    merge_threshold = np.float32(merge_threshold)
    boxes, n_iter = self_merging(boxes, merge_threshold, merge_method, max_iter, weights=scores)
    mask = np.any(boxes > 0, axis=1)
    if nms_threshold < merge_threshold:
        mask = mask & fast_nms(boxes, None, max_output_size, nms_threshold)[2]
    return boxes, scores, mask, order, n_iter


def greedy_nms_keep(boxes, iou_threshold):
    """Classic greedy NMS on rank-ordered boxes (`IoU >= thr` suppresses).

    Not part of the reference: used by the tests to show that the matrix iteration above and a
    greedy scan select the same boxes.
    """
    boxes = np.asarray(boxes, dtype=np.float32)
    n = len(boxes)
    keep = np.zeros(n, dtype=bool)
    kept = []
    for i in range(n):
        ok = True
        if kept:
            iou = compute_iou(boxes[kept], boxes[i:i + 1], as_matrix=True)[:, 0]
            ok = not np.any(iou >= iou_threshold)
        if ok:
            keep[i] = True
            kept.append(i)
    return keep
