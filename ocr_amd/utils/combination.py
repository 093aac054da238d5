"""Grouping of word boxes into lines and paragraphs on the host (NumPy).

Same rules and results as the reference's `combine_boxes` chain
(utils/image/bounding_box/combination.py:150-361): horizontal grouping (same line), vertical
grouping (stacked lines of similar height), then grouping of results that contain each other, with
the rows of a paragraph merged per text line and aligned on their left border.  Connected
components are found with a union-find; groups are listed by their smallest member and members in
ascending order before the positional sort, which is the order the reference produces.

This is host-side bookkeeping on at most a few hundred kept boxes per page; it decides which crops
the recogniser sees when `BaseOCR.infer(combine=True)` (models/ocr/base_ocr.py:154-157).
"""
import numpy as np


def _components(mask):
    """Connected components of the symmetric closure of a boolean adjacency matrix."""
    n = len(mask)
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for a, b in zip(*np.nonzero(mask)):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    groups = {}
    for i in range(n):
        groups.setdefault(find(i), []).append(i)
    return [groups[k] for k in sorted(groups)]


def _union(boxes):
    if len(boxes) == 1:
        return boxes[0]
    return np.concatenate([boxes[:, :2].min(axis=0), boxes[:, 2:].max(axis=0)])


def _flatten(lists):
    if len(lists) == 1:
        return lists[0]
    out = []
    for item in lists:
        out.extend(item)
    return out


def _merge_rows(rows, indices):
    """Rows of one paragraph that overlap vertically by half the mean height become one row."""
    if len(rows) < 2:
        return rows, indices
    h_mean = np.mean(rows[:, 3] - rows[:, 1])
    overlap = np.maximum(0., np.minimum(rows[:, None, 3], rows[None, :, 3]) -
                         np.maximum(rows[:, None, 1], rows[None, :, 1])) / h_mean
    groups = _components(overlap >= 0.5)
    return (np.array([_union(rows[g]) for g in groups]),
            [_flatten([indices[i] for i in g]) for g in groups])


def _align_rows(rows, mode, center_threshold=1e-2):
    if len(rows) == 1:
        return rows
    if mode == 'auto':
        xc = (rows[:, 0] + rows[:, 2]) / 2.
        if not np.all(np.abs(xc - np.mean(xc)) < center_threshold):
            mode = 'left'
    if mode == 'left':
        rows[:, 0] = np.min(rows[:, 0])
    elif mode == 'right':
        rows[:, 2] = np.max(rows[:, 2])
    return rows


def _group(boxes, indices, groups, sort_axis, rows=None, check_rows=None, align_borders=None,
           center_threshold=1e-2):
    if rows is not None:
        check_rows = True if check_rows is None else check_rows
        align_borders = 'auto' if align_borders is None else align_borders
    out_boxes, out_indices, members = [], [], []
    for g in groups:
        if len(g) > 1:
            g = sorted(g, key=lambda i: boxes[i, sort_axis])
        sub = boxes[g]
        out_boxes.append(_union(sub))
        if rows is None:
            out_indices.append([indices[i] for i in g])
            members.append(sub)
        elif len(g) == 1:
            out_indices.append(indices[g[0]])
            members.append(rows[g[0]])
        else:
            out_indices.append(_flatten([indices[i] for i in g]))
            members.append(np.concatenate([rows[i] for i in g], axis=0))
    if check_rows:
        merged = [_merge_rows(r, i) for r, i in zip(members, out_indices)]
        members, out_indices = [m[0] for m in merged], [m[1] for m in merged]
    if align_borders:
        members = [_align_rows(r, align_borders, center_threshold) for r in members]
    return np.array(out_boxes), out_indices, members


def _as_boxes(boxes):
    if isinstance(boxes, dict):
        if boxes.get('format', 'xyxy') != 'xyxy':
            raise ValueError('combine_boxes expects xyxy boxes')
        boxes = boxes['boxes']
    return np.asarray(boxes)


def combine_boxes_horizontal(boxes, *, indices=None, h_factor=1.5, x_threshold=None, y_overlap_threshold=0.55, **_):
    """Words on the same line: horizontal gap <= x_threshold (default 1.5 x median height) and vertical
    overlap >= 55 % of the taller box (combination.py:152-234)."""
    boxes = _as_boxes(boxes)
    if indices is None:
        indices = list(range(len(boxes)))
    if len(boxes) <= 1:
        return boxes, [indices], [boxes]
    h = boxes[:, 3] - boxes[:, 1]
    if x_threshold is None:
        x_threshold = np.median(h) * h_factor
    max_h = np.maximum(h[:, None], h[None, :])
    gap = np.abs(boxes[:, None, 2] - boxes[None, :, 0])
    overlap_x = (np.minimum(boxes[:, None, 2], boxes[None, :, 2]) - np.maximum(boxes[:, None, 0], boxes[None, :, 0])) > 0.
    gap = gap * ~overlap_x
    overlap_y = np.maximum(0., np.minimum(boxes[:, None, 3], boxes[None, :, 3]) -
                           np.maximum(boxes[:, None, 1], boxes[None, :, 1]))
    frac = np.zeros_like(overlap_y)
    np.divide(overlap_y, max_h, out=frac, where=max_h > 0)
    mask = (gap <= x_threshold) & (frac >= y_overlap_threshold)
    return _group(boxes, indices, _components(mask), 0)


def combine_boxes_vertical(boxes, *, indices=None, y_threshold=None, h_threshold=0.02, x_overlap_threshold=0.,
                           shift_factor=0.5, **_):
    """Lines of one paragraph: vertical gap <= y_threshold (default median height / 4), heights within
    h_threshold, left halves overlapping horizontally (combination.py:238-329)."""
    boxes = _as_boxes(boxes)
    if indices is None:
        indices = list(range(len(boxes)))
    if len(boxes) <= 1:
        return boxes, [indices], [boxes]
    h = boxes[:, 3] - boxes[:, 1]
    if y_threshold is None:
        y_threshold = max(np.median(h) / 4., 1e-2)
    h_diff = np.abs(h[:, None] - h[None, :])
    gap = np.abs(boxes[:, None, 3] - boxes[None, :, 1])
    overlap_y = (np.minimum(boxes[:, None, 3], boxes[None, :, 3]) - np.maximum(boxes[:, None, 1], boxes[None, :, 1])) > 0
    gap = gap * ~overlap_y
    shifted_w = (boxes[:, 2] - boxes[:, 0]) * shift_factor
    shifted_x_max = boxes[:, 2] - shifted_w
    overlap_x = (np.minimum(shifted_x_max[:, None], shifted_x_max[None, :]) -
                 np.maximum(boxes[:, None, 0], boxes[None, :, 0]))
    if x_overlap_threshold == 0.:
        x_mask = overlap_x > 0.
    else:
        x_mask = overlap_x / np.minimum(shifted_w[:, None], shifted_w[None, :]) > x_overlap_threshold
    mask = (gap <= y_threshold) & (h_diff <= h_threshold) & x_mask
    return _group(boxes, indices, _components(mask), 1)


def _ioa_matrix(boxes):
    """Intersection over the area of the row box (utils/image/bounding_box/metrics.py:48-71)."""
    area = (boxes[:, 3] - boxes[:, 1]) * (boxes[:, 2] - boxes[:, 0])
    iw = np.maximum(0, np.minimum(boxes[:, None, 2], boxes[None, :, 2]) - np.maximum(boxes[:, None, 0], boxes[None, :, 0]))
    ih = np.maximum(0, np.minimum(boxes[:, None, 3], boxes[None, :, 3]) - np.maximum(boxes[:, None, 1], boxes[None, :, 1]))
    out = np.zeros_like(iw)
    np.divide(iw * ih, area[:, None], out=out, where=area[:, None] != 0)
    return out


def combine_boxes_overlap(boxes, indices=None, overlap_threshold=0.5, rows=None, **kwargs):
    """Results that mostly contain each other are one paragraph (combination.py:333-337)."""
    boxes = _as_boxes(boxes)
    if indices is None:
        indices = list(range(len(boxes)))
    mask = _ioa_matrix(boxes) > overlap_threshold
    return _group(boxes, indices, _components(mask), 1, rows=rows, check_rows=kwargs.get('check_rows'),
                  align_borders=kwargs.get('align_borders'), center_threshold=kwargs.get('center_threshold', 1e-2))


def combine_boxes(boxes, indices=None, **kwargs):
    """words -> lines -> paragraphs (combination.py:341-361).

    Returns (combined_boxes (P,4), combined_indices, rows) with rows[p] the (k,4) line boxes of paragraph p."""
    kwargs.pop('image', None)
    kwargs.pop('source', None)
    b, idx, _ = combine_boxes_horizontal(boxes, indices=indices, **kwargs)
    b, idx, rows = combine_boxes_vertical(b, indices=idx, **kwargs)
    return combine_boxes_overlap(b, indices=idx, rows=rows, **kwargs)
