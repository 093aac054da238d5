"""Grouping of word boxes into lines and paragraphs on the host (NumPy).

Same rules and results as the reference's `combine_boxes` chain
(utils/image/bounding_box/combination.py:150-361): horizontal grouping (same line), vertical
grouping (stacked lines of similar height), then grouping of results that contain each other, with
the rows of a paragraph merged per text line and aligned on their left border.  Connected
components are found with a union-find and kept as one label per box; a level of grouping then is
ONE stable ordering of the boxes by (group, position) whose runs give every group's hull, members
and index lists (groups in the order of their smallest member, ties of position in ascending box
order: the order the reference produces).

This is host-side bookkeeping on at most a few hundred kept boxes per page; it decides which crops
the recogniser sees when `BaseOCR.infer(combine=True)` (models/ocr/base_ocr.py:154-157).
"""
import numpy as np


def _labels(mask):
    """Connected components of the symmetric closure of a boolean adjacency matrix, as one label per node: the
    smallest node of its component (union-find with path halving)."""
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
    return np.fromiter((find(i) for i in range(n)), dtype=np.int64, count=n)


def _runs(labels, key=None):
    """Nodes ordered by (label, key, node) and the [start, end) of every label's run in that order.  Labels are the
    smallest member of a group, so the runs come in the order of the groups' smallest members; the sort is stable, so
    equal keys (and key=None) keep the members in ascending order."""
    order = np.argsort(labels, kind='stable') if key is None else np.lexsort((key, labels))
    lab = labels[order]
    starts = np.flatnonzero(np.concatenate([[True], lab[1:] != lab[:-1]]))
    return order, starts, np.concatenate([starts[1:], [len(order)]])


def _hulls(boxes, starts):
    """Bounding box of every run of `boxes` (already ordered; runs start at `starts`)."""
    return np.concatenate([np.minimum.reduceat(boxes[:, :2], starts), np.maximum.reduceat(boxes[:, 2:], starts)], axis=1)


def _fuse_text_lines(rows, words):
    """Rows of one paragraph that overlap vertically by half the mean row height become one row (its hull; the word
    lists concatenated in row order)."""
    top, bottom = rows[:, 1], rows[:, 3]
    shared = np.maximum(0., np.minimum(bottom[:, None], bottom[None, :]) - np.maximum(top[:, None], top[None, :]))
    order, starts, ends = _runs(_labels(shared / np.mean(bottom - top) >= 0.5))
    return _hulls(rows[order], starts), [[w for i in order[a:b] for w in words[i]] for a, b in zip(starts, ends)]


def _justify(rows, mode, center_threshold):
    """Rows of a paragraph share their left (or right) border; 'auto' leaves centred paragraphs (every row centre within
    `center_threshold` of the mean centre) alone and aligns the others on the left."""
    if mode == 'auto':
        centre = (rows[:, 0] + rows[:, 2]) / 2.
        mode = 'left' if np.abs(centre - np.mean(centre)).max() >= center_threshold else None
    if mode == 'left':
        rows[:, 0] = rows[:, 0].min()
    elif mode == 'right':
        rows[:, 2] = rows[:, 2].max()
    return rows


def _group(boxes, indices, labels, sort_axis, rows=None, check_rows=None, align_borders=None, center_threshold=1e-2):
    """One level of grouping from a label per box.  Boxes are ordered once by (group, position along `sort_axis`);
    every group's hull, members and index lists are runs of that order.
    rows is None (words -> lines, lines -> paragraphs): members[g] = the ordered boxes of group g, indices[g] = their
    index entries.  rows given (paragraphs that contain each other): members[g] = the rows of all its members, fused
    per text line and justified; indices[g] = one word list per row."""
    if len(boxes) == 0:
        return np.array([]), [], []
    order, starts, ends = _runs(labels, boxes[:, sort_axis])
    ordered = boxes[order]
    hulls = _hulls(ordered, starts)
    if rows is None:
        return (hulls, [[indices[i] for i in order[a:b]] for a, b in zip(starts, ends)],
                [ordered[a:b] for a, b in zip(starts, ends)])
    check_rows = True if check_rows is None else check_rows
    align_borders = 'auto' if align_borders is None else align_borders
    members, out_indices = [], []
    for a, b in zip(starts, ends):
        group = order[a:b]
        r = rows[group[0]] if b - a == 1 else np.concatenate([rows[i] for i in group], axis=0)
        words = indices[group[0]] if b - a == 1 else [w for i in group for w in indices[i]]
        if check_rows and len(r) > 1:
            r, words = _fuse_text_lines(r, words)
        if align_borders and len(r) > 1:
            r = _justify(r, align_borders, center_threshold)
        members.append(r)
        out_indices.append(words)
    return hulls, out_indices, members


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
    return _group(boxes, indices, _labels(mask), 0)


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
    return _group(boxes, indices, _labels(mask), 1)


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
    return _group(boxes, indices, _labels(mask), 1, rows=rows, check_rows=kwargs.get('check_rows'),
                  align_borders=kwargs.get('align_borders'), center_threshold=kwargs.get('center_threshold', 1e-2))


def combine_boxes(boxes, indices=None, **kwargs):
    """words -> lines -> paragraphs (combination.py:341-361).

    Returns (combined_boxes (P,4), combined_indices, rows) with rows[p] the (k,4) line boxes of paragraph p."""
    kwargs.pop('image', None)
    kwargs.pop('source', None)
    b, idx, _ = combine_boxes_horizontal(boxes, indices=indices, **kwargs)
    b, idx, rows = combine_boxes_vertical(b, indices=idx, **kwargs)
    return combine_boxes_overlap(b, indices=idx, rows=rows, **kwargs)
