"""Box filters for `BaseOCR.infer(box_filters=...)`.

Interface of the reference (utils/image/bounding_box/filters.py:20-78,200-241): a filter is any callable
`f(boxes=, indices=, rows=, **kw)` that answers with the positions of the boxes to keep (or a boolean mask when it
derives from `BoxFilter`); filters apply in sequence and a box survives only if all of them keep it.  `RegionFilter`
`SizeFilter` and the frame-to-frame `RepetitionFilter` of the stream are provided.

Built differently from the reference: a filter here is a list of vectorised `Bound`s (one measured quantity per
box and a half-open or closed interval for it), all evaluated in one pass; the chain keeps ONE index array into
the original boxes and materialises boxes / rows / indices once at the end.  Host NumPy, a few hundred boxes.
"""
import numpy as np

from .boxes import convert_box_format


# ---- measured quantities of xyxy boxes -------------------------------------------------------------------------

def _width(b):
    return b[:, 2] - b[:, 0]


def _height(b):
    return b[:, 3] - b[:, 1]


def _area(b):
    return _height(b) * _width(b)


def _center_x(b):
    return (b[:, 0] + b[:, 2]) / 2.


def _center_y(b):
    return (b[:, 1] + b[:, 3]) / 2.


class Bound:
    """`lo <= measure(boxes)` and `measure(boxes) < hi` (`<= hi` when `closed`); a missing side is unbounded."""

    def __init__(self, measure, lo=None, hi=None, closed=False, strict_lo=False):
        self.measure, self.lo, self.hi, self.closed, self.strict_lo = measure, lo, hi, closed, strict_lo

    def holds(self, boxes):
        value = self.measure(boxes)
        ok = np.ones(len(boxes), dtype=bool)
        if self.lo is not None:
            ok &= (value > self.lo) if self.strict_lo else (value >= self.lo)
        if self.hi is not None:
            ok &= (value <= self.hi) if self.closed else (value < self.hi)
        return ok


def _as_positions(answer, n):
    """What a filter answered -> sorted unique positions in [0, n)."""
    answer = np.asarray(answer)
    if answer.dtype == bool:
        return np.flatnonzero(answer)
    return answer.astype(np.int64).reshape(-1)


def filter_boxes(filters, boxes, indices=None, rows=None, **kwargs):
    """Applies `filters` (one callable or a sequence) in order; returns (boxes, indices, rows) of the survivors.
    Same contract as filters.py:22-49: later filters see only what earlier ones kept, positions are relative to what
    a filter is shown, an empty set stops the chain."""
    chain = [filters] if callable(filters) else list(filters)
    alive = np.arange(len(boxes))                       # positions in the caller's arrays
    for f in chain:
        if len(alive) == 0:
            break
        shown_rows = None if rows is None else [rows[i] for i in alive]
        shown_idx = None if indices is None else [indices[i] for i in alive]
        alive = alive[_as_positions(f(boxes=boxes[alive], indices=shown_idx, rows=shown_rows, **kwargs), len(alive))]
    if len(alive) == len(boxes):
        return boxes, indices, rows
    return (boxes[alive], None if indices is None else [indices[i] for i in alive],
            None if rows is None else [rows[i] for i in alive])


class BoxFilter:
    """Base class with the reference's hooks (filters.py:51-76): `filter(...)` gives a mask or positions, `start` /
    `finish` bracket it, the call returns positions."""

    bounds = ()

    def __call__(self, boxes, indices=None, rows=None, source='xyxy', **kwargs):
        self.start()
        try:
            answer = self.filter(boxes=boxes, indices=indices, rows=rows, source=source)
        finally:
            self.finish()
        return _as_positions(answer, len(boxes))

    def start(self):
        pass

    def finish(self):
        pass

    def filter(self, boxes, indices=None, rows=None, **kwargs):
        """All `bounds` of the filter at once."""
        if not self.bounds:
            raise NotImplementedError()
        ok = np.ones(len(boxes), dtype=bool)
        for bound in self.bounds:
            ok &= bound.holds(boxes)
        return ok


class RegionFilter(BoxFilter):
    """Boxes that touch a region (`mode='overlap'`: open intersection) or whose centre lies in it (`mode='center'`,
    borders included), filters.py:200-220.  `region` is relative xyxy, or the format named by `source`."""

    def __init__(self, region, mode='overlap', source='xyxy', **_):
        self.mode = mode
        x0, y0, x1, y1 = convert_box_format(np.asarray(region, dtype=np.float32).reshape(1, 4), target='xyxy', source=source)[0]
        self.x_min, self.y_min, self.x_max, self.y_max = x0, y0, x1, y1
        if mode == 'overlap':
            self.bounds = (Bound(lambda b: b[:, 0], hi=x1), Bound(lambda b: b[:, 2], lo=x0, strict_lo=True),
                           Bound(lambda b: b[:, 1], hi=y1), Bound(lambda b: b[:, 3], lo=y0, strict_lo=True))
        elif mode == 'center':
            self.bounds = (Bound(_center_x, lo=x0, hi=x1, closed=True), Bound(_center_y, lo=y0, hi=y1, closed=True))
        else:
            self.bounds = ()

    def filter(self, boxes, indices=None, rows=None, **kwargs):
        if not self.bounds:
            raise ValueError('Unknown mode for region filtering : {} !'.format(self.mode))
        return super().filter(boxes)


class SizeFilter(BoxFilter):
    """Height / width / area limits, each `min <= value < max` (filters.py:222-241); None = no limit."""

    def __init__(self, min_h=None, min_w=None, max_h=None, max_w=None, min_area=None, max_area=None, **_):
        self.min_h, self.max_h, self.min_w, self.max_w = min_h, max_h, min_w, max_w
        self.min_area, self.max_area = min_area, max_area
        self.bounds = (Bound(_height, min_h, max_h), Bound(_width, min_w, max_w), Bound(_area, min_area, max_area))

    def filter(self, boxes, indices=None, rows=None, **kwargs):
        return super().filter(boxes) if len(boxes) else np.zeros((0,), dtype=bool)


# ---- frame-to-frame repetition (camera / video stream) ----------------------------------------------------------

def rows_iou(a, b):
    """IoU matrix (len(a), len(b)) of two sets of xyxy rows; 0 where the union is empty
    (utils/image/bounding_box/metrics.py:19-46)."""
    a, b = np.asarray(a, dtype=np.float32).reshape(-1, 4), np.asarray(b, dtype=np.float32).reshape(-1, 4)
    iw = np.maximum(np.float32(0), np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]))
    ih = np.maximum(np.float32(0), np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]))
    inter = iw * ih
    area_a = (a[:, 3] - a[:, 1]) * (a[:, 2] - a[:, 0])
    area_b = (b[:, 3] - b[:, 1]) * (b[:, 2] - b[:, 0])
    union = area_a[:, None] - inter + area_b[None, :]
    out = np.zeros_like(inter)
    np.divide(inter, union, out=out, where=union != 0)
    return out


def _shape_of(rows, indices):
    """What two paragraphs must share before their rows are compared: the number of rows and, when `indices` lists the
    words of every row, the number of words per row (filters.py:127-138)."""
    if isinstance(indices, (list, tuple)):
        return tuple(len(i) if isinstance(i, (list, tuple)) else 1 for i in indices)
    return (1,) * len(rows)


class _Track:
    """A paragraph waiting to be seen again at the same place."""
    __slots__ = ('shape', 'box', 'rows', 'indices', 'first_frame', 'seen', 'missed', 'hit')

    def __init__(self, shape, box, rows, indices, frame):
        self.shape, self.box, self.rows, self.indices = shape, box, rows, indices
        self.first_frame, self.seen, self.missed, self.hit = frame, 1, 0, True


class RepetitionFilter(BoxFilter):
    """Keeps a paragraph only once it has been detected at the same place in `n_repeat` frames
    (filters.py:88-188): a paragraph matches a waiting one of the same shape when every row of the waiting one
    overlaps one of its rows with IoU > `iou_threshold`; a first sighting waits, exactly one match counts, several
    matches are ambiguous and count for nothing.  Waiting paragraphs not seen for more than `max_unseen` frames are
    dropped.  With `use_memory`, a paragraph that was let through once is never let through again.

    One call = one frame (`start()` / `finish()` bracket it, as `BoxFilter.__call__` does)."""

    def __init__(self, iou_threshold=0.5, n_repeat=2, max_unseen=3, use_memory=False):
        self.iou_threshold, self.n_repeat, self.max_unseen, self.use_memory = iou_threshold, n_repeat, max_unseen, use_memory
        self.index = 0
        self.waiting, self.memory, self._fresh = [], [], []

    def __len__(self):
        return len(self.waiting)

    def clear(self):
        self.waiting = []

    def start(self):
        self.index += 1
        self._fresh = []
        for t in self.waiting:
            t.hit = False

    def finish(self):
        for t in self.waiting:
            t.missed = 0 if t.hit else t.missed + 1
        self.waiting = [t for t in self.waiting if t.missed <= self.max_unseen] + self._fresh
        self._fresh = []

    def _matches(self, pool, shape, rows):
        """Positions in `pool` of the tracks of this shape whose rows all sit on rows of `rows`."""
        same = [i for i, t in enumerate(pool) if t.shape == shape]
        if not same:
            return []
        iou = rows_iou(rows, np.concatenate([pool[i].rows for i in same]))
        if len(rows) == 1:
            # a single row against single-row tracks: one column per track
            return [same[j] for j in np.flatnonzero(iou[0] > self.iou_threshold)]
        best = iou.max(axis=0).reshape(len(same), len(rows))
        return [same[j] for j in np.flatnonzero(np.all(best > self.iou_threshold, axis=1))]

    def _keep(self, box, rows, indices):
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, 4)
        shape = _shape_of(rows, indices)
        if self.use_memory and self._matches(self.memory, shape, rows):
            return False
        if self.n_repeat <= 1:
            return True
        found = self._matches(self.waiting, shape, rows)
        if not found:
            self._fresh.append(_Track(shape, box, rows, indices, self.index))
        elif len(found) == 1:
            t = self.waiting[found[0]]
            t.seen += 1
            t.hit = True
            if t.seen >= self.n_repeat:
                del self.waiting[found[0]]
                if self.use_memory:
                    self.memory.append(t)
                return True
        return False

    def filter(self, boxes, indices=None, rows=None, **kwargs):
        if rows is None:
            rows = [np.asarray(b).reshape(1, 4) for b in boxes]
        if indices is None:
            indices = [None] * len(boxes)
        return np.array([self._keep(b, r, i) for b, r, i in zip(boxes, rows, indices)], dtype=bool)
