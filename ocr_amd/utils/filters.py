"""Box filters for `BaseOCR.infer(box_filters=...)`.

Interface of the reference (utils/image/bounding_box/filters.py:20-78,200-241): a filter is any callable
`f(boxes=, indices=, rows=, **kw)` that answers with the positions of the boxes to keep (or a boolean mask when it
derives from `BoxFilter`); filters apply in sequence and a box survives only if all of them keep it.  `RegionFilter`
and `SizeFilter` are provided; the frame-to-frame `RepetitionFilter` of the camera stream is not built.

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
