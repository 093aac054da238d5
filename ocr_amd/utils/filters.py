"""Box filters of `BaseOCR.infer(box_filters=...)` (utils/image/bounding_box/filters.py:20-78,200-241): a filter is a
callable `f(boxes=, indices=, rows=) -> indexes to keep`, applied in sequence to the (combined) boxes before they are
recognised.  `RegionFilter` and `SizeFilter` are restated here; the frame-to-frame `RepetitionFilter` of the camera
stream is not built.  Host NumPy on a few hundred boxes."""
import numpy as np

from .boxes import convert_box_format


def filter_boxes(filters, boxes, indices=None, rows=None, **kwargs):
    """filters.py:22-49: a box is kept if it passes all the filters."""
    if callable(filters):
        filters = [filters]
    for f in filters:
        if len(boxes) == 0:
            return boxes, indices, rows
        keep = f(boxes=boxes, indices=indices, rows=rows, **kwargs)
        if len(keep) < len(boxes):
            boxes = boxes[keep]
            rows = [rows[idx] for idx in keep]
            indices = [indices[idx] for idx in keep]
    return boxes, indices, rows


class BoxFilter:
    """filters.py:51-76: `filter` returns a boolean mask or a list of indexes; the call returns indexes."""

    def __call__(self, boxes, indices, rows, source='xyxy', **kwargs):
        self.start()
        res = self.filter(boxes=boxes, indices=indices, rows=rows, source=source)
        self.finish()
        if isinstance(res, np.ndarray) and res.dtype == bool:
            res = np.where(res)[0]
        return res

    def start(self):
        pass

    def finish(self):
        pass

    def filter(self, boxes, indices, rows, **kwargs):
        raise NotImplementedError()


class RegionFilter(BoxFilter):
    """filters.py:200-220: keeps the boxes that overlap the region (`mode='overlap'`) or whose centre lies in it
    (`mode='center'`); `region` is relative [x_min, y_min, x_max, y_max] (or another format named by `source`)."""

    def __init__(self, region, mode='overlap', source='xyxy', **_):
        self.mode = mode
        region = np.asarray(region, dtype=np.float32).reshape(1, 4)
        self.x_min, self.y_min, self.x_max, self.y_max = convert_box_format(region, target='xyxy', source=source)[0]

    def filter(self, boxes, **kwargs):
        if self.mode == 'overlap':
            return np.logical_and(np.logical_and(boxes[:, 0] < self.x_max, boxes[:, 2] > self.x_min),
                                  np.logical_and(boxes[:, 1] < self.y_max, boxes[:, 3] > self.y_min))
        if self.mode == 'center':
            center = (boxes[:, :2] + boxes[:, 2:]) / 2.
            return np.logical_and(np.logical_and(self.x_min <= center[:, 0], center[:, 0] <= self.x_max),
                                  np.logical_and(self.y_min <= center[:, 1], center[:, 1] <= self.y_max))
        raise ValueError('Unknown mode for region filtering : {} !'.format(self.mode))


class SizeFilter(BoxFilter):
    """filters.py:222-241: height / width / area bounds, `min <= value < max`."""

    def __init__(self, min_h=None, min_w=None, max_h=None, max_w=None, min_area=None, max_area=None, **_):
        self.min_h, self.max_h = min_h, max_h
        self.min_w, self.max_w = min_w, max_w
        self.min_area, self.max_area = min_area, max_area

    def filter(self, boxes, **kwargs):
        valids = np.ones((len(boxes),), dtype=bool)
        h = boxes[:, 3] - boxes[:, 1]
        w = boxes[:, 2] - boxes[:, 0]
        for lo, hi, val in ((self.min_h, self.max_h, h), (self.min_w, self.max_w, w), (self.min_area, self.max_area, h * w)):
            if lo is not None:
                valids[val < lo] = False
            if hi is not None:
                valids[val >= hi] = False
        return valids
