"""OCR over a sequence of frames: the logic of `BaseOCR.stream` / `stream_fn` of the reference
(models/ocr/base_ocr.py:33-51,377-516,561-564,679-725) without its camera, display and file-saving surface.

Per frame: detect; skip the frame when nothing was found; compare the score map with the one of the last new frame
(Dice coefficient, on the GPU) and, on a run of more than `duplicate_tolerance` duplicates, let only some frames
through; recognise what the box filters keep (size, region, repetition over frames, not already emitted at the same
place); drop texts that are empty, a single word, without a letter, already emitted, or within an edit distance
of `max_dist` of one of the last `k` emitted texts; hand every remaining text to `callback`.

Frames come from any iterable (arrays, file names, dicts with 'image'); opening cameras, video files or HTTP screen
mirrors is not part of this library.
"""
import logging
from dataclasses import dataclass, field

import numpy as np
import torch

from ...utils.filters import RegionFilter, RepetitionFilter, SizeFilter

logger = logging.getLogger(__name__)


@dataclass
class StreamState:
    """base_ocr.py:33-51."""
    is_file: bool = False
    emitted_texts: set = field(default_factory=set, repr=False)
    last_texts: list = field(default_factory=list)
    last_boxes: np.ndarray = field(default=None, repr=False)
    n_skipped: int = 0
    n_duplicates: int = 0
    prev_output: object = field(default=None, repr=False)
    repetition_filter: RepetitionFilter = None

    def set_new_frame(self, output):
        self.prev_output = output
        self.last_texts = []
        self.last_boxes = None
        self.n_duplicates = 0


def edit_distance(hypothesis, truth, normalize=True):
    """Levenshtein distance with unit costs, divided by `len(truth)` when `normalize`
    (utils/text/metrics.py:86-159 with its default costs, `partial=False`)."""
    prev = np.arange(len(truth) + 1, dtype=np.float64)
    for i, h in enumerate(hypothesis, 1):
        cur = np.empty_like(prev)
        cur[0] = i
        for j, t in enumerate(truth, 1):
            cur[j] = min(prev[j] + 1, prev[j - 1] + (h != t), cur[j - 1] + 1)
        prev = cur
    dist = float(prev[-1])
    if normalize:
        # the reference divides by zero for an empty truth (inf / nan): an empty truth never reaches it in the stream
        return dist / len(truth) if len(truth) else (float('inf') if dist else float('nan'))
    return dist


def keep_text_result(ocr_result, reject, last_emitted=(), threshold=0., *, k=5, max_dist=0.2, skip_non_alpha=True,
                     skip_single_word=True, **_):
    """base_ocr.py:679-725: cleans `ocr_result['text']` in place and says whether it is worth emitting.  Texts
    rejected for being near one of the last `k` emitted texts are added to `reject`."""
    text = ocr_result['text'].replace('</s>', '').strip()
    ocr_result['text'] = text
    if not text:
        return False
    if skip_single_word and ' ' not in text:
        return False
    if skip_non_alpha and not any(c.isalpha() for c in text):
        return False
    if text in reject:
        return False
    scores = np.asarray(ocr_result['scores'])
    if threshold != 0. and np.any(scores <= threshold):
        if np.all(scores <= threshold):
            return False
        parts = text.split(' \n')
        text = ' \n'.join(p if s > threshold else '' for p, s in zip(parts, scores)).strip()
        ocr_result['text'] = text
    if last_emitted and k:
        for emitted in list(last_emitted)[-k:]:
            if edit_distance(emitted, text, normalize=True) < max_dist:
                reject.add(text)
                return False
    return True


def ioa_matrix(boxes, others):
    """Intersection over the area of `boxes` (utils/image/bounding_box/metrics.py:48-71), shape (len(boxes), len(others))."""
    a = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    b = np.asarray(others, dtype=np.float32).reshape(-1, 4)
    iw = np.maximum(np.float32(0), np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]))
    ih = np.maximum(np.float32(0), np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]))
    area = ((a[:, 3] - a[:, 1]) * (a[:, 2] - a[:, 0]))[:, None]
    out = np.zeros_like(iw)
    np.divide(iw * ih, area, out=out, where=area != 0)
    return out


class OcrStream:
    """State and per-frame step of one stream.  `model` is a `BaseOCR`, `detector` an `EAST`."""

    def __init__(self, model, detector, *, filters=None, region=(0.2, 0.05, 0.6, 0.95), n_repeat=2, ioa_threshold=0.9,
                 callback=None, callbacks=None, duplicate_threshold=0.9, duplicate_tolerance=5, threshold=-0.2,
                 is_file=False, **infer_kwargs):
        self.model, self.detector = model, detector
        self.state = StreamState(is_file=is_file)
        if filters is None:                                        # base_ocr.py:605-610
            filters = [SizeFilter(min_h=0.025, min_w=0.1)]
            if region:
                filters.append(RegionFilter(region=list(region), source='xyxy', mode='center'))
            if n_repeat:
                filters.append(RepetitionFilter(n_repeat=n_repeat, use_memory=False))
        elif not isinstance(filters, list):
            filters = [filters]
        else:
            filters = list(filters)
        self.ioa_threshold = ioa_threshold
        if ioa_threshold:
            filters.append(self._not_emitted_here)
        self.filters = filters
        for f in filters:
            if isinstance(f, RepetitionFilter):
                self.state.repetition_filter = f
                break
        if callback is not None and not callable(callback):
            if not hasattr(callback, 'put'):
                raise AssertionError('`callback` must be callable or have a put method !')
            callback = callback.put
        self.callback, self.callbacks = callback, callbacks
        self.duplicate_threshold, self.duplicate_tolerance, self.threshold = duplicate_threshold, duplicate_tolerance, threshold
        self.infer_kwargs = infer_kwargs
        self.emitted = []                                           # (frame index, result dict) of everything let through

    def _not_emitted_here(self, boxes, **_):
        """Paragraphs that are not (almost) inside a box emitted since the last new frame (base_ocr.py:561-564)."""
        if self.state.last_boxes is None:
            return np.arange(len(boxes))
        return np.flatnonzero(np.all(ioa_matrix(boxes, self.state.last_boxes) < self.ioa_threshold, axis=1))

    def dice(self, score_map):
        prev = self.state.prev_output
        if isinstance(score_map, torch.Tensor) and score_map.is_cuda:
            return float(self.model.ctx.dice_coeff(score_map, prev).item())
        dev = self.model.ctx.device
        as_dev = lambda m: torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(dev)   # noqa: E731
        return float(self.model.ctx.dice_coeff(as_dev(score_map), as_dev(prev)).item())

    def step(self, frame, frame_index):
        """One frame -> the list of results emitted for it (possibly empty)."""
        st = self.state
        det = self.detector.infer(frame, output_on_device=True)
        boxes = det['boxes']
        score_map = det['output'][:, :, 0]
        n_boxes = len(boxes['boxes']) if isinstance(boxes, dict) else len(boxes)
        if n_boxes == 0:
            st.set_new_frame(score_map)
            st.n_skipped += 1
            return []
        duplicate = False
        if st.prev_output is not None and tuple(st.prev_output.shape) == tuple(score_map.shape):
            duplicate = self.dice(score_map) >= self.duplicate_threshold
        if duplicate:
            st.n_duplicates += 1
            if st.n_duplicates > self.duplicate_tolerance and st.n_skipped % self.duplicate_tolerance < 2:
                st.n_skipped += 1
                return []
        else:
            st.set_new_frame(score_map)
        res = self.model.infer({'image': det['image'], 'boxes': boxes}, combine=True, box_filters=self.filters,
                               **self.infer_kwargs).get('ocr', [])
        found_any = len(res) > 0
        if found_any:
            new_boxes = np.array([r['boxes'] for r in res], dtype=np.float32).reshape(-1, 4)
            st.last_boxes = new_boxes if st.last_boxes is None else np.concatenate([st.last_boxes, new_boxes], axis=0)
            res = [r for r in res if keep_text_result(r, st.emitted_texts, st.last_texts, threshold=self.threshold,
                                                      **self.infer_kwargs)]
        if not res:
            st.n_skipped += 1
            return []
        st.n_skipped = 0
        for r in res:
            st.last_texts.append(r['text'])
            st.emitted_texts.add(r['text'])
            if self.callback:
                self.callback(r['text'])
            self.emitted.append((frame_index, r))
        if self.callbacks:
            from ...utils.saving import apply_callbacks
            apply_callbacks(self.callbacks, {}, {'image': det['image'], 'frame_index': frame_index, 'boxes': boxes,
                                                 'ocr': res})
        return res


def run_stream(model, frames, *, detector='east', detector_kwargs=None, max_frames=None, **kwargs):
    """Runs `OcrStream.step` over `frames`; returns the `OcrStream` (its `.emitted`, `.state`)."""
    if isinstance(frames, (str, int)):
        raise NotImplementedError('stream: cameras, video files and screen mirrors are not opened by this library; '
                                  'pass an iterable of frames (arrays or image file names)')
    if isinstance(detector, str):
        from .. import get_pretrained
        detector = get_pretrained(detector, **(detector_kwargs or {}))
    runner = OcrStream(model, detector, **kwargs)
    for i, frame in enumerate(frames):
        if max_frames is not None and i >= max_frames:
            break
        runner.step(frame, i)
    return runner
