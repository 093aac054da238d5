"""Host-side size arithmetic of the image preparation, with the reference's dtypes.

The kernels take sizes as arguments; they are computed here exactly as the reference computes
them (`get_output_size`, utils/image/image_processing.py:144-213; the intermediate size of
`resize_image`, :68-72; `normalize_boxes(..., 'absolute')`, utils/image/bounding_box/converter.py:
186-196; the slice of `crop_box`, processing.py:128-132).
"""
import numpy as np


def east_input_size(h, w, multiple=32):
    """Each side rounded up to the next multiple (the image is stretched, not padded)."""
    def up(v):
        return v if v % multiple == 0 else (v // multiple + 1) * multiple
    return up(int(h)), up(int(w))


def crnn_output_size(crop_h, crop_w, height=64, multiple=64):
    """Padded network input size for a crop: (height, W') with W' = int(w * height / h) rounded up
    to the next multiple.  The ratio is taken in float64 like the reference (int / int division)."""
    ratio = max(float(height) / float(crop_h), -1.0 / float(crop_w))
    w_out = int(np.int32(np.float64(crop_w) * np.float64(ratio)))
    if w_out % multiple != 0:
        w_out = (w_out // multiple + 1) * multiple
    return int(height), int(w_out)


def crnn_intermediate_size(crop_h, crop_w, out_h, out_w):
    """Size the crop is resized to before padding: int32(float32(hw) * min(float32(out) / float32(hw)))."""
    img = np.array([crop_h, crop_w], dtype=np.float32)
    ratio = np.min(np.array([out_h, out_w], dtype=np.float32) / img)
    inter = (img * ratio).astype(np.int32)
    return int(inter[0]), int(inter[1])


def relative_to_absolute(boxes_rel, image_h, image_w):
    """float32 relative xyxy -> int32 absolute, truncation toward zero of the float32 product."""
    boxes_rel = np.asarray(boxes_rel, dtype=np.float32).reshape(-1, 4)
    factors = np.array([image_w, image_h, image_w, image_h], dtype=np.float32)
    return (boxes_rel * factors).astype(np.int32)


def resolve_slices(abs_boxes, image_h, image_w):
    """What `image[y0:y1, x0:x1]` selects for int boxes: (n,4) resolved x0,y0,x1,y1 and a validity
    mask (False = empty crop, which the reference warns about and skips, models/ocr/base_ocr.py:182-184).
    Negative indices wrap around as in NumPy."""
    abs_boxes = np.asarray(abs_boxes).reshape(-1, 4)
    out = np.zeros((len(abs_boxes), 4), dtype=np.int32)
    valid = np.zeros((len(abs_boxes),), dtype=bool)
    for i, (x0, y0, x1, y1) in enumerate(abs_boxes.tolist()):
        xl, xh, _ = slice(x0, x1).indices(image_w)
        yl, yh, _ = slice(y0, y1).indices(image_h)
        out[i] = (xl, yl, max(xl, xh), max(yl, yh))
        valid[i] = (xh > xl) and (yh > yl)
    return out, valid


def _resolve_index(v, size):
    """Vectorised `slice(v, ...).indices(size)` for one bound (step 1): negative wraps once, then clamps."""
    v = np.where(v < 0, v + size, v)
    return np.clip(v, 0, size)


def dezoom_absolute(abs_boxes, factor, image_h, image_w):
    """int32 pixel xyxy boxes scaled about their centres by `factor`, as `crop_box(..., dezoom_factor=)` sees them
    (utils/image/bounding_box/converter.py:202-232 after the conversion to pixels): float32 arithmetic, the new corner
    and size truncated to int32, the size clipped to the image."""
    ab = np.asarray(abs_boxes, dtype=np.int32).reshape(-1, 4)
    x, y = ab[:, 0].astype(np.float32), ab[:, 1].astype(np.float32)
    w, h = (ab[:, 2] - ab[:, 0]).astype(np.float32), (ab[:, 3] - ab[:, 1]).astype(np.float32)
    new_h, new_w = h * factor, w * factor
    nx = np.maximum((x + w / 2.) - new_w / 2., 0).astype(np.int32)
    ny = np.maximum((y + h / 2.) - new_h / 2., 0).astype(np.int32)
    nw = np.minimum(new_w.astype(np.int32), image_w - nx)
    nh = np.minimum(new_h.astype(np.int32), image_h - ny)
    return np.stack([nx, ny, nx + nw, ny + nh], axis=1)


def _network_sizes(ch, cw, height, multiple):
    """Crop sizes (int arrays) -> (padded width, intermediate height, intermediate width, valid) as the reference
    sizes a recogniser input (get_output_size + resize_image, utils/image/image_processing.py:54-79,202-242)."""
    valid = (cw > 0) & (ch > 0)
    chs, cws = np.maximum(ch, 1), np.maximum(cw, 1)
    # get_output_size: ratio = max(height / h, -1 / w) in float64, W' = int32(w * ratio), next multiple
    ratio = np.maximum(np.float64(height) / chs.astype(np.float64), -1.0 / cws.astype(np.float64))
    w_out = (cws.astype(np.float64) * ratio).astype(np.int32).astype(np.int64)
    w_out = np.where(w_out % multiple != 0, (w_out // multiple + 1) * multiple, w_out)
    # resize_image: intermediate = int32(float32(hw) * min(float32(out) / float32(hw)))
    hf, wf = chs.astype(np.float32), cws.astype(np.float32)
    r32 = np.minimum(np.float32(height) / hf, w_out.astype(np.float32) / wf)
    ih, iw = (hf * r32).astype(np.int32), (wf * r32).astype(np.int32)
    valid = valid & (w_out > 0) & (ih > 0) & (iw > 0)
    return w_out, ih, iw, valid


def plan_crops_rotated(quads_rel, image_h, image_w, height=64, multiple=64):
    """`plan_crops` for quadrilaterals (opt-in, not in the reference): quads_rel (m,4,2) relative corners in the
    order top-left, top-right, bottom-right, bottom-left of the text, as the detector's decode writes them.
    The crop is the parallelogram spanned from the top-left corner by the top and the left edge, cw x ch cells with
    cw, ch = the edge lengths in pixels rounded to whole numbers.  Returns dict(geom (m,8) float32 = origin x, y,
    step along the text (x, y), step across (x, y), cw, ch; inter, out_w, index as `plan_crops`)."""
    q = np.asarray(quads_rel, dtype=np.float32).reshape(-1, 4, 2) * np.array([image_w, image_h], dtype=np.float32)
    finite = np.isfinite(q).all(axis=(1, 2))
    q = np.where(finite[:, None, None], q, np.float32(0))
    top, left = q[:, 1] - q[:, 0], q[:, 3] - q[:, 0]
    len_top = np.minimum(np.sqrt(top[:, 0] * top[:, 0] + top[:, 1] * top[:, 1]), np.float32(1e9))
    len_left = np.minimum(np.sqrt(left[:, 0] * left[:, 0] + left[:, 1] * left[:, 1]), np.float32(1e9))
    cw = np.maximum(1, np.rint(len_top)).astype(np.int64)
    ch = np.maximum(1, np.rint(len_left)).astype(np.int64)
    finite &= (cw < 65536) & (ch < 65536)
    cw, ch = np.where(finite, cw, 0), np.where(finite, ch, 0)
    w_out, ih, iw, valid = _network_sizes(ch, cw, height, multiple)
    keep = np.nonzero(valid)[0]
    cwf, chf = np.maximum(cw, 1).astype(np.float32)[:, None], np.maximum(ch, 1).astype(np.float32)[:, None]
    geom = np.concatenate([q[:, 0], top / cwf, left / chf, cwf, chf], axis=1).astype(np.float32)
    return {'geom': geom[keep].reshape(-1, 8), 'inter': np.stack([ih, iw], axis=1)[keep].astype(np.int32).reshape(-1, 2),
            'out_w': w_out[keep].astype(np.int32), 'index': keep.astype(np.int64)}


def plan_crops(boxes_rel, image_h, image_w, height=64, multiple=64, dezoom_factor=1.):
    """Everything the crop kernel needs for the boxes of one image (vectorised over the boxes).

    Returns dict(rects (m,4) int32, inter (m,2) int32, out_w (m,) int32, index (m,) indices of the
    boxes that give a non-empty network input).  `dezoom_factor` enlarges (> 1) or shrinks the pixel boxes
    about their centres before the crop (models/ocr/base_ocr.py:178-181)."""
    ab = relative_to_absolute(boxes_rel, image_h, image_w)
    if dezoom_factor != 1.:
        ab = dezoom_absolute(ab, dezoom_factor, image_h, image_w)
    ab = ab.astype(np.int64)
    x0, x1 = _resolve_index(ab[:, 0], image_w), _resolve_index(ab[:, 2], image_w)
    y0, y1 = _resolve_index(ab[:, 1], image_h), _resolve_index(ab[:, 3], image_h)
    cw, ch = x1 - x0, y1 - y0
    w_out, ih, iw, valid = _network_sizes(ch, cw, height, multiple)
    keep = np.nonzero(valid)[0]
    rects = np.stack([x0, y0, x1, y1], axis=1)[keep].astype(np.int32)
    return {'rects': rects.reshape(-1, 4), 'inter': np.stack([ih, iw], axis=1)[keep].astype(np.int32).reshape(-1, 2),
            'out_w': w_out[keep].astype(np.int32), 'index': keep.astype(np.int64)}


def quantise_u8(image):
    """Pixels for the GPU path, which reads uint8 pages.  Float images hold u/255 (what the reference's loader
    produces, utils/image/image_io.py:126-128): they are ROUNDED to the nearest integer (u/255*255 falls just below u
    for a number of values, so truncation would darken those pixels by one); other integer types are cast."""
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return image
    if image.dtype.kind == 'f':
        return np.clip(np.rint(image.astype(np.float32) * np.float32(255)), 0, 255).astype(np.uint8)
    return image.astype(np.uint8)
