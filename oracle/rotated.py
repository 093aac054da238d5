"""Opt-in extensions that keep the rotation of the detected quadrilaterals (test infrastructure): rotated IoU,
greedy NMS on quadrilaterals, and the crop of an oriented rectangle.

PARITY UNPINNED: the reference has none of this - it collapses quadrilaterals to axis-aligned boxes before NMS
(utils/image/bounding_box/non_max_suppression.py:27, converter.py:151-154) and crops axis-aligned slices
(utils/image/bounding_box/processing.py:111-132, rotation is `NotImplementedError` in converter.py:204).  This file
states the arithmetic the GPU kernels are held to, operation by operation; it is pinned only by the known answers
in the tests (unit squares, shifted and rotated squares) and by the property that an unrotated rectangle on integer
corners gives exactly the reference's axis-aligned crop.

All polygon arithmetic is float64 in the order written here (the kernels are compiled without multiply-add
contraction); the image arithmetic is float32.
"""
import numpy as np

from . import image as oimg

GRAY_WEIGHTS = oimg.GRAY_WEIGHTS


# ---- rotated IoU ------------------------------------------------------------------------------------------------

def _shoelace2(pts):
    """Twice the signed area of a polygon given as a list of (x, y), summed in vertex order."""
    s = 0.0
    n = len(pts)
    for i in range(n):
        x0, y0 = pts[i]
        x1, y1 = pts[(i + 1) % n]
        s = s + (x0 * y1 - x1 * y0)
    return s


def _oriented(quad):
    """The 4 corners as float64 tuples, reversed when their signed area is negative."""
    pts = [(float(np.float64(x)), float(np.float64(y))) for x, y in np.asarray(quad, dtype=np.float32).reshape(4, 2)]
    if _shoelace2(pts) < 0.0:
        pts = pts[::-1]
    return pts


def _clip(subject, a, b):
    """Sutherland-Hodgman: the part of `subject` on the left of (or on) the line a -> b."""
    out = []
    ux, uy = b[0] - a[0], b[1] - a[1]
    n = len(subject)
    for i in range(n):
        s, e = subject[i], subject[(i + 1) % n]
        ds = ux * (s[1] - a[1]) - uy * (s[0] - a[0])
        de = ux * (e[1] - a[1]) - uy * (e[0] - a[0])
        if de >= 0.0:
            if ds < 0.0:
                t = ds / (ds - de)
                out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
            out.append(e)
        elif ds >= 0.0:
            t = ds / (ds - de)
            out.append((s[0] + t * (e[0] - s[0]), s[1] + t * (e[1] - s[1])))
    return out


def quad_iou(p, q):
    """IoU of two convex quadrilaterals ((4,2) or 8 numbers each, any orientation); 0 when the union is empty."""
    P, Q = _oriented(p), _oriented(q)
    area_p, area_q = 0.5 * _shoelace2(P), 0.5 * _shoelace2(Q)
    poly = P
    for i in range(4):
        if not poly:
            break
        poly = _clip(poly, Q[i], Q[(i + 1) % 4])
    inter = 0.5 * _shoelace2(poly) if len(poly) >= 3 else 0.0
    if inter < 0.0:
        inter = 0.0
    union = area_p + area_q - inter
    return inter / union if union > 0.0 else 0.0


def quad_aabb(quads):
    """(n,4) float32 [xmin, ymin, xmax, ymax] of (n,4,2) quadrilaterals."""
    q = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2)
    return np.concatenate([q.min(axis=1), q.max(axis=1)], axis=1)


def rotated_nms(quads, scores, iou_threshold, max_output_size=None):
    """Greedy NMS with rotated IoU: repeatedly takes the remaining quadrilateral with the highest score (the
    lowest index among equals) and removes every remaining one whose IoU with it is >= `iou_threshold`.
    Pairs whose axis-aligned bounds do not overlap (strictly) are not tested: their IoU is 0.
    -> indices of the kept quadrilaterals, in the order they were taken."""
    q = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2)
    s = np.asarray(scores, dtype=np.float32).copy()
    bb = quad_aabb(q)
    alive = np.ones(len(q), dtype=bool) & ~np.isnan(s)
    kept = []
    while alive.any() and (max_output_size is None or len(kept) < max_output_size):
        cand = np.flatnonzero(alive)
        w = int(cand[np.argmax(s[cand])])            # argmax returns the first of equal values
        kept.append(w)
        alive[w] = False
        near = np.flatnonzero(alive & (np.maximum(bb[:, 0], bb[w, 0]) < np.minimum(bb[:, 2], bb[w, 2])) &
                              (np.maximum(bb[:, 1], bb[w, 1]) < np.minimum(bb[:, 3], bb[w, 3])))
        for i in near:
            if quad_iou(q[i], q[w]) >= iou_threshold:
                alive[i] = False
    return np.array(kept, dtype=np.int64)


# ---- crop of an oriented rectangle ---------------------------------------------------------------------------

def crop_geometry(quad_abs):
    """Quadrilateral in pixels, corners in the order top-left, top-right, bottom-right, bottom-left of the TEXT
    -> (origin (2,), step along the text (2,), step across the text (2,), cw, ch): the crop is the parallelogram
    spanned from the top-left corner by the top and the left edge, sampled on a cw x ch grid whose cell (x, y) has
    its centre at origin + (x + 0.5) * step_x + (y + 0.5) * step_y.  cw, ch = the edge lengths rounded to whole
    pixels (at least 1).  float32 arithmetic."""
    q = np.asarray(quad_abs, dtype=np.float32).reshape(4, 2)
    top, left = q[1] - q[0], q[3] - q[0]
    cw = max(1, int(np.rint(np.sqrt(top[0] * top[0] + top[1] * top[1]))))
    ch = max(1, int(np.rint(np.sqrt(left[0] * left[0] + left[1] * left[1]))))
    return q[0].copy(), (top / np.float32(cw)).astype(np.float32), (left / np.float32(ch)).astype(np.float32), cw, ch


def rectify_gray(image_u8, origin, step_x, step_y, cw, ch):
    """Gray values (ch, cw) float32 of the oriented rectangle: bilinear interpolation of the image's gray values
    (u8 / 255, weighted R, G, B sum) at the cell centres, neighbours clamped to the image."""
    H, W = image_u8.shape[:2]
    gray = oimg.rgb_to_grayscale(oimg.to_float(image_u8))[..., 0]
    f32 = np.float32
    xs = (np.arange(cw, dtype=np.float32) + f32(0.5))[None, :]
    ys = (np.arange(ch, dtype=np.float32) + f32(0.5))[:, None]
    px = (f32(origin[0]) + xs * f32(step_x[0])) + ys * f32(step_y[0])
    py = (f32(origin[1]) + xs * f32(step_x[1])) + ys * f32(step_y[1])
    sx, sy = px - f32(0.5), py - f32(0.5)
    fx0, fy0 = np.floor(sx), np.floor(sy)
    tx, ty = (sx - fx0).astype(np.float32), (sy - fy0).astype(np.float32)
    x0 = np.clip(fx0.astype(np.int64), 0, W - 1)
    x1 = np.clip(fx0.astype(np.int64) + 1, 0, W - 1)
    y0 = np.clip(fy0.astype(np.int64), 0, H - 1)
    y1 = np.clip(fy0.astype(np.int64) + 1, 0, H - 1)
    top = gray[y0, x0] * (f32(1) - tx) + gray[y0, x1] * tx
    bot = gray[y1, x0] * (f32(1) - tx) + gray[y1, x1] * tx
    return (top * (f32(1) - ty) + bot * ty).astype(np.float32)


def prepare_crnn_input_rotated(image_u8, quad_abs, height=64, multiple=64):
    """Oriented rectangle of a page -> (64, Wpad, 1) float32 recogniser input (None for an empty one): the
    rectangle is rectified by `rectify_gray`, then resized, padded and normalised exactly like an axis-aligned
    crop (`oracle.image.prepare_crnn_input`)."""
    origin, step_x, step_y, cw, ch = crop_geometry(quad_abs)
    (ih, iw), (oh, ow) = oimg.crnn_sizes(ch, cw, height, multiple)
    if ih <= 0 or iw <= 0 or ow <= 0:
        return None
    img = rectify_gray(image_u8, origin, step_x, step_y, cw, ch)[..., None]
    if (ih, iw) != img.shape[:2]:
        img = oimg.aa_resize_numpy(img, ih, iw)
    img = oimg.pad_repeat_last(img, oh, ow)
    return oimg.normalize_mean_std(img)
