"""RBOX decode of the EAST output map, restated in NumPy (test infrastructure).

Follows `models/detection/east.py:64-95` (scaling of the raw map) and `:102-187`
(`restore_polys_from_map`, `restore_polys`, `get_rotation_matrix`, `filter_polys`).

Two trigonometry modes:

* `trig='numpy32'` evaluates sin/cos on the float32 angle exactly like the reference does
  (NumPy float32 loops).  Used to pin this file against vectors produced by the reference.
* `trig='rounded64'` evaluates sin/cos in float64 and rounds the result to float32, i.e. the
  correctly rounded float32 value.  The HIP kernel does the same, so the two agree bit for bit;
  it differs from 'numpy32' by at most one float32 ulp of sin/cos.

Output order is row-major over the map (y, then x).  The reference's own order comes from an
unstable argsort over heavily tied keys (east.py:133) and is implementation-defined; compare
against it as a set.
"""
import numpy as np


def split_east_output(output):
    """Raw (h,w,6) sigmoid map -> score, geometry (pixels) and angle, east.py:81-85."""
    output = np.asarray(output, dtype=np.float32)
    score = output[..., 0]
    geo = output[..., 1:5] * 512
    theta = (output[..., 5] - 0.5) * np.pi          # float32 (Python scalar is weak)
    return score, geo.astype(np.float32), theta.astype(np.float32)


def _sincos(theta32, trig):
    if trig == 'numpy32':
        return np.cos(theta32), np.sin(theta32)
    if trig == 'rounded64':
        t = theta32.astype(np.float64)
        return np.cos(t).astype(np.float32), np.sin(t).astype(np.float32)
    raise ValueError(trig)


def restore_polys_from_map(score_map, geo_map, theta_map, input_hw, output_hw, *,
                           threshold=0.35, normalize=True, trig='numpy32', return_index=False):
    """Decode one image.  Returns dict(boxes=(N,4,2) f32, scores=(N,) f32, format='poly')."""
    score_map = np.asarray(score_map, dtype=np.float32)
    geo_map = np.asarray(geo_map, dtype=np.float32)
    theta_map = np.asarray(theta_map, dtype=np.float32)
    if score_map.ndim == 3:
        score_map = score_map[:, :, 0]
    if theta_map.ndim == 3:
        theta_map = theta_map[:, :, 0]
    in_h, in_w = int(input_hw[0]), int(input_hw[1])
    out_h, out_w = int(output_hw[0]), int(output_hw[1])

    ys, xs = np.nonzero(score_map > threshold)            # row-major
    scores = score_map[ys, xs]
    d = geo_map[ys, xs]                                   # (N,4) float32: top, bottom, left, right
    angle = theta_map[ys, xs]

    scale_y, scale_x = in_h // out_h, in_w // out_w       # east.py:155
    x = xs.astype(np.int64) * scale_x
    y = ys.astype(np.int64) * scale_y

    y_min, y_max = y - d[:, 0], y + d[:, 1]               # int64 (op) float32 -> float64
    x_min, x_max = x - d[:, 2], x + d[:, 3]

    c32, s32 = _sincos(-angle, trig)                      # float32
    c, s = c32.astype(np.float64), s32.astype(np.float64)

    tx = np.stack([x_min, x_max, x_max, x_min], axis=1) - x[:, None]
    ty = np.stack([y_min, y_min, y_max, y_max], axis=1) - y[:, None]
    # (N,4,2) @ (N,2,2): px = tx*cos - ty*sin ; py = tx*sin + ty*cos   (east.py:165-174)
    px = tx * c[:, None] + ty * (-s)[:, None] + x[:, None]
    py = tx * s[:, None] + ty * c[:, None] + y[:, None]
    res = np.stack([px, py], axis=-1)                     # (N,4,2) float64

    lim = np.array([in_w, in_h]).reshape(1, 1, 2)
    outside = np.any(res < 0, axis=-1) | np.any(res >= lim, axis=-1)
    keep = np.count_nonzero(outside, axis=-1) <= 1        # east.py:183-187

    res = res[keep]
    if normalize:
        res = (res / lim).astype(np.float32)
    out = {'boxes': res, 'scores': scores[keep], 'format': 'poly'}
    if return_index:
        out['index'] = (ys[keep] * out_w + xs[keep]).astype(np.int32)
    return out


def decode_map(output, input_hw, *, threshold=0.35, trig='numpy32', return_index=False):
    """Raw (h,w,6) map -> decoded polys, exactly what `EAST.decode_output` feeds to `nms`."""
    score, geo, theta = split_east_output(output)
    return restore_polys_from_map(score, geo, theta, input_hw, score.shape[:2],
                                  threshold=threshold, trig=trig, return_index=return_index)
