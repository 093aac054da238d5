"""The CPU statement of the rotated extensions (oracle/rotated.py) against known answers; the GPU kernels are
compared with it in tests/test_gpu_rotated.py.  The reference has no rotated IoU / rotated crop to compare with."""
import math

import numpy as np

from oracle import boxes as obox, image as oimg
from oracle.rotated import crop_geometry, prepare_crnn_input_rotated, quad_iou, rectify_gray, rotated_nms


def _rot_rect(cx, cy, w, h, ang):
    c, s = math.cos(ang), math.sin(ang)
    pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    return (pts @ np.array([[c, s], [-s, c]]) + [cx, cy]).astype(np.float32)


def test_rotated_iou_known_answers():
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    assert quad_iou(sq, sq) == 1.0 and quad_iou(sq, sq[::-1]) == 1.0
    assert quad_iou(sq, sq + np.float32([1, 0])) == 0.0 and quad_iou(sq, sq + np.float32([5, 5])) == 0.0
    assert abs(quad_iou(sq, sq + np.float32([0.5, 0])) - 1 / 3) < 1e-12
    assert abs(quad_iou(sq, sq + np.float32([0.5, 0.5])) - 0.25 / 1.75) < 1e-12
    assert abs(quad_iou(sq, sq * np.float32(0.5)) - 0.25) < 1e-12                      # one inside the other
    assert abs(quad_iou(sq, _rot_rect(0.5, 0.5, 1, 1, math.pi / 4)) - 1 / math.sqrt(2)) < 1e-6
    assert quad_iou(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32)) == 0.0  # empty union
    # symmetric, and equal to the axis-aligned IoU for axis-aligned rectangles
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = np.sort(rng.random((2, 2)).astype(np.float32), axis=0)
        b = np.sort(rng.random((2, 2)).astype(np.float32), axis=0)
        qa = np.float32([[a[0, 0], a[0, 1]], [a[1, 0], a[0, 1]], [a[1, 0], a[1, 1]], [a[0, 0], a[1, 1]]])
        qb = np.float32([[b[0, 0], b[0, 1]], [b[1, 0], b[0, 1]], [b[1, 0], b[1, 1]], [b[0, 0], b[1, 1]]])
        want = obox.compute_iou(np.float32([[*a[0], *a[1]]]), np.float32([[*b[0], *b[1]]]), as_matrix=True)[0, 0]
        assert abs(quad_iou(qa, qb) - want) < 1e-6 and abs(quad_iou(qa, qb) - quad_iou(qb, qa)) < 1e-12


def test_rotated_nms_small_case():
    quads = np.stack([_rot_rect(0.3, 0.3, 0.2, 0.05, 0.3), _rot_rect(0.302, 0.301, 0.2, 0.05, 0.31),
                      _rot_rect(0.3, 0.3, 0.2, 0.05, 0.3 + math.pi / 2), _rot_rect(0.7, 0.7, 0.2, 0.05, 0.0)])
    scores = np.float32([0.5, 0.9, 0.8, 0.4])
    # the crossed rectangle shares only a small square with the first two: a rotated IoU keeps it, while the
    # axis-aligned bounds of the three nearly coincide
    assert list(rotated_nms(quads, scores, 0.3)) == [1, 2, 3]
    assert list(rotated_nms(quads, scores, 0.3, max_output_size=2)) == [1, 2]
    assert list(rotated_nms(quads, np.float32([0.5, 0.5, 0.5, 0.5]), 0.3)) == [0, 2, 3]      # ties: lowest index first


def test_unrotated_rectangle_is_the_slice():
    rng = np.random.default_rng(1)
    image = rng.integers(0, 256, (90, 140, 3), dtype=np.uint8)
    x0, y0, x1, y1 = 12, 20, 112, 52
    quad = np.float32([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
    origin, sx, sy, cw, ch = crop_geometry(quad)
    assert (cw, ch) == (x1 - x0, y1 - y0) and list(sx) == [1, 0] and list(sy) == [0, 1]
    gray = rectify_gray(image, origin, sx, sy, cw, ch)
    np.testing.assert_array_equal(gray, oimg.rgb_to_grayscale(oimg.to_float(image[y0:y1, x0:x1]))[..., 0])
    np.testing.assert_array_equal(prepare_crnn_input_rotated(image, quad),
                                  oimg.prepare_crnn_input(image[y0:y1, x0:x1], resize='numpy'))
    # a quarter turn reads the same pixels, transposed: text running downwards
    turned = np.float32([[x1, y0], [x1, y1], [x0, y1], [x0, y0]])
    o2, sx2, sy2, cw2, ch2 = crop_geometry(turned)
    g2 = rectify_gray(image, o2, sx2, sy2, cw2, ch2)
    np.testing.assert_array_equal(g2, gray.T[::-1])
