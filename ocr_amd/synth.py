"""Seeded synthetic inputs for tests and benchmarks: text-rendered pages, the EAST "teacher" maps
that correspond to them, and word crops.

No dataset or trained checkpoint is available offline, so:
  * pages are rendered with PIL (dark pseudo-words on a light noisy background);
  * "teacher" maps are what a trained EAST would ideally output for a page: score ~0.9 inside the
    0.3-shrunk word quadrilateral (0.05 elsewhere), the four edge distances / 512, and the angle
    encoded as theta / pi + 0.5.  Scores are jittered so that no two pixels of a page are equal.
"""
import math

import numpy as np

_CONSONANTS = 'bcdfghjklmnprstvwz'
_VOWELS = 'aeiou'


def make_words(rng, n, min_len=2, max_len=10):
    words = []
    for _ in range(n):
        L = int(rng.integers(min_len, max_len + 1))
        w = ''.join((_CONSONANTS if i % 2 == 0 else _VOWELS)[int(rng.integers(0, 18 if i % 2 == 0 else 5))]
                    for i in range(L))
        if rng.random() < 0.2:
            w = w.capitalize()
        words.append(w)
    return words


def layout_words(rng, H, W, n_words, min_h=16, max_h=32, rotated_frac=0.1, max_angle_deg=10.0):
    """Non-overlapping word boxes laid out on text lines.

    Returns a list of dicts(cx, cy, w, h, angle) in pixels / radians (angle counter-clockwise)."""
    boxes = []
    y = 8
    per_line_target = None
    while y < H - max_h - 8 and len(boxes) < n_words:
        line_h = int(rng.integers(min_h, max_h + 1))
        x = int(rng.integers(6, 30))
        while x < W - 40 and len(boxes) < n_words:
            n_chars = int(rng.integers(2, 11))
            w = int(n_chars * line_h * 0.55) + 4
            if x + w >= W - 6:
                break
            angle = 0.0
            if rng.random() < rotated_frac:
                angle = math.radians(float(rng.uniform(-max_angle_deg, max_angle_deg)))
            boxes.append({'cx': x + w / 2.0, 'cy': y + line_h / 2.0 + 4, 'w': float(w), 'h': float(line_h),
                          'angle': angle, 'chars': n_chars})
            x += w + int(rng.integers(line_h // 2 + 6, line_h + 24))
        y += line_h + int(rng.integers(10, 18))
    return boxes


def render_page(rng, H, W, n_words=200, **layout_kw):
    """-> (image uint8 (H,W,3), list of word dicts with 'text')."""
    from PIL import Image, ImageDraw, ImageFont
    boxes = layout_words(rng, H, W, n_words, **layout_kw)
    base = rng.integers(200, 245)
    img = np.clip(rng.normal(base, 6.0, (H, W, 3)), 0, 255).astype(np.uint8)
    page = Image.fromarray(img)
    fonts = {}
    for b in boxes:
        text = make_words(rng, 1, b['chars'], b['chars'])[0]
        b['text'] = text
        size = max(8, int(b['h'] * 0.9))
        if size not in fonts:
            fonts[size] = ImageFont.load_default(size=size)
        font = fonts[size]
        pw, ph = int(b['w']), int(b['h'])
        patch = Image.new('RGBA', (pw, ph), (0, 0, 0, 0))
        d = ImageDraw.Draw(patch)
        ink = int(rng.integers(0, 70))
        d.text((2, 0), text, font=font, fill=(ink, ink, ink, 255))
        if b['angle'] != 0.0:
            patch = patch.rotate(math.degrees(b['angle']), expand=True, resample=Image.BILINEAR)
        ox = int(round(b['cx'] - patch.size[0] / 2.0))
        oy = int(round(b['cy'] - patch.size[1] / 2.0))
        page.paste(patch, (ox, oy), patch)
    return np.asarray(page.convert('RGB'), dtype=np.uint8).copy(), boxes


def teacher_maps(rng, boxes, H, W, scale=4, shrink=0.3):
    """EAST-style (H/4, W/4, 6) float32 map for the given word boxes (see module docstring)."""
    mh, mw = H // scale, W // scale
    out = np.zeros((mh, mw, 6), dtype=np.float32)
    out[..., 0] = 0.05
    out[..., 5] = 0.5
    for b in boxes:
        th = b['angle']
        e1 = np.array([math.cos(th), -math.sin(th)])       # along the text line
        e2 = np.array([math.sin(th), math.cos(th)])        # down
        a, hb = b['w'] / 2.0, b['h'] / 2.0
        r = math.hypot(a, hb) + scale
        x0, x1 = max(0, int((b['cx'] - r) // scale)), min(mw, int((b['cx'] + r) // scale) + 1)
        y0, y1 = max(0, int((b['cy'] - r) // scale)), min(mh, int((b['cy'] + r) // scale) + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        ys, xs = np.mgrid[y0:y1, x0:x1]
        px, py = xs * scale - b['cx'], ys * scale - b['cy']
        u = px * e1[0] + py * e1[1]
        v = px * e2[0] + py * e2[1]
        inner = min(a, hb) * shrink
        inside = (np.abs(u) <= a - inner) & (np.abs(v) <= hb - inner)
        if not inside.any():
            continue
        sub = out[y0:y1, x0:x1]
        sub[..., 0][inside] = 0.9
        sub[..., 1][inside] = ((v + hb) / 512.0)[inside]      # top
        sub[..., 2][inside] = ((hb - v) / 512.0)[inside]      # bottom
        sub[..., 3][inside] = ((u + a) / 512.0)[inside]       # left
        sub[..., 4][inside] = ((a - u) / 512.0)[inside]       # right
        sub[..., 5][inside] = th / math.pi + 0.5
    # small geometry noise (as a real network would have) and unique scores
    text = out[..., 0] > 0.5
    out[..., 1:5][text] += rng.normal(0, 0.6 / 512.0, (int(text.sum()), 4)).astype(np.float32)
    out[..., 5][text] += rng.normal(0, 0.002, int(text.sum())).astype(np.float32)
    jitter = rng.permutation(mh * mw).reshape(mh, mw).astype(np.float64) / (mh * mw)
    out[..., 0] = np.where(text, 0.85 + 0.1 * jitter, 0.02 + 0.06 * jitter).astype(np.float32)
    np.clip(out[..., 1:5], 0.0, 1.0, out=out[..., 1:5])
    return out.astype(np.float32)


def make_pages(seed, n_pages, H, W, n_words=200, **layout_kw):
    """-> images (n,H,W,3) uint8, maps (n,H/4,W/4,6) float32, list of word lists."""
    rng = np.random.default_rng(seed)
    imgs, maps, words = [], [], []
    for _ in range(n_pages):
        img, boxes = render_page(rng, H, W, n_words, **layout_kw)
        imgs.append(img)
        maps.append(teacher_maps(rng, boxes, H, W))
        words.append(boxes)
    return np.stack(imgs), np.stack(maps), words


def make_crops(seed, n, min_h=12, max_h=96, min_aspect=1.0, max_aspect=8.0):
    """Word-like uint8 crops of random size: list of (h,w,3) arrays."""
    from PIL import Image, ImageDraw, ImageFont
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        h = int(rng.integers(min_h, max_h + 1))
        w = max(4, int(h * rng.uniform(min_aspect, max_aspect)))
        base = rng.integers(190, 245)
        img = Image.fromarray(np.clip(rng.normal(base, 6.0, (h, w, 3)), 0, 255).astype(np.uint8))
        d = ImageDraw.Draw(img)
        font = ImageFont.load_default(size=max(8, int(h * 0.8)))
        n_chars = max(1, int(w / (h * 0.5)))
        d.text((2, 0), make_words(rng, 1, n_chars, n_chars)[0], font=font, fill=(20, 20, 20))
        out.append(np.asarray(img, dtype=np.uint8).copy())
    return out
