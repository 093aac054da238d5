"""Image preparation of the reference restated on CPU (test infrastructure).

Follows `utils/image/image_io.py:122-135` (crop -> dtype -> gray -> resize),
`utils/image/image_processing.py:17-213` (`resize_image`, `pad_image`, `get_output_size`),
`utils/image/image_normalization.py:28-39,57-58` and `models/ocr/crnn.py:39-45` (CRNN resize
arguments).  Keras' `image.resize(bilinear, antialias=True)` is restated as
`torch.nn.functional.interpolate(mode='bilinear', antialias=True, align_corners=False)`, which is
what Keras' torch backend executes; `aa_resize_numpy` spells the same filter out in NumPy
float32.  The reference has no test of the resized *values*: parity unpinned for them (shapes and
padding are pinned by tests/test_utils_image.py:44-102).
"""
import numpy as np

GRAY_WEIGHTS = (0.2989, 0.5870, 0.1140)      # keras.ops.image.rgb_to_grayscale


def get_output_size(img_hw, size=None, multiples=None, preserve_aspect_ratio=False):
    """`get_output_size` (image_processing.py:144-213) without `round` / `max_shape`."""
    img_size = np.array(img_hw, dtype=np.int32)
    if isinstance(size, tuple):
        if size[0] and size[1]:
            return np.array(size, dtype=np.int32)
        elif size[0] or size[1]:
            size = (size[0] or -1, size[1] or -1)
        else:
            size = None
    if size is None:
        out_size = img_size
    else:
        out_size = size = np.array(size, dtype=np.int32)
    if np.any(out_size == -1):
        if not preserve_aspect_ratio:
            out_size = np.where(out_size != -1, out_size, img_size)
        else:
            ratio = np.max(out_size / img_size)                       # float64
            out_size = (img_size.astype(ratio.dtype) * ratio).astype(np.int32)
    if multiples is not None:
        multiples = np.array(multiples, dtype=np.int32)
        new_values = (out_size // multiples + 1) * multiples
        out_size = np.where(out_size % multiples != 0, new_values, out_size)
    if size is not None:
        out_size = np.where(size != -1, size, out_size)
    return out_size.astype(np.int32)


def intermediate_size(img_hw, output_size):
    """Size handed to the resize op when `preserve_aspect_ratio` (image_processing.py:68-72).

    All in float32: `int32(float32(hw) * min(float32(out) / float32(hw)))`.
    """
    img = np.array(img_hw, dtype=np.float32)
    ratio = np.min(np.asarray(output_size).astype(np.float32) / img)
    return (img * ratio).astype(np.int32)


def crnn_sizes(crop_h, crop_w, height=64, multiple=64):
    """(intermediate (h,w), padded output (H,W)) for one crop under the CRNN resize arguments."""
    out = get_output_size((crop_h, crop_w), size=(height, None), multiples=multiple,
                          preserve_aspect_ratio=True)
    inter = intermediate_size((crop_h, crop_w), out)
    return (int(inter[0]), int(inter[1])), (int(out[0]), int(out[1]))


def east_size(h, w, multiple=32):
    """EAST network input size: each side rounded up to the next multiple (stretch, no pad)."""
    out = get_output_size((h, w), size=None, multiples=multiple)
    return int(out[0]), int(out[1])


def to_float(image_u8):
    """`convert_data_dtype(uint8 -> float32)`: x / 255 (utils/keras/ops/core.py:271)."""
    return (image_u8 / 255).astype(np.float32)


def rgb_to_grayscale(image_f32):
    """Weighted channel sum, float32, order R, G, B."""
    w = np.array(GRAY_WEIGHTS, dtype=np.float32)
    img = np.asarray(image_f32, dtype=np.float32)
    g = img[..., 0] * w[0]
    g = g + img[..., 1] * w[1]
    g = g + img[..., 2] * w[2]
    return g[..., None].astype(np.float32)


def aa_weights(in_size, out_size):
    """Antialiased triangle-filter taps in float32 (one output coordinate per row).

    Returns (xmin int array, list of float32 weight arrays).  The filter support widens with the
    down-scaling factor; when up-scaling it is the plain 2-tap bilinear kernel.
    """
    f32, f64 = np.float32, np.float64
    scale = f32(in_size) / f32(out_size)
    support = scale if scale >= 1.0 else f32(1.0)
    invscale = f32(1.0 / f64(scale)) if scale >= 1.0 else f32(1.0)
    xmins, weights = [], []
    for i in range(out_size):
        # float32 values, but the "+ 0.5" of the bounds and of the filter argument is done in float64
        center = f32(f64(scale) * (i + 0.5))
        xmin = max(int(f64(f32(center - support)) + 0.5), 0)
        xsize = min(int(f64(f32(center + support)) + 0.5), in_size) - xmin
        ws = np.zeros(xsize, dtype=np.float32)
        total = f32(0.0)
        for j in range(xsize):
            x = abs(f32((f64(f32(f32(j + xmin) - center)) + 0.5) * f64(invscale)))
            w = f32(1.0) - x if x < 1.0 else f32(0.0)
            ws[j] = w
            total = total + w
        if total != 0:
            ws = ws / total
        xmins.append(xmin)
        weights.append(ws.astype(np.float32))
    return np.array(xmins, dtype=np.int64), weights


def aa_resize_numpy(image, out_h, out_w):
    """Separable antialiased bilinear resize of an (H,W,C) float32 image: width first, then height."""
    img = np.asarray(image, dtype=np.float32)
    h, w, c = img.shape
    if w != out_w:
        xm, ws = aa_weights(w, out_w)
        tmp = np.zeros((h, out_w, c), dtype=np.float32)
        for i in range(out_w):
            acc = img[:, xm[i], :] * ws[i][0]
            for j in range(1, len(ws[i])):
                acc = acc + img[:, xm[i] + j, :] * ws[i][j]
            tmp[:, i, :] = acc
        img = tmp
    if h != out_h:
        ym, ws = aa_weights(h, out_h)
        tmp = np.zeros((out_h, img.shape[1], c), dtype=np.float32)
        for i in range(out_h):
            acc = img[ym[i]] * ws[i][0]
            for j in range(1, len(ws[i])):
                acc = acc + img[ym[i] + j] * ws[i][j]
            tmp[i] = acc
        img = tmp
    return img


def aa_resize_torch(image, out_h, out_w):
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(x, size=(int(out_h), int(out_w)), mode='bilinear', antialias=True,
                      align_corners=False)
    return y[0].permute(1, 2, 0).contiguous().numpy()


def pad_repeat_last(image, out_h, out_w):
    """`pad_image(pad_mode='repeat_last')`: right columns first, then bottom rows (:125-134)."""
    h, w = image.shape[:2]
    if out_w > w:
        image = np.concatenate([image, np.tile(image[:, -1:, :], (1, out_w - w, 1))], axis=1)
    if out_h > h:
        image = np.concatenate([image, np.tile(image[-1:, :, :], (out_h - h, 1, 1))], axis=0)
    return image


def normalize_mean_std(image, mean=0.5, std=0.5):
    """'east' and 'easyocr' styles: (x - 0.5) / 0.5 in float32 (image_normalization.py:28-39)."""
    return ((np.asarray(image, dtype=np.float32) - np.float32(mean)) / np.float32(std)).astype(np.float32)


def prepare_east_input(image_u8, resize='torch'):
    """uint8 (H,W,3) -> float32 (H',W',3) network input (base_image_model.py:93-131)."""
    img = to_float(image_u8)
    h, w = img.shape[:2]
    oh, ow = east_size(h, w)
    if (oh, ow) != (h, w):
        img = aa_resize_torch(img, oh, ow) if resize == 'torch' else aa_resize_numpy(img, oh, ow)
    return normalize_mean_std(img)


def prepare_crnn_input(crop_u8, resize='torch', height=64, multiple=64):
    """uint8 crop (h,w,3) -> float32 (64, Wpad, 1) CRNN input, or None for an empty crop."""
    if crop_u8.shape[0] == 0 or crop_u8.shape[1] == 0:
        return None
    img = to_float(crop_u8)
    if img.shape[2] == 3:
        img = rgb_to_grayscale(img)
    (ih, iw), (oh, ow) = crnn_sizes(img.shape[0], img.shape[1], height, multiple)
    if ih <= 0 or iw <= 0:
        return None
    if (ih, iw) != img.shape[:2]:
        img = aa_resize_torch(img, ih, iw) if resize == 'torch' else aa_resize_numpy(img, ih, iw)
    img = pad_repeat_last(img, oh, ow)
    return normalize_mean_std(img)
