"""Reference-faithful CPU pipeline (test infrastructure and the timed CPU baseline of bench.py).

Does what `BaseOCR.infer(image, combine=False, method='greedy')` does in the reference
(models/ocr/base_ocr.py:146-247 on top of models/detection/base_detector.py:72-81 and
models/detection/east.py:64-95): one EAST forward per image, NumPy decode, LANMS, then ONE
recogniser forward PER TEXT ROW at batch 1, logits padded with 0 and decoded greedily.
"""
import numpy as np

from . import boxes as obox, ctc as octc, east_decode, image as oimg, lanms_c, nets


def detect_page(image_u8, east_params, maps_override=None, obj_threshold=0.35, nms_threshold=0.2,
                merge_threshold=0.3, merge_method='union', trig='rounded64', run_network=True):
    h, w = image_u8.shape[:2]
    maps = None
    if run_network:
        x = oimg.prepare_east_input(image_u8)
        maps = nets.east_forward(east_params, x[None])[0].numpy()
        in_hw = x.shape[:2]
    else:
        in_hw = oimg.east_size(h, w)
    used = maps if maps_override is None else maps_override
    dec = east_decode.decode_map(used, in_hw, threshold=obj_threshold, trig=trig)
    xyxy = obox.poly_to_xyxy(dec['boxes'])
    if len(xyxy) == 0:
        return maps, np.zeros((0, 4), np.float32), np.zeros((0,), np.float32)
    boxes, scores, mask, _, _ = lanms_c.lanms(xyxy, dec['scores'], None, nms_threshold,
                                              merge_threshold=merge_threshold, merge_method=merge_method)
    return maps, boxes[mask], scores[mask]


def recognise_rows(image_u8, boxes_rel, crnn_params, tokenizer, return_logits=False):
    """One forward per row (batch 1), as base_ocr.py:171-201; empty inputs are skipped (:182-184).
    `return_logits` adds the list of per-row (T_i,V) logits as a fourth result."""
    outputs, index = [], []
    for i, box in enumerate(boxes_rel):
        _, crop = obox.crop_box(image_u8, box)
        inp = oimg.prepare_crnn_input(crop)
        if inp is None or any(s == 0 for s in inp.shape):
            continue
        outputs.append(nets.crnn_forward(crnn_params, inp[None])[0].numpy())
        index.append(i)
    if not outputs:
        return ([], [], np.zeros((0,), np.float32)) + (([],) if return_logits else ())
    lengths = np.array([len(o) for o in outputs], dtype=np.int32)
    tmax, v = int(lengths.max()), outputs[0].shape[1]
    padded = np.zeros((len(outputs), tmax, v), dtype=np.float32)      # pad_batch(pad_value=0.)
    for k, o in enumerate(outputs):
        padded[k, :len(o)] = o
    texts, scores = tokenizer.ctc_decode(padded, lengths)
    if return_logits:
        return index, texts, scores, outputs
    return index, texts, scores


def ocr_page(image_u8, east_params, crnn_params, tokenizer, maps_override=None, **kw):
    """`logits`: the per-row (T_i,V) logits the strings were decoded from (for tie checks)."""
    _, boxes, _ = detect_page(image_u8, east_params, maps_override, **kw)
    index, texts, scores, logits = recognise_rows(image_u8, boxes, crnn_params, tokenizer, return_logits=True)
    return {'boxes': boxes, 'index': index, 'texts': texts, 'scores': scores, 'logits': logits}


def ocr_pages_batched(images_u8, east_params, crnn_params, tokenizer, maps_override=None, **kw):
    """The same arithmetic with the batching a CPU implementation would choose (not what the reference does): one EAST
    forward over all pages, one recogniser forward per padded crop width over all rows of all pages.  Used only as the
    "optimised CPU" figure beside the reference-faithful one in bench.py."""
    n = len(images_u8)
    run_network = kw.pop('run_network', True)
    maps = None
    if run_network:
        x = np.stack([oimg.prepare_east_input(im) for im in images_u8])
        maps = nets.east_forward(east_params, x).numpy()
    per_page, rows = [], []
    for b in range(n):
        _, boxes, _ = detect_page(images_u8[b], east_params, None if maps_override is None else maps_override[b],
                                  run_network=False, **kw)
        per_page.append(boxes)
        for i, box in enumerate(boxes):
            _, crop = obox.crop_box(images_u8[b], box)
            inp = oimg.prepare_crnn_input(crop)
            if inp is None or any(s == 0 for s in inp.shape):
                continue
            rows.append((b, i, inp))
    logits = [None] * len(rows)
    by_width = {}
    for k, (_, _, inp) in enumerate(rows):
        by_width.setdefault(inp.shape[1], []).append(k)
    for _, ks in sorted(by_width.items()):
        out = nets.crnn_forward(crnn_params, np.stack([rows[k][2] for k in ks])).numpy()
        for j, k in enumerate(ks):
            logits[k] = out[j]
    results = [{'boxes': per_page[b], 'index': [], 'texts': [], 'scores': []} for b in range(n)]
    if rows:
        lengths = np.array([len(o) for o in logits], dtype=np.int32)
        padded = np.zeros((len(rows), int(lengths.max()), logits[0].shape[1]), dtype=np.float32)
        for k, o in enumerate(logits):
            padded[k, :len(o)] = o
        texts, scores = tokenizer.ctc_decode(padded, lengths)
        for k, (b, i, _) in enumerate(rows):
            results[b]['index'].append(i)
            results[b]['texts'].append(texts[k])
            results[b]['scores'].append(scores[k])
    return results
