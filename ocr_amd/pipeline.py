"""Batched OCR pipeline on one GPU: pages in, boxes + strings out.

Same stages and default arguments as `BaseOCR.infer(..., combine=False, method='greedy')` on top of
`EAST.infer` in the reference (models/ocr/base_ocr.py:98-255, models/detection/base_detector.py:
51-81, models/detection/east.py:64-95), but every stage runs once per batch of pages instead of
once per image / once per text row:

    EAST forward (B pages) -> RBOX decode -> LANMS + NMS -> [boxes to host: size arithmetic]
    -> crop/resize/pack (all crops of all pages) -> CRNN + CTC greedy per padded width -> strings

Only two small device->host copies per batch: the kept boxes (to compute crop sizes with the
reference's float32/float64 host arithmetic) and the token ids.
"""
import numpy as np
import torch

from .engine import HipContext, F32
from .utils import image_sizes as S
from .utils.text import CharTokenizer


class OcrPipeline:
    def __init__(self, east_params, crnn_params, device=0, tokenizer=None, obj_threshold=0.35, nms_threshold=0.2,
                 merge_threshold=0.3, merge_method='union', max_crop_elems=1 << 20, ctx=None):
        self.ctx = ctx or HipContext(device)
        self.device = self.ctx.device
        if east_params is not None:
            self.ctx.east_load_weights(east_params)
        if crnn_params is not None:
            self.ctx.crnn_load_weights(crnn_params)
        self.tokenizer = tokenizer or CharTokenizer()
        self.obj_threshold = obj_threshold
        self.nms_threshold = nms_threshold
        self.merge_threshold = merge_threshold
        self.merge_method = merge_method
        self.max_crop_elems = max_crop_elems

    # ---- detection --------------------------------------------------------------------------
    def detect(self, images, maps_override=None):
        """images (B,H,W,3) uint8 on the GPU -> (maps, kept boxes (B,cap,4), scores, counts) on the GPU.

        `maps_override` replaces the network output as the input of decode (used with synthetic
        "teacher" maps when no trained detector weights exist); the network still runs."""
        B, H, W, _ = images.shape
        maps = self.ctx.east_forward(images)
        used = maps if maps_override is None else maps_override
        dec = self.ctx.east_decode(used, (H, W), self.obj_threshold)
        kept = self.ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], merge_threshold=self.merge_threshold,
                              nms_threshold=self.nms_threshold, merge_method=self.merge_method)
        return maps, dec, kept

    # ---- recognition ------------------------------------------------------------------------
    def recognise(self, images, boxes_per_image):
        """images (B,H,W,3) uint8 GPU; boxes_per_image: list of (m_b,4) float32 relative xyxy (host).

        Returns per image: list of dict(box, text, score) for the boxes that give a valid crop."""
        B, H, W, _ = images.shape
        rects, inter, out_w, box_img, box_idx = [], [], [], [], []
        for b, boxes in enumerate(boxes_per_image):
            plan = S.plan_crops(boxes, H, W)
            rects.append(plan['rects']), inter.append(plan['inter']), out_w.append(plan['out_w'])
            box_img.append(np.full(len(plan['index']), b, dtype=np.int32))
            box_idx.append(plan['index'])
        rects, inter, out_w = np.concatenate(rects), np.concatenate(inter), np.concatenate(out_w)
        box_img, box_idx = np.concatenate(box_img), np.concatenate(box_idx)
        results = [[] for _ in range(B)]
        n = len(rects)
        if n == 0:
            return results
        # crops grouped by padded width: each group is one recogniser batch
        order = np.argsort(out_w, kind='stable')
        rects, inter, out_w, box_img, box_idx = rects[order], inter[order], out_w[order], box_img[order], box_idx[order]
        off = np.concatenate([[0], np.cumsum(64 * out_w.astype(np.int64))])
        dev = images.device
        meta = torch.from_numpy(np.concatenate([rects.reshape(-1), inter.reshape(-1), out_w, box_img]).astype(np.int32)).to(dev)
        d_rects, d_inter = meta[:4 * n].view(n, 4), meta[4 * n:6 * n].view(n, 2)
        d_out_w, d_box_img = meta[6 * n:7 * n], meta[7 * n:8 * n]
        d_off = torch.from_numpy(off[:-1].astype(np.int64)).to(dev)
        packed = self.ctx.crop_resize_pack(images, d_rects, d_box_img, d_inter, d_out_w, d_off, int(off[-1]))
        tokens_out, scores_out = [], []
        widths, starts = np.unique(out_w, return_index=True)
        bounds = list(starts) + [n]
        for gi, wpad in enumerate(widths):
            lo, hi = bounds[gi], bounds[gi + 1]
            step = max(1, self.max_crop_elems // int(wpad))
            for c0 in range(lo, hi, step):
                c1 = min(hi, c0 + step)
                crops = packed[off[c0]:off[c1]].view(c1 - c0, 64, int(wpad))
                out = self.ctx.crnn_forward(crops, want_logits=False, want_tokens=True, compute_dtype=F32)
                tokens_out.append(out['tokens']), scores_out.append(out['scores'])
        scores = torch.cat(scores_out).cpu().numpy()
        tokens = [t.cpu().numpy() for t in tokens_out]
        k = 0
        for tk in tokens:
            for row in tk:
                b = int(box_img[k])
                results[b].append({'box_index': int(box_idx[k]), 'text': self.tokenizer.decode(row),
                                   'score': float(scores[k])})
                k += 1
        for b in range(B):
            results[b].sort(key=lambda r: r['box_index'])
        return results

    # ---- end to end -------------------------------------------------------------------------
    def __call__(self, images, maps_override=None):
        """images (B,H,W,3) uint8 GPU -> list (per page) of dict(boxes (m,4) f32, texts, scores)."""
        B = images.shape[0]
        _, _, kept = self.detect(images, maps_override)
        counts = kept['counts'].cpu().numpy()
        mx = int(counts.max()) if B else 0
        boxes = kept['boxes'][:, :max(mx, 1)].cpu().numpy()
        per_image = [boxes[b, :int(counts[b])] for b in range(B)]
        rec = self.recognise(images, per_image)
        out = []
        for b in range(B):
            texts = [''] * len(per_image[b])
            scores = [None] * len(per_image[b])
            valid = np.zeros(len(per_image[b]), dtype=bool)
            for r in rec[b]:
                texts[r['box_index']] = r['text']
                scores[r['box_index']] = r['score']
                valid[r['box_index']] = True
            out.append({'boxes': per_image[b], 'texts': texts, 'scores': scores, 'valid': valid})
        return out
