"""Batched OCR pipeline on one GPU: pages in, boxes + strings out.

Same stages and default arguments as `BaseOCR.infer(..., combine=False, method='greedy')` on top of
`EAST.infer` in the reference (models/ocr/base_ocr.py:98-255, models/detection/base_detector.py:
51-81, models/detection/east.py:64-95), but every stage runs once per batch of pages instead of
once per image / once per text row:

    EAST forward (B pages) -> RBOX decode -> LANMS + NMS -> [boxes to host: size arithmetic]
    -> crop/resize/pack (all crops of all pages) -> CRNN + CTC greedy per padded width -> strings

Only two small device->host copies per batch: the kept boxes (to compute crop sizes with the
reference's float32/float64 host arithmetic) and the token ids.

`rotated=True` (opt-in, not the reference's behaviour: it collapses quadrilaterals to axis-aligned boxes and crops
axis-aligned slices) keeps the detector's quadrilaterals instead: greedy NMS with rotated IoU on the decoded
candidates, then every kept quadrilateral is rectified by bilinear sampling into the recogniser input; `boxes` of the
results are (m,4,2) corner arrays.
"""
import numpy as np
import torch

from ._lib import OcrHipError
from .engine import HipContext, F32
from .utils import image_sizes as S
from .utils.text import CharTokenizer


class OcrPipeline:
    def __init__(self, east_params, crnn_params, device=0, tokenizer=None, obj_threshold=0.35, nms_threshold=0.2,
                 merge_threshold=0.3, merge_method='union', max_crop_elems=1 << 20, ctx=None, compute_dtype=F32,
                 rotated=False):
        self.ctx = ctx or HipContext(device)
        self.compute_dtype = compute_dtype          # F32 (the reference's precision) or BF16 operands with float32 accumulation
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
        self.rotated = rotated
        self._rec_seq = 0              # recogniser calls enqueued so far
        self._failed_upto = -1         # calls up to this number were in flight when a persistent BiLSTM launch timed out
        self.lstm_fallbacks = 0        # batches recognised again with one launch per time step after such a time-out

    # ---- detection --------------------------------------------------------------------------
    def detect(self, images, maps_override=None):
        """images (B,H,W,3) uint8 on the GPU -> (maps, kept boxes (B,cap,4), scores, counts) on the GPU.

        `maps_override` replaces the network output as the input of decode (used with synthetic
        "teacher" maps when no trained detector weights exist); the network still runs."""
        B, H, W, _ = images.shape
        maps = self.ctx.east_forward(images, compute_dtype=self.compute_dtype)
        used = maps if maps_override is None else maps_override
        if self.rotated:
            dec = self.ctx.east_decode(used, (H, W), self.obj_threshold, with_polys=True)
            kept = self.ctx.rotated_nms(dec['polys'], dec['scores'], dec['counts'], self.nms_threshold, max_out=self.KEEP_CAP)
            return maps, dec, {'boxes': kept['polys'].view(B, self.KEEP_CAP, 8), 'scores': kept['scores'],
                               'counts': kept['counts']}
        dec = self.ctx.east_decode(used, (H, W), self.obj_threshold)
        kept = self.ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'], merge_threshold=self.merge_threshold,
                              nms_threshold=self.nms_threshold, merge_method=self.merge_method)
        return maps, dec, kept

    # ---- the three stages of a batch; each one only enqueues GPU work and small async copies ------
    KEEP_CAP = 2048          # kept boxes per page copied to the host in the first shot

    def _stage_detect(self, images, maps_override=None):
        """Enqueue detection and the device->host copy of the kept boxes (pinned memory + event)."""
        B = images.shape[0]
        _, _, kept = self.detect(images, maps_override)
        k = min(self.KEEP_CAP, kept['boxes'].shape[1])
        h_boxes = torch.empty((B, k, kept['boxes'].shape[2]), dtype=torch.float32, pin_memory=True)
        h_counts = torch.empty((B,), dtype=torch.int32, pin_memory=True)
        h_boxes.copy_(kept['boxes'][:, :k], non_blocking=True)
        h_counts.copy_(kept['counts'], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {'images': images, 'kept': kept, 'h_boxes': h_boxes, 'h_counts': h_counts, 'boxes_ready': ev}

    def _stage_recognise(self, st):
        """Wait for the boxes, do the host size arithmetic, enqueue crop + recogniser + token copy."""
        st['boxes_ready'].synchronize()
        images = st['images']
        B, H, W, _ = images.shape
        counts = st['h_counts'].numpy()
        if int(counts.max(initial=0)) > st['h_boxes'].shape[1]:          # rare: more kept boxes than KEEP_CAP
            boxes = st['kept']['boxes'][:, :int(counts.max())].cpu().numpy()
        else:
            boxes = st['h_boxes'].numpy()
        rotated = boxes.shape[2] == 8
        per_image = [boxes[b, :int(counts[b])].copy() for b in range(B)]
        if rotated:
            per_image = [bx.reshape(-1, 4, 2) for bx in per_image]
        st['per_image'] = per_image
        st['kept'] = None
        if rotated:
            plans = [S.plan_crops_rotated(bx, H, W) for bx in per_image]
            # the kernel reads 8 floats per crop where the axis-aligned one reads 4 integers: same buffer, bits kept
            rects = np.concatenate([p['geom'] for p in plans]).view(np.int32)
        else:
            plans = [S.plan_crops(bx, H, W) for bx in per_image]
            rects = np.concatenate([p['rects'] for p in plans])
        inter = np.concatenate([p['inter'] for p in plans])
        out_w = np.concatenate([p['out_w'] for p in plans])
        box_idx = np.concatenate([p['index'] for p in plans])
        box_img = np.concatenate([np.full(len(p['index']), b, dtype=np.int32) for b, p in enumerate(plans)])
        n = len(rects)
        st['n'] = n
        if n == 0:
            return st
        # crops sorted by padded width: the recogniser runs its feature extractor once per width and its
        # recurrent / dense / CTC part once over all sequences
        order = np.argsort(out_w, kind='stable')
        rects, inter, out_w, box_img, box_idx = rects[order], inter[order], out_w[order], box_img[order], box_idx[order]
        off = np.concatenate([[0], np.cumsum(64 * out_w.astype(np.int64))])
        dev = images.device
        h_meta = torch.from_numpy(np.concatenate([rects.reshape(-1), inter.reshape(-1), out_w, box_img]).astype(np.int32)).pin_memory()
        h_off = torch.from_numpy(off[:-1].astype(np.int64)).pin_memory()
        meta = h_meta.to(dev, non_blocking=True)
        d_off = h_off.to(dev, non_blocking=True)
        r = rects.shape[1]                                   # 4 integers or 8 floats per crop
        d_rects, d_inter = meta[:r * n].view(n, r), meta[r * n:(r + 2) * n].view(n, 2)
        d_out_w, d_box_img = meta[(r + 2) * n:(r + 3) * n], meta[(r + 3) * n:(r + 4) * n]
        if rotated:
            packed = self.ctx.crop_rotated_pack(images, d_rects.view(torch.float32), d_box_img, d_inter, d_out_w, d_off,
                                                int(off[-1]))
        else:
            packed = self.ctx.crop_resize_pack(images, d_rects, d_box_img, d_inter, d_out_w, d_off, int(off[-1]))
        out = self.ctx.crnn_forward_packed(packed, out_w, compute_dtype=self.compute_dtype)
        st['seq'] = self._rec_seq
        self._rec_seq += 1
        st['out_w'] = out_w
        h_tokens = torch.empty(out['tokens'].shape, dtype=torch.int32, pin_memory=True)
        h_scores = torch.empty(out['scores'].shape, dtype=torch.float32, pin_memory=True)
        h_tokens.copy_(out['tokens'], non_blocking=True)
        h_scores.copy_(out['scores'], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st.update(h_tokens=h_tokens, h_scores=h_scores, tokens_ready=ev, box_img=box_img, box_idx=box_idx,
                  _keep=(h_meta, h_off, meta, d_off, packed, out))
        return st

    def _stage_strings(self, st):
        """Wait for the tokens and turn them into strings."""
        per_image = st['per_image']
        B = len(per_image)
        out = [{'boxes': per_image[b], 'texts': [''] * len(per_image[b]), 'scores': [None] * len(per_image[b]),
                'valid': np.zeros(len(per_image[b]), dtype=bool)} for b in range(B)]
        if st['n'] == 0:
            return out
        st['tokens_ready'].synchronize()
        if st['seq'] > self._failed_upto:
            try:
                self.ctx.check_errors(synchronize=False)    # a timed-out recurrence fails the batch instead of returning garbage
            except OcrHipError as e:
                if 'lstm hand-over timed out' not in str(e):
                    raise
                # the persistent recurrence did not get the whole GPU (a co-tenant): every recogniser call enqueued so
                # far counts as failed; each is run again, once, with one launch per time step (no hand-over between
                # workgroups), in this process
                self._failed_upto = self._rec_seq - 1
        if st['seq'] <= self._failed_upto:
            self._recognise_again(st)
        tokens, scores = st['h_tokens'].numpy(), st['h_scores'].numpy()
        box_img, box_idx = st['box_img'], st['box_idx']
        texts = self.tokenizer.decode_batch(tokens[:st['n']])
        for k, (b, i, text, score) in enumerate(zip(box_img.tolist(), box_idx.tolist(), texts, scores[:st['n']].tolist())):
            o = out[b]
            o['texts'][i] = text
            o['scores'][i] = score
            o['valid'][i] = True
        return out

    def _recognise_again(self, st):
        """The recogniser of a batch once more on the crops still packed on the GPU, `lstm_persistent = 0`."""
        packed = st['_keep'][4]
        before = self.ctx.options.get('lstm_persistent', 2)
        self.ctx.set_option('lstm_persistent', 0)
        try:
            out = self.ctx.crnn_forward_packed(packed, st['out_w'], compute_dtype=self.compute_dtype)
            st['h_tokens'].copy_(out['tokens'])
            st['h_scores'].copy_(out['scores'])
            self.ctx.check_errors(synchronize=True)
        finally:
            self.ctx.set_option('lstm_persistent', before)
        self.lstm_fallbacks += 1

    # ---- end to end -------------------------------------------------------------------------
    def __call__(self, images, maps_override=None):
        """images (B,H,W,3) uint8 GPU -> list (per page) of dict(boxes (m,4) f32, texts, scores, valid)."""
        return self._stage_strings(self._stage_recognise(self._stage_detect(images, maps_override)))

    def run_batches(self, batches):
        """Streams batches through the three stages, software-pipelined: while the GPU works on the
        detection of batch i+1 the host does the size arithmetic of batch i and decodes the strings of
        batch i-1, so neither side waits for the other.  `batches` yields (images, maps_override);
        results are yielded in order."""
        detected, recognised = None, None
        for images, maps_override in batches:
            nxt = self._stage_detect(images, maps_override)
            # the recognition of the batch detected one turn ago is enqueued before the host turns to the strings of the
            # batch before it: the GPU queue never runs dry, also not at the end of the stream
            newer = self._stage_recognise(detected) if detected is not None else None
            if recognised is not None:
                yield self._stage_strings(recognised)
            recognised = newer
            detected = nxt
        newer = self._stage_recognise(detected) if detected is not None else None
        if recognised is not None:
            yield self._stage_strings(recognised)
        if newer is not None:
            yield self._stage_strings(newer)

    def recognise(self, images, boxes_per_image):
        """images (B,H,W,3) uint8 GPU; boxes_per_image: list of (m_b,4) float32 relative xyxy (host)
        -> per image: dict(boxes, texts, scores, valid)."""
        B = images.shape[0]
        k = max(1, max(len(b) for b in boxes_per_image))
        h_boxes = torch.zeros((B, k, 4), dtype=torch.float32)
        for b, bx in enumerate(boxes_per_image):
            h_boxes[b, :len(bx)] = torch.from_numpy(np.ascontiguousarray(bx, dtype=np.float32))
        ev = torch.cuda.Event()
        ev.record()
        st = {'images': images, 'kept': None, 'h_boxes': h_boxes, 'boxes_ready': ev,
              'h_counts': torch.tensor([len(b) for b in boxes_per_image], dtype=torch.int32)}
        return self._stage_strings(self._stage_recognise(st))
