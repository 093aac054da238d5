"""Batched OCR pipeline on one GPU: pages in, boxes + strings out.

Same stages as `BaseOCR.infer` on top of `EAST.infer` in the reference (models/ocr/base_ocr.py:98-255,
models/detection/base_detector.py:51-81, models/detection/east.py:64-95), but every stage runs once per
batch of pages instead of once per image / once per text row:

    EAST forward (B pages) -> RBOX decode -> LANMS + NMS -> [boxes to host: size arithmetic, grouping]
    -> crop/resize/pack (all crops of all pages) -> CRNN -> CTC decoding on the GPU -> strings

The default is the fast configuration `combine=False, method='greedy'` (one crop per detected word, greedy
decoding fused into the class layer).  `combine=True, method='beam_search', num_beams=10` are the reference's own
defaults (base_ocr.py:112-115,154-157): the words of a page are grouped into lines and paragraphs on the host
(utils/combination.py), every LINE is one recogniser input, the texts of a paragraph are joined with ' \n', and the
logits go through the CTC prefix beam search kernel.

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
from .utils.combination import combine_boxes
from .utils.text import CharTokenizer


class OcrPipeline:
    def __init__(self, east_params, crnn_params, device=0, tokenizer=None, obj_threshold=0.35, nms_threshold=0.2,
                 merge_threshold=0.3, merge_method='union', max_crop_elems=1 << 20, ctx=None, compute_dtype=F32,
                 rotated=False, combine=False, method='greedy', num_beams=10):
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
        if method == 'beam':
            method = 'beam_search'
        if method not in ('greedy', 'beam_search'):
            raise ValueError('Unknown CTC decoding method : {}'.format(method))
        if method == 'beam_search' and not 1 <= int(num_beams) <= HipContext.MAX_GPU_BEAMS:
            raise ValueError('num_beams must be 1..{} for the GPU beam search'.format(HipContext.MAX_GPU_BEAMS))
        if combine and rotated:
            raise ValueError('combine=True groups axis-aligned boxes; it does not go with rotated=True')
        self.combine, self.method, self.num_beams = bool(combine), method, int(num_beams)
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
            crop_boxes = per_image
            if self.combine:
                # words -> lines -> paragraphs (base_ocr.py:154-157); every line is one recogniser input
                st['groups'] = []
                crop_boxes = []
                for bx in per_image:
                    if len(bx):
                        para, _, rows = combine_boxes(bx)
                        rows = [np.asarray(r, dtype=np.float32).reshape(-1, 4) for r in rows]
                    else:
                        para, rows = bx, []
                    owner = np.concatenate([np.full(len(r), i, dtype=np.int64) for i, r in enumerate(rows)]) if rows else np.zeros((0,), np.int64)
                    st['groups'].append((np.asarray(para, dtype=np.float32).reshape(-1, 4), rows, owner))
                    crop_boxes.append(np.concatenate(rows) if rows else np.zeros((0, 4), np.float32))
            plans = [S.plan_crops(bx, H, W) for bx in crop_boxes]
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
        beam = self.method == 'beam_search'
        seq_T = self.ctx.seq_len(out_w).astype(np.int32)
        seq_row0 = (np.cumsum(seq_T) - seq_T).astype(np.int32)          # first row of each sequence in the flat logits
        h_meta = torch.from_numpy(np.concatenate([rects.reshape(-1), inter.reshape(-1), out_w, box_img, seq_T, seq_row0]).astype(np.int32)).pin_memory()
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
        out = self.ctx.crnn_forward_packed(packed, out_w, want_logits=beam, compute_dtype=self.compute_dtype)
        if beam:
            d_T, d_row0 = meta[(r + 4) * n:(r + 5) * n], meta[(r + 5) * n:(r + 6) * n]
            out['tokens'], out['scores'] = self.ctx.ctc_beam_search(out['logits'], d_T, d_row0, t_stride=int(seq_T.max()),
                                                                    num_beams=self.num_beams, blank=self.tokenizer.blank_token_idx)
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
        if 'groups' in st:
            return self._paragraph_strings(st)
        out = [{'boxes': per_image[b], 'texts': [''] * len(per_image[b]), 'scores': [None] * len(per_image[b]),
                'valid': np.zeros(len(per_image[b]), dtype=bool)} for b in range(B)]
        if st['n'] == 0:
            return out
        tokens, scores = self._wait_tokens(st)
        box_img, box_idx = st['box_img'], st['box_idx']
        texts = self.tokenizer.decode_batch(tokens[:st['n']])
        for k, (b, i, text, score) in enumerate(zip(box_img.tolist(), box_idx.tolist(), texts, scores[:st['n']].tolist())):
            o = out[b]
            o['texts'][i] = text
            o['scores'][i] = score
            o['valid'][i] = True
        return out

    def _wait_tokens(self, st):
        """Tokens and scores of a batch on the host, after the check (and, if needed, the repeat) of its recogniser."""
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
        return st['h_tokens'].numpy(), st['h_scores'].numpy()

    def _paragraph_strings(self, st):
        """combine=True: per page dict(boxes (P,4) paragraph boxes, rows [(k,4) line boxes], texts [lines joined with
        ' \n'], scores [(k',) per valid line], valid (P,)): what BaseOCR.infer returns per entry of 'ocr'
        (base_ocr.py:171-236); a paragraph none of whose lines gave a crop is not valid."""
        out = []
        for para, rows, _ in st['groups']:
            out.append({'boxes': para, 'rows': rows, 'texts': [''] * len(para), 'scores': [np.zeros((0,), np.float32)] * len(para),
                        'valid': np.zeros(len(para), dtype=bool)})
        if st['n'] == 0:
            return out
        tokens, scores = self._wait_tokens(st)
        texts = self.tokenizer.decode_batch(tokens[:st['n']])
        lines = [dict() for _ in out]               # page -> {line index: (text, score)}
        for b, i, text, score in zip(st['box_img'].tolist(), st['box_idx'].tolist(), texts, scores[:st['n']].tolist()):
            lines[b][i] = (text, score)
        for b, (para, rows, owner) in enumerate(st['groups']):
            o = out[b]
            for p in range(len(para)):
                mine = [lines[b][i] for i in np.nonzero(owner == p)[0].tolist() if i in lines[b]]
                if mine:
                    o['texts'][p] = ' \n'.join(t for t, _ in mine)
                    o['scores'][p] = np.array([sc for _, sc in mine], dtype=np.float32)
                    o['valid'][p] = True
        return out

    def _recognise_again(self, st):
        """The recogniser of a batch once more on the crops still packed on the GPU, `lstm_persistent = 0`."""
        packed = st['_keep'][4]
        before = self.ctx.options.get('lstm_persistent', 2)
        self.ctx.set_option('lstm_persistent', 0)
        try:
            beam = self.method == 'beam_search'
            out = self.ctx.crnn_forward_packed(packed, st['out_w'], want_logits=beam, compute_dtype=self.compute_dtype)
            if beam:
                n, r, meta = st['n'], st['_keep'][0].numel() // st['n'] - 6, st['_keep'][2]
                out['tokens'], out['scores'] = self.ctx.ctc_beam_search(
                    out['logits'], meta[(r + 4) * n:(r + 5) * n], meta[(r + 5) * n:(r + 6) * n], t_stride=int(out['tokens'].shape[1]),
                    num_beams=self.num_beams, blank=self.tokenizer.blank_token_idx)
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
