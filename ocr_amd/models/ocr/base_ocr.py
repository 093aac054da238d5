"""`BaseOCR` / `CRNN` with the reference's interface (models/ocr/base_ocr.py:53-255,371-375,
models/ocr/crnn.py:15-58), running on libocrhip.so.

Same defaults as the reference (`combine=True`, `method='beam_search'`, 10 beams).  Beam search runs on
the host from the GPU logits; `method='greedy'` decodes on the GPU.  Callbacks and the `map.json` prediction cache
(`predict(save=True)`) work as in the reference (utils/saving.py), so do `box_filters` (utils/filters.py);
`dezoom_factor` scales every row's pixel box about its centre before the crop, as the reference does; `stream()`
runs the reference's frame / text de-duplication over an iterable of frames; the image saver, the displayer and the
camera / video readers are not built.
"""
import logging
import os

import numpy as np
import torch

from .. import register_model
from ..detection.east import get_image_data, load_image
from ...runtimes import HIPRuntime
from ...utils.image_sizes import plan_crops, plan_crops_rotated, quantise_u8
from ...utils.saving import FunctionCallback, JSONSaver, QueueCallback, apply_callbacks, load_json
from ...utils.text import CharTokenizer, ENGLISH_G2_CHARACTERS

logger = logging.getLogger(__name__)


class BaseOCR:
    def __init__(self, lang='multi', weights=None, *, name='crnn_en', characters=ENGLISH_G2_CHARACTERS,
                 input_size=(64, None, 1), image_normalization='easyocr', resize_kwargs=None, runtime='hip', device=0,
                 max_output_length=None, **kwargs):
        if runtime != 'hip':
            raise ValueError('Unsupported runtime !\n  Accepted : {}\n  Got : {}'.format(('hip',), runtime))
        self.lang = lang
        self.name = name
        self.input_size = tuple(input_size)
        self.image_normalization = image_normalization
        self.resize_kwargs = resize_kwargs or {'antialias': True, 'interpolation': 'bilinear',
                                               'preserve_aspect_ratio': True, 'pad_mode': 'repeat_last',
                                               'multiples': 64}
        self.max_output_length = max_output_length
        self.tokenizer = CharTokenizer(characters)
        self.runtime = runtime
        if isinstance(weights, dict):
            self.model = HIPRuntime('state_dict:' + name, state_dict=weights, device=device, reload=True)
        else:
            self.model = HIPRuntime(weights if weights is not None
                                    else 'random:crnn:{}:0'.format(self.tokenizer.vocab_size), device=device)
        self.ctx = self.model.engine.ctx
        if self.ctx.vocab_size != self.tokenizer.vocab_size:
            raise ValueError('The checkpoint predicts {} classes but the vocabulary has {} symbols'.format(
                self.ctx.vocab_size, self.tokenizer.vocab_size))
        register_model(name, self)

    is_encoder_decoder = False

    @property
    def vocab_size(self):
        return self.tokenizer.vocab_size

    @property
    def blank_token_idx(self):
        return self.tokenizer.blank_token_idx

    @property
    def compiled_infer(self):
        return self.model

    # ---- one row, as the reference prepares it (base_image_model.py:93-131 with use_box=True)
    def get_input(self, data, dezoom_factor=1., **kwargs):
        """{'filename': image, 'boxes': {'boxes': row, 'format': 'xyxy'}} -> (64, Wpad, 1) float32 GPU tensor
        (a tensor with a zero dimension for an empty crop)."""
        image = load_image(data)
        boxes = data['boxes']['boxes'] if isinstance(data['boxes'], dict) else data['boxes']
        out = self._pack(torch.from_numpy(np.ascontiguousarray(image)).to(self.ctx.device)[None],
                         [np.asarray(boxes, dtype=np.float32).reshape(-1, 4)[:1]], dezoom_factor=dezoom_factor)
        if out is None:
            return torch.zeros((64, 0, 1), device=self.ctx.device)
        packed, plan, off = out
        return packed.view(64, int(plan['out_w'][0]), 1)

    def _pack(self, d_images, boxes_per_image, dezoom_factor=1.):
        B, H, W, _ = d_images.shape
        plans, box_img = [], []
        rotated = any(np.ndim(boxes) == 3 for boxes in boxes_per_image)      # (m,4,2) quadrilaterals
        for b, boxes in enumerate(boxes_per_image):
            if rotated:
                if dezoom_factor != 1.:
                    raise NotImplementedError('dezoom_factor is not supported for quadrilaterals')    # converter.py:204
                p = plan_crops_rotated(boxes, H, W)
            else:
                p = plan_crops(boxes, H, W, dezoom_factor=dezoom_factor)
            plans.append(p)
            box_img.append(np.full(len(p['index']), b, dtype=np.int32))
        rect_key = 'geom' if rotated else 'rects'
        plan = {k: np.concatenate([p[k] for p in plans]) for k in (rect_key, 'inter', 'out_w', 'index')}
        plan['box_img'] = np.concatenate(box_img)
        n = len(plan['out_w'])
        if n == 0:
            return None
        off = np.concatenate([[0], np.cumsum(64 * plan['out_w'].astype(np.int64))])
        dev = d_images.device
        crop = self.ctx.crop_rotated_pack if rotated else self.ctx.crop_resize_pack
        packed = crop(
            d_images, torch.from_numpy(plan[rect_key]).to(dev), torch.from_numpy(plan['box_img']).to(dev),
            torch.from_numpy(plan['inter']).to(dev), torch.from_numpy(plan['out_w']).to(dev),
            torch.from_numpy(off[:-1].astype(np.int64)).to(dev), int(off[-1]))
        return packed, plan, off

    def recognise_rows(self, image, rows, method='greedy', num_beams=10, dezoom_factor=1.):
        """All rows of one image in one go -> (index of valid rows, texts, scores).

        method 'greedy': argmax + collapse on the GPU.  method 'beam_search' / 'beam': CTC prefix beam search on
        the GPU (one wave per row, ocrhip_ctc_beam_search) for up to 16 beams; wider beams go through the host
        implementation of the same search (utils/ctc_beam.py)."""
        if method == 'beam':
            method = 'beam_search'
        if method not in ('greedy', 'beam_search'):
            raise ValueError('Unknown CTC decoding method : {}'.format(method))
        image = quantise_u8(image)          # float input (u / 255) is rounded back to 8 bits
        if image.shape[2] == 1:
            image = np.repeat(image, 3, axis=2)
        d_img = torch.from_numpy(np.ascontiguousarray(image)).to(self.ctx.device)[None]
        rows = np.asarray(rows, dtype=np.float32)
        rows = rows.reshape(-1, 4, 2) if rows.ndim == 3 else rows.reshape(-1, 4)      # quadrilaterals or xyxy
        out = self._pack(d_img, [rows], dezoom_factor=dezoom_factor)
        if out is None:
            return [], [], np.zeros((0,), np.float32)
        packed, plan, off = out
        n = len(plan['index'])
        order = np.argsort(plan['out_w'], kind='stable')          # the recogniser wants non-decreasing widths
        widths = plan['out_w'][order]
        sorted_packed = torch.cat([packed[off[j]:off[j + 1]] for j in order]) if n > 1 else packed
        res = self.ctx.crnn_forward_packed(sorted_packed, widths, want_logits=(method == 'beam_search'))
        if method == 'greedy':
            toks, sc = res['tokens'].cpu().numpy(), res['scores'].cpu().numpy()
            self.ctx.check_errors(synchronize=False)
        elif 1 <= int(num_beams) <= self.ctx.MAX_GPU_BEAMS:
            T = self.ctx.seq_len(widths).astype(np.int32)
            row0 = (np.cumsum(T) - T).astype(np.int32)
            dev = res['logits'].device
            d_tok, d_sc = self.ctx.ctc_beam_search(res['logits'], torch.from_numpy(T).to(dev), torch.from_numpy(row0).to(dev),
                                                   t_stride=int(T.max()), num_beams=int(num_beams), blank=self.blank_token_idx)
            toks, sc = d_tok.cpu().numpy(), d_sc.cpu().numpy()
            self.ctx.check_errors(synchronize=False)
        else:
            from ...utils.ctc_beam import ctc_beam_search
            T = self.ctx.seq_len(widths)
            row0 = np.concatenate([[0], np.cumsum(T)])
            flat = res['logits'].cpu().numpy()
            self.ctx.check_errors(synchronize=False)
            padded = np.zeros((n, int(T.max()), flat.shape[1]), np.float32)      # pad_batch(pad_value=0.)
            for k in range(n):
                padded[k, :T[k]] = flat[row0[k]:row0[k + 1]]
            toks, sc = ctc_beam_search(padded, T, num_beams=num_beams, blank=self.blank_token_idx)
        texts, scores = [None] * n, np.zeros(n, np.float32)
        decoded = self.tokenizer.decode_batch(np.asarray(toks)[:n])
        for k, j in enumerate(order):
            texts[j] = decoded[k]
            scores[j] = sc[k]
        return list(plan['index']), texts, scores

    def infer(self, data, *, detector='east', detector_kwargs={'merge_method': 'union'}, combine=True,
              box_filters=None, dezoom_factor=1., method='beam_search', num_beams=10, num_sentences=1, length_power=.25,
              threshold=0., box_callback=None, callbacks=None, predicted=None, overwrite=False, return_output=True,
              **kwargs):
        """base_ocr.py:98-255.  Returns dict(filename, boxes, ocr=[dict(boxes, rows, text, format, scores)], image).

        With `combine=True` the detected words are grouped into lines and paragraphs first
        (utils/combination.py) and every line is one recogniser input; the texts of a paragraph are
        joined with ' \n' like the reference does."""
        if predicted and not overwrite and isinstance(data, str) and data in predicted:
            if callbacks:
                apply_callbacks(callbacks, predicted[data], {}, save=False)
            return predicted[data]
        filename, image = get_image_data(data)
        if not isinstance(data, dict) or 'boxes' not in data:
            if isinstance(detector, str):
                from .. import get_pretrained
                detector = get_pretrained(detector, **detector_kwargs)
            data = detector.infer(image)
        boxes = data['boxes']
        if isinstance(boxes, dict):
            boxes = boxes['boxes']
        boxes = np.asarray(boxes, dtype=np.float32)
        quads = boxes.reshape(-1, 4, 2) if boxes.ndim == 3 else None      # 'poly' boxes of nms_method='rotated' (opt-in)
        if quads is not None:
            # grouping and filters work on axis-aligned bounds; every quadrilateral stays one recogniser input
            boxes = np.concatenate([quads.min(axis=1), quads.max(axis=1)], axis=1)
            combine = False
        boxes = boxes.reshape(-1, 4)
        if combine and len(boxes):
            from ...utils.combination import combine_boxes
            boxes, word_indices, rows = combine_boxes(boxes, **kwargs)
            rows = [np.asarray(r, dtype=np.float32).reshape(-1, 4) for r in rows]
        else:
            rows = [boxes[i:i + 1] for i in range(len(boxes))]
            word_indices = [[[i]] for i in range(len(boxes))]
        if box_filters:                                     # base_ocr.py:164-165
            from ...utils.filters import filter_boxes
            boxes, word_indices, rows = filter_boxes(box_filters, boxes, list(word_indices), rows)
        flat_rows = np.concatenate(rows).reshape(-1, 4) if rows else np.zeros((0, 4), np.float32)
        if quads is not None:
            flat_rows = quads[[idx[0][0] for idx in word_indices]] if len(word_indices) else np.zeros((0, 4, 2), np.float32)
        owner = np.concatenate([np.full(len(r), i) for i, r in enumerate(rows)]) if rows else np.zeros((0,), int)
        index, texts, scores = self.recognise_rows(np.asarray(image), flat_rows, method=method, num_beams=num_beams,
                                                   dezoom_factor=dezoom_factor)
        if len(flat_rows) - len(index):
            logger.warning('Invalid input encountered : %d empty crops skipped', len(flat_rows) - len(index))
        ocr_results = []
        for i, rows_i in enumerate(rows):
            sel = [k for k, j in enumerate(index) if owner[j] == i]
            if not sel:
                continue
            sc = np.array([scores[k] for k in sel], dtype=np.float32)
            if threshold < 0. and np.any(sc < threshold):
                continue
            info = {'boxes': boxes[i], 'rows': rows_i, 'text': ' \n'.join(texts[k] for k in sel), 'format': 'xyxy',
                    'scores': sc}
            if quads is not None:
                info['quad'] = flat_rows[i]
            ocr_results.append(info)
            if box_callback is not None:
                box_callback(info, image=image)
        output = {'filename': filename, 'boxes': data.get('boxes', None) if isinstance(data, dict) else None,
                  'ocr': ocr_results, 'image': image}
        entry = None
        if callbacks:
            infos = {k: v for k, v in output.items() if 'image' not in k}
            entry = apply_callbacks(callbacks, infos, output, save=True)
        return output if return_output else (predicted or {}).get(entry, {})

    def stream(self, stream, *, detector='east', detector_kwargs=None, callback=None, post_processing=None, **kwargs):
        """base_ocr.py:518-669 over an iterable of frames (arrays, file names): detection per frame, duplicate frames
        and repeated / already emitted texts dropped, `callback(text)` for every new text.  Returns the
        `streaming.OcrStream` that ran (`.emitted`: (frame index, result) pairs; `.state`).  See models/ocr/streaming.py;
        cameras and video files are not opened here."""
        from .streaming import run_stream
        return run_stream(self, stream, detector=detector, detector_kwargs=detector_kwargs,
                          callback=post_processing if post_processing is not None else callback, **kwargs)

    def get_inference_callbacks(self, *, save=True, directory=None, post_processing=None, **_):
        """base_ocr.py:310-368 without the image saver and the displayer: with `save`, the results already in
        `<directory>/map.json` and a saver that adds every new one (keyed by the input's filename)."""
        predicted, callbacks = {}, []
        if save:
            if directory is None:
                directory = os.path.join('pretrained_models', self.name, 'predictions')      # base_model.py:129
            map_file = os.path.join(directory, 'map.json')
            predicted = load_json(map_file, {})
            callbacks.append(JSONSaver(data=predicted, filename=map_file, primary_key='filename'))
        if post_processing is not None:
            for fn in post_processing if isinstance(post_processing, list) else [post_processing]:
                if callable(fn):
                    callbacks.append(FunctionCallback(fn))
                elif hasattr(fn, 'put'):
                    callbacks.append(QueueCallback(fn))
        return predicted, callbacks

    def predict(self, inputs, *, predicted=None, callbacks=None, return_results=True, return_output=None,
                save=False, directory=None, post_processing=None, **kwargs):
        """base_ocr.py:371-375 over base_model.py:679-712: a single image / filename / dict, or a list of them -> list of
        results.  `save=True` (the reference's default; off by default here so that nothing is written unasked) resumes
        from and appends to `<directory>/map.json`: files already in it are not processed again, and the results
        returned for saved inputs are the JSON entries, as in the reference."""
        if isinstance(inputs, (str, dict)) or (hasattr(inputs, 'shape') and len(inputs.shape) == 3):
            inputs = [inputs]
        join_callbacks = predicted is None
        if predicted is None:
            predicted, callbacks = self.get_inference_callbacks(save=save, directory=directory, post_processing=post_processing)
        callbacks = callbacks or []
        if return_output is None:
            return_output = not any(isinstance(c, JSONSaver) for c in callbacks)
        results = []
        for inp in inputs:
            out = self.infer(inp, predicted=predicted, callbacks=callbacks, return_output=return_output, **kwargs)
            if return_results:
                results.append(out)
        if join_callbacks:
            for c in callbacks:
                c.join()
        return results


class CRNN(BaseOCR):
    """models/ocr/crnn.py: EasyOCR recogniser (64 x W x 1 input, 'easyocr' normalisation, char vocabulary
    ['<blank>'] + characters).  `pretrained` names one of the EasyOCR recognisers the reference lists
    (architectures/crnn_arch.py:45-203, e.g. 'zh_sim_g2', 'latin_g2'): its character table becomes the vocabulary."""

    def __init__(self, lang='multi', weights=None, *, pretrained=None, **kwargs):
        if pretrained is not None and 'characters' not in kwargs:
            from ...utils.text import get_characters
            kwargs['characters'] = get_characters(pretrained)
        super().__init__(lang, weights, **kwargs)
