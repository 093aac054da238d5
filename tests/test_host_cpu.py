"""CPU-only checks: the C ABI library loads and exports every symbol the header declares, and the
host-side logic (size arithmetic, slice semantics, tokenizer, registries) follows the reference."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ocr_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ocrhip.h')).read()
    declared = set(re.findall(r'\b(ocrhip_[a-z_0-9]+)\s*\(', header))
    assert len(declared) >= 17
    assert declared == set(_lib.SIGNATURES.keys())
    lib = _lib.load()                      # raises if the .so is missing or a symbol is not exported
    for name in declared:
        assert hasattr(lib, name)
    assert lib.ocrhip_version() >= 100


def test_size_arithmetic_matches_reference_golden(golden):
    from ocr_amd.utils import image_sizes as S
    from oracle import image as oimg
    for h, w, out in zip(golden['crnn_size_h'], golden['crnn_size_w'], golden['crnn_size_out']):
        assert S.crnn_output_size(int(h), int(w)) == (int(out[0]), int(out[1]))
        (ih, iw), (oh, ow) = oimg.crnn_sizes(int(h), int(w))
        assert S.crnn_intermediate_size(int(h), int(w), oh, ow) == (ih, iw)
    for (h, w), out in zip(golden['east_size_in'], golden['east_size_out']):
        assert S.east_input_size(int(h), int(w)) == (int(out[0]), int(out[1]))


def test_crop_rectangles_match_reference_golden(golden):
    from ocr_amd.utils import image_sizes as S
    H, W = int(golden['crop_image_shape'][0]), int(golden['crop_image_shape'][1])
    rel = golden['crop_rel']
    np.testing.assert_array_equal(S.relative_to_absolute(rel, H, W), golden['crop_abs'])
    rects, valid = S.resolve_slices(golden['crop_abs'], H, W)
    for r, v, shp in zip(rects, valid, golden['crop_shapes']):
        assert (int(r[3] - r[1]), int(r[2] - r[0])) == (int(shp[0]), int(shp[1]))
        assert bool(v) == (shp[0] > 0 and shp[1] > 0)


def test_uint8_to_float_is_one_rounding():
    # x / 255 in float64 then float32 (reference) == float32 division (kernels) for every byte value
    u = np.arange(256)
    np.testing.assert_array_equal((u / 255).astype(np.float32), u.astype(np.float32) / np.float32(255))


def test_tokenizer_follows_reference_rules():
    from ocr_amd.utils.text import CharTokenizer, ENGLISH_G2_CHARACTERS
    from oracle.ctc import CharTokenizer as OracleTok
    tok = CharTokenizer()
    assert tok.vocab_size == 97 and tok.blank_token_idx == 0
    dup = CharTokenizer('abca')            # setdefault: the second 'a' keeps id 1, vocabulary shrinks
    assert dup.vocab_size == 4
    rng = np.random.default_rng(0)
    o = OracleTok(ENGLISH_G2_CHARACTERS)
    for _ in range(50):
        row = rng.integers(-1, 97, 30)
        assert tok.decode(row) == o.decode(row)
    assert tok.decode([5, 0, 0, 7]) == tok._id_to_symbol[5]          # two blanks in a row end the text
    assert tok.decode([-1, -1]) == ''


def test_batch_decoding_equals_row_by_row_decoding():
    from ocr_amd.utils.text import CharTokenizer, get_characters
    rng = np.random.default_rng(3)
    for chars in (None, get_characters('zh_sim_g2'), 'abca'):
        tok = CharTokenizer(chars) if chars else CharTokenizer()
        V = tok.vocab_size
        rows = rng.integers(-2, V + 3, (300, 40)).astype(np.int32)          # ids without a symbol decode to nothing
        rows[rng.random(rows.shape) < 0.3] = tok.blank_token_idx            # single and double blanks (a double one ends the text)
        rows[::3, 12:] = -1                                                 # the padding of the CTC kernel
        assert tok.decode_batch(rows) == [tok.decode(r) for r in rows]
        assert tok.decode_batch(rows[:, :1]) == [tok.decode(r) for r in rows[:, :1]]
        assert tok.decode_batch(np.zeros((0, 7), np.int32)) == []


def test_registries_raise_like_the_reference():
    from ocr_amd.models import get_pretrained
    from ocr_amd.runtimes import build_runtime
    from ocr_amd.utils.boxes import nms, convert_box_format
    with pytest.raises(ValueError, match='does not exist'):
        get_pretrained('no_such_model')
    with pytest.raises(ValueError, match='Unsupported runtime'):
        build_runtime('trt', 'x')
    with pytest.raises(ValueError, match='Unknown NMS method'):
        nms(np.zeros((2, 4), np.float32), nms_threshold=0.2, method='nope')
    b, s, m = nms(np.zeros((0, 4), np.float32), nms_threshold=0.2)
    assert len(m) == 0
    b, s, m = nms(np.ones((3, 4), np.float32), nms_threshold=1.0)
    assert m.all()
    called = {}

    def custom(boxes, scores, max_output_size, nms_threshold, **kw):
        called['ok'] = True
        return boxes, scores, np.ones(len(boxes), bool)
    nms(np.ones((3, 4), np.float32), nms_threshold=0.2, method=custom)
    assert called['ok']
    polys = np.array([[[0.1, 0.2], [0.5, 0.2], [0.5, 0.4], [0.1, 0.4]]], np.float32)
    np.testing.assert_allclose(convert_box_format(polys, source='poly', target='xyxy'), [[0.1, 0.2, 0.5, 0.4]], atol=1e-7)


def test_float_images_round_trip_to_uint8():
    """uint8 -> float32 / 255 (the reference's loader) -> the GPU path's uint8 is the identity for all 256 values."""
    from ocr_amd.utils.image_sizes import quantise_u8
    u = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)
    np.testing.assert_array_equal(quantise_u8(u.astype(np.float32) / np.float32(255)), u)
    np.testing.assert_array_equal(quantise_u8((u / 255.0).astype(np.float64)), u)
    assert quantise_u8(u) is u
    np.testing.assert_array_equal(quantise_u8(np.array([[-0.1, 1.2]], np.float32)), [[0, 255]])


def test_easyocr_character_tables():
    """Vocabulary sizes of the EasyOCR tables the reference lists (crnn_arch.py:45-203), blank included, after the
    reference tokenizer's `setdefault` de-duplication (tokenizer.py:189-219)."""
    import pytest
    from ocr_amd.utils.text import CharTokenizer, ENGLISH_G2_CHARACTERS, easyocr_models, get_characters
    sizes = {name: CharTokenizer(info['characters']).vocab_size for name, info in easyocr_models().items()}
    assert sizes['english_g2'] == 97 and sizes['latin_g2'] == 352 and sizes['zh_sim_g2'] == 6719
    assert sizes['zh_sim_g1'] == 6710 and sizes['zh_tra_g1'] == 5381 and sizes['japanese_g1'] == 3111
    assert sizes['japanese_g2'] == 2215 and sizes['korean_g1'] == 1568 and sizes['korean_g2'] == 1009
    # two tables repeat characters: ids after a repeat shift down, the head is smaller than the table
    assert len(get_characters('devanagari_g1')) == 187 and sizes['devanagari_g1'] == 183
    assert len(get_characters('arabic_g1')) == 184 and sizes['arabic_g1'] == 183
    assert get_characters('english_g2') == ENGLISH_G2_CHARACTERS
    tok = CharTokenizer(get_characters('zh_sim_g2'))
    ids = [tok._symbol_to_id[c] for c in '中文OCR']
    assert tok.decode(ids + [0, 0, 5]) == '中文OCR'
    with pytest.raises(ValueError):
        get_characters('klingon_g9')


def test_dezoom_matches_reference_vectors_and_known_answers():
    """`dezoom_factor` of convert_box_format / crop_box: vectors made by the reference's own function
    (tests/golden/make_dezoom_golden.py) and its known answers (tests/test_utils_boxes.py:116-135)."""
    import os
    from conftest import GOLDEN_DIR
    from ocr_amd.utils.boxes import convert_box_format
    from ocr_amd.utils.image_sizes import dezoom_absolute, relative_to_absolute
    g = np.load(os.path.join(GOLDEN_DIR, 'dezoom_golden.npz'))
    H, W = [int(v) for v in g['image_hw']]
    for f in g['factors']:
        f = float(f)
        rel = convert_box_format(g['boxes'], target='xyxy', source='xyxy', dezoom_factor=f)
        np.testing.assert_array_equal(rel, g['rel_%g' % f])
        ab = convert_box_format(g['boxes'], target='xyxy', source='xyxy', dezoom_factor=f, normalize_mode='absolute',
                                image_h=H, image_w=W)
        assert ab.dtype == np.int32
        np.testing.assert_array_equal(ab, g['abs_%g' % f])
        # the crop planner's own copy of the pixel arithmetic
        np.testing.assert_array_equal(dezoom_absolute(relative_to_absolute(g['boxes'], H, W), f, H, W), g['abs_%g' % f])
    kat = lambda box, f: convert_box_format(np.array(box), source='xywh', dezoom_factor=f)
    np.testing.assert_allclose(kat([0., 0., 1., 1.], 0.5), [[0.25, 0.25, 0.5, 0.5]], atol=1e-6)
    np.testing.assert_allclose(kat([0., 0., 1., 1.], 2), [[0, 0, 1, 1]], atol=1e-6)
    np.testing.assert_allclose(kat([0.25, 0.25, .5, .5], 2), [[0, 0, 1, 1]], atol=1e-6)
    np.testing.assert_allclose(kat([0.5, 0.5, .5, .5], 2), [[0.25, 0.25, .75, .75]], atol=1e-6)


def test_load_checkpoint_reads_the_files_the_original_projects_ship(tmp_path):
    """torch checkpoints as EasyOCR / SakuraRiven save them: `module.` prefixes of DataParallel, batch-norm
    `num_batches_tracked` counters, optionally nested under 'state_dict'; read with the weights-only loader."""
    import torch
    from ocr_amd.weights import load_checkpoint, random_crnn_params
    params = random_crnn_params(97, seed=3)
    saved = {'module.' + k: v for k, v in params.items()}
    saved['module.FeatureExtraction.ConvNet.12.num_batches_tracked'] = torch.tensor(1234)
    for name, obj in (('flat.pth', saved), ('nested.pth', {'state_dict': saved, 'epoch': 3})):
        path = str(tmp_path / name)
        torch.save(obj, path)
        got = load_checkpoint(path)
        assert set(got) == set(params) | {'FeatureExtraction.ConvNet.12.num_batches_tracked'}
        for k, v in params.items():
            assert torch.equal(got[k], v)
