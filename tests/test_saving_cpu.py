"""map.json prediction cache (ocr_amd/utils/saving.py, BaseOCR.predict(save=True)) against the file the reference
writes for the same result dictionary (tests/golden/map_golden.json, made by tests/golden/make_map_golden.py with the
reference's own to_json / dump_json).  Host logic only: the recogniser is stubbed, no GPU."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_map_golden import sample_infos  # noqa: E402

from ocr_amd.utils.saving import FunctionCallback, JSONSaver, apply_callbacks, dump_json, load_json, to_json  # noqa: E402


def test_map_json_is_byte_identical_to_the_reference_file(tmp_path):
    data = {}
    saver = JSONSaver(data=data, filename=str(tmp_path / 'pred' / 'map.json'), primary_key='filename')
    for k in range(2):
        infos = sample_infos(k)
        assert apply_callbacks([saver], infos, dict(infos, image=np.zeros((4, 4, 3), np.uint8))) == infos['filename']
    got = (tmp_path / 'pred' / 'map.json').read_text(encoding='utf-8')
    want = open(os.path.join(HERE, 'golden', 'map_golden.json'), encoding='utf-8').read()
    assert got == want
    assert load_json(str(tmp_path / 'pred' / 'map.json')) == json.loads(want) == data


def test_to_json_types(tmp_path):
    import torch
    assert to_json(np.float32(0.1)) == float(np.float32(0.1))
    assert to_json(np.array(3, np.int64)) == 3 and isinstance(to_json(np.array(3, np.int64)), int)
    assert to_json(torch.tensor([[1.5, 2.0]])) == [[1.5, 2.0]]
    assert to_json({'a': (1, 2), 'b': None, 'c': True, 'd': b'xy'}) == {'a': [1, 2], 'b': None, 'c': True, 'd': 'xy'}
    f = tmp_path / 'x.txt'
    f.write_text('x')
    assert to_json(str(f)) == str(f).replace('\\', '/')          # existing paths get unix separators
    assert to_json('no\\such\\file') == 'no\\such\\file'        # other strings are left alone
    assert load_json(str(tmp_path / 'missing.json'), {}) == {}
    dump_json(str(tmp_path / 'd.json'), {'k': np.arange(3)})
    assert load_json(str(tmp_path / 'd.json')) == {'k': [0, 1, 2]}


def test_callbacks_swallow_errors_and_skip_savers_when_not_saving(tmp_path):
    data, seen = {}, []
    saver = JSONSaver(data=data, filename=str(tmp_path / 'map.json'), primary_key='filename')

    def boom(**kw):
        raise RuntimeError('callback failure')
    cbs = [FunctionCallback(boom), saver, FunctionCallback(lambda **kw: seen.append(kw['filename']))]
    assert apply_callbacks(cbs, {'filename': 'a.jpg', 'ocr': []}, {}) == 'a.jpg'
    assert seen == ['a.jpg'] and 'a.jpg' in data
    assert apply_callbacks(cbs, {'filename': 'b.jpg', 'ocr': []}, {}, save=False) is None
    assert 'b.jpg' not in data and seen == ['a.jpg', 'b.jpg']
    assert apply_callbacks([saver], {'filename': np.zeros(2)}, {}) is None       # raw images have no key
    assert apply_callbacks([], {}, {}) is None


def test_predict_resumes_from_map_json(tmp_path, monkeypatch):
    """base_model.py:679-712 + base_ocr.py:142-144: files already in map.json are not processed again; with a JSON
    saver the results returned are the stored entries."""
    from ocr_amd.models.ocr import base_ocr as mod
    calls = []

    class Stub(mod.BaseOCR):
        def __init__(self):
            self.name = 'stub'

        def recognise_rows(self, image, rows, method="greedy", num_beams=10, dezoom_factor=1.):
            calls.append(len(rows))
            return list(range(len(rows))), ['w%d' % i for i in range(len(rows))], np.full(len(rows), -0.5, np.float32)

    monkeypatch.setattr(mod, 'get_image_data', lambda data: (data if isinstance(data, str) else data.get('filename'), np.zeros((32, 32, 3), np.uint8)))
    m = Stub()
    boxes = np.array([[0.1, 0.1, 0.4, 0.3], [0.5, 0.5, 0.9, 0.8]], np.float32)
    inputs = [{'filename': 'p%d.jpg' % i, 'boxes': {'boxes': boxes, 'format': 'xyxy'}} for i in range(2)]
    d = str(tmp_path / 'out')
    res = m.predict(inputs, save=True, directory=d, combine=False, method='greedy')
    assert len(calls) == 2 and [r['filename'] for r in res] == ['p0.jpg', 'p1.jpg']
    stored = load_json(os.path.join(d, 'map.json'))
    assert set(stored) == {'p0.jpg', 'p1.jpg'} and res[0] == stored['p0.jpg']
    assert [o['text'] for o in stored['p0.jpg']['ocr']] == ['w0', 'w1'] and 'image' not in stored['p0.jpg']
    # second run: filenames given as strings are served from the file, nothing is recognised again
    res2 = m.predict(['p0.jpg', 'p1.jpg'], save=True, directory=d, combine=False, method='greedy')
    assert len(calls) == 2 and res2 == [stored['p0.jpg'], stored['p1.jpg']]
    # overwrite=True processes them again; without a saver the full result (with the image) comes back
    res3 = m.predict({'filename': 'p0.jpg', 'boxes': boxes}, combine=False, method='greedy')
    assert len(calls) == 3 and 'image' in res3[0] and res3[0]['ocr'][1]['text'] == 'w1'
