"""Prediction cache on disk: the `map.json` of the reference's OCR models.

`BaseOCR.predict(..., save=True)` keeps one entry per input file in `<directory>/map.json`
(models/ocr/base_ocr.py:310-368); an image whose filename is already a key is not processed again
(`base_ocr.py:142-144`), and every new result is written through a `JSONSaver` callback
(utils/callbacks/file_saver.py:145-192) after conversion by `to_json` (utils/generic_utils.py:72-104) and
`dump_json(..., indent=4)` (utils/file_utils.py:333-339).  The file this module writes is byte for byte the file
the reference writes for the same result dictionary (pinned by tests/golden/map_golden.json).

Host-side only: nothing here touches the GPU path.
"""
import datetime
import enum
import glob
import json
import logging
import os
import threading
import uuid

import numpy as np

logger = logging.getLogger(__name__)


def _is_path(data):
    """utils/file_utils.py:37-41: a string that names an existing file (glob match)."""
    if not isinstance(data, str) or len(data) >= 512 or len(data) == 0:
        return False
    if data[0] == '~':
        data = os.path.expanduser(data)
    return len(glob.glob(data)) > 0


def _array_to_json(data):
    data = data.detach().cpu().numpy() if hasattr(data, 'detach') else np.asarray(data)
    if data.ndim == 0:
        return to_json(data.item())
    if data.size >= 256:
        logger.warning('Serialising an array of shape %s to json', data.shape)
    return [to_json(d) for d in data]


# (test, conversion) pairs tried in order; the first test that holds decides.  The order matters where types nest:
# bool before int (bool is an int), arrays before sequences, enums before anything their value could match.
_JSON_RULES = (
    (lambda d: d is None or isinstance(d, bool), lambda d: d),
    (lambda d: isinstance(d, enum.Enum), lambda d: to_json(d.name)),
    (lambda d: hasattr(d, 'shape'), _array_to_json),
    (lambda d: isinstance(d, bytes), lambda d: to_json(d.decode('utf-8'))),
    (lambda d: isinstance(d, uuid.UUID), str),
    (lambda d: isinstance(d, datetime.datetime), lambda d: d.strftime('%Y-%m-%d %H:%M:%S')),
    (lambda d: isinstance(d, (float, np.floating)), float),
    (lambda d: isinstance(d, (int, np.integer)), int),
    (lambda d: isinstance(d, (list, tuple, set)), lambda d: [to_json(x) for x in d]),
    (lambda d: isinstance(d, dict), lambda d: {to_json(k): to_json(v) for k, v in d.items()}),
    (lambda d: isinstance(d, str), lambda d: d.replace('\\', '/') if _is_path(d) else d),
    (lambda d: hasattr(d, 'get_config'), lambda d: to_json(d.get_config())),
    (lambda d: callable(d) and hasattr(d, '__name__'), lambda d: d.__name__),
)


def to_json(data):
    """Anything a result dictionary holds -> what `json.dump` accepts, with the reference's conventions
    (utils/generic_utils.py:72-104): arrays / tensors become nested lists of Python numbers (float32 widened to
    double, as `float()` does), enums their name, datetimes 'Y-m-d H:M:S', strings that name existing files get '/'
    separators, objects with `get_config()` their config, functions their name; anything else `str()` with a warning."""
    for applies, convert in _JSON_RULES:
        if applies(data):
            return convert(data)
    logger.warning('No json form for %s (%r): stored as str', type(data).__name__, data)
    return str(data)


def load_json(filename, default=None):
    """utils/file_utils.py:207-212."""
    if not os.path.exists(filename):
        return {} if default is None else default
    with open(filename, 'r', encoding='utf-8') as fh:
        return json.loads(fh.read())


def dump_json(filename, data, **kwargs):
    """utils/file_utils.py:333-339."""
    text = json.dumps(to_json(data), **kwargs)
    with open(filename, 'w', encoding='utf-8') as fh:
        fh.write(text)


class JSONSaver:
    """utils/callbacks/file_saver.py:145-192: stores `to_json(infos)` under `infos[primary_key]` in `data` and rewrites
    the file (synchronously, the reference's `save_in_parallel=False` mode)."""

    def __init__(self, data, filename, primary_key, **_):
        self.data, self.filename, self.primary_key = data, filename, primary_key
        self.mutex = threading.Lock()

    def __repr__(self):
        return '<JSONSaver file={}>'.format(self.filename)

    def __call__(self, infos, output=None, **_):
        if self.primary_key not in infos:
            return None
        key = infos[self.primary_key]
        if not isinstance(key, str):
            return None
        with self.mutex:
            self.data[key] = to_json(infos)
            directory = os.path.dirname(self.filename)
            if directory:
                os.makedirs(directory, exist_ok=True)
            dump_json(self.filename, self.data, indent=4)
        return key

    def join(self):
        pass


class FunctionCallback:
    """utils/callbacks/function_callback.py:18-30: `fn(**infos, **output)` after every result."""

    def __init__(self, fn, include_outputs=True):
        self.fn, self.include_outputs = fn, include_outputs

    def __call__(self, infos, output=None, **kwargs):
        kwargs.update(infos)
        if self.include_outputs and output:
            kwargs.update(output)
        return self.fn(**kwargs)

    def join(self):
        pass


class QueueCallback:
    """utils/callbacks/function_callback.py:32-41: puts the result's entries on a queue."""

    def __init__(self, queue):
        self.queue = queue

    def __call__(self, infos, output=None, **kwargs):
        kwargs.update(infos)
        return self.queue.put(kwargs)

    def join(self):
        pass


def apply_callbacks(callbacks, infos, output, save=True, **kwargs):
    """utils/callbacks/__init__.py:30-45: exceptions of a callback are logged, not raised; file savers are skipped
    when `save` is false; returns the key the JSON saver stored the entry under."""
    if not callbacks:
        return None
    entry = None
    for callback in callbacks:
        if isinstance(callback, JSONSaver) and not save:
            continue
        try:
            res = callback(infos, output, **kwargs)
            if isinstance(callback, JSONSaver):
                entry = res
        except Exception as e:                                     # noqa: BLE001 (the reference swallows them too)
            logger.error('- An exception occured while calling %s : %s', callback, e)
    return entry
