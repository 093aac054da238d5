"""ctypes front end of oracle/lanms_c.c (test infrastructure): LANMS + matrix NMS for large N."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build():
    src = os.path.join(_HERE, 'lanms_c.c')
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(['make', '-C', _HERE, '_build/liboracle.so'], check=True, capture_output=True)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = ctypes.CDLL(_LIB)
        lib.oracle_lanms_merge.restype = ctypes.c_int
        lib.oracle_lanms_merge.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        lib.oracle_matrix_nms.restype = ctypes.c_int
        lib.oracle_matrix_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _lib = lib
    return _lib


def lanms(boxes, scores=None, max_output_size=None, nms_threshold=0.25, *, merge_threshold=0.3, max_iter=None,
          merge_method='union'):
    """Same contract as oracle.nms.lanms: (boxes, scores, valid_mask, order, n_iter), rank order."""
    from .nms import sort_boxes, get_valid_mask
    lib = _load()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    if scores is not None:
        scores = np.asarray(scores, dtype=np.float32)
    boxes, scores, order = sort_boxes(boxes, scores)
    boxes = np.ascontiguousarray(boxes)
    n = len(boxes)
    if n == 0:
        return boxes, scores, np.zeros((0,), dtype=bool), order, 0
    n_iter = lib.oracle_lanms_merge(boxes.ctypes.data, n, float(np.float32(merge_threshold)),
                                    {'union': 0, 'average': 1}[merge_method], int(max_iter or 0))
    mask = np.any(boxes > 0, axis=1)
    if np.float32(nms_threshold) < np.float32(merge_threshold):
        keep = np.zeros(n, dtype=np.uint8)
        lib.oracle_matrix_nms(boxes.ctypes.data, n, float(np.float32(nms_threshold)), keep.ctypes.data)
        keep = keep.astype(bool)
        if max_output_size is not None:
            keep = keep & (np.cumsum(keep.astype(np.int32)) <= max_output_size)
        mask = mask & keep
    return boxes, scores, mask, order, int(n_iter)
