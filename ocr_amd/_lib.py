"""ctypes binding of libocrhip.so (the C ABI declared in include/ocrhip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libocrhip.so')

OCRHIP_OK, OCRHIP_EINVAL, OCRHIP_ECAP, OCRHIP_EHIP, OCRHIP_ESTATE = 0, -1, -2, -3, -4
F32, BF16, U8 = 0, 1, 2
MERGE_UNION, MERGE_AVERAGE, MERGE_WEIGHTED = 0, 1, 2

c_int, c_float, c_void_p, c_char_p, c_int64 = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p,
                                                ctypes.c_int64)


class TensorDesc(ctypes.Structure):
    _fields_ = [('name', c_char_p), ('data', c_void_p), ('ndim', c_int), ('shape', c_int64 * 4)]


class KernelStat(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 64), ('launches', c_int), ('total_ms', ctypes.c_double),
                ('flops', ctypes.c_double), ('bytes', ctypes.c_double)]


# name -> (restype, argtypes); every symbol of include/ocrhip.h
SIGNATURES = {
    'ocrhip_version': (c_int, []),
    'ocrhip_last_error': (c_char_p, []),
    'ocrhip_create': (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    'ocrhip_destroy': (c_int, [c_void_p]),
    'ocrhip_east_load_weights': (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_int]),
    'ocrhip_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'ocrhip_dice_coeff': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    'ocrhip_profile_only': (c_int, [c_void_p, c_char_p]),
    'ocrhip_image_resize_norm': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ocrhip_east_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ocrhip_east_get_activation': (c_int, [c_void_p, c_char_p, c_void_p, c_int64, ctypes.POINTER(c_int64)]),
    'ocrhip_east_decode': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ocrhip_east_decode_static': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ocrhip_lanms': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_int,
                             c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ocrhip_crop_resize_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'ocrhip_crop_rotated_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'ocrhip_rotated_nms': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    'ocrhip_crnn_load_weights': (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_int]),
    'ocrhip_crnn_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ocrhip_crnn_vocab_size': (c_int, [c_void_p]),
    'ocrhip_crnn_seq_len': (c_int, [c_void_p, c_int]),
    'ocrhip_crnn_forward_packed': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    'ocrhip_check_errors': (c_int, [c_void_p, c_void_p, c_int]),
    'ocrhip_crnn_get_activation': (c_int, [c_void_p, c_char_p, c_void_p, c_int64, ctypes.POINTER(c_int64)]),
    'ocrhip_ctc_greedy': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    'ocrhip_ctc_beam_search': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p]),
    'ocrhip_profile_enable': (c_int, [c_void_p, c_int]),
    'ocrhip_profile_collect': (c_int, [c_void_p, ctypes.POINTER(KernelStat), c_int, ctypes.POINTER(c_int)]),
}

_lib = None


def load():
    """Loads the shared library once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # The library shares the process with PyTorch-ROCm (its tensors and streams are passed in), so
    # both must run on ONE HIP runtime: import torch first, the loader then binds libocrhip.so to
    # the libamdhip64 torch has already loaded.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libocrhip.so not found at {}.  Build it with `python -m ocr_amd.csrc.build` '
            '(there is no CPU fallback).'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class OcrHipError(RuntimeError):
    pass


def check(rc):
    if rc == OCRHIP_OK:
        return
    msg = load().ocrhip_last_error()
    msg = msg.decode('utf-8', 'replace') if msg else 'error {}'.format(rc)
    if rc == OCRHIP_EINVAL:
        raise ValueError(msg)
    raise OcrHipError('{} (code {})'.format(msg, rc))
