"""Public OCR API with the reference's names (models/ocr/__init__.py:29-56)."""
from .base_ocr import BaseOCR, CRNN  # noqa: F401

_pretrained = {'en': 'crnn_en'}


def get_model(model=None, lang=None):
    assert model is not None or lang is not None
    if model is None:
        if lang not in _pretrained:
            raise ValueError('No default model for language {}'.format(lang))
        model = _pretrained[lang]
    if isinstance(model, str):
        from .. import get_pretrained
        model = get_pretrained(model)
    return model


def ocr(image, *, model=None, lang='en', **kwargs):
    return get_model(model=model, lang=lang).predict(image, **kwargs)


def stream(stream, *, model=None, lang='en', **kwargs):
    return get_model(model=model, lang=lang).stream(stream, **kwargs)
