from .east import EAST  # noqa: F401


def get_model(model=None):
    from .. import get_pretrained
    return get_pretrained(model or 'east') if isinstance(model or 'east', str) else model


def detect(image, *, model=None, **kwargs):
    """models/detection/__init__.py: detect()."""
    return get_model(model).infer(image, **kwargs)
