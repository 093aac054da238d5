"""Model registry: process-wide singletons keyed by name, like the reference's `ModelInstances`
(models/interfaces/base_model.py:30-54) and `get_pretrained` (models/__init__.py:22-34)."""

_instances = {}


def register_model(name, model):
    _instances[name] = model
    return model


def get_pretrained(name, **kwargs):
    if isinstance(name, dict):
        name, kwargs = name.pop('name'), {**kwargs, **name}
    if name not in _instances:
        raise ValueError('Model {} does not exist or has an invalid `config.json` !'.format(name))
    return _instances[name]
