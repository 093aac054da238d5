"""`Runtime`-shaped engine objects: the drop-in seam of the reference's model framework.

The reference switches a model to a non-Keras back-end through `BaseModel(runtime=<key>)` ->
`build_runtime(runtime, path, **kwargs)` (models/interfaces/base_model.py:179-189,
utils/keras/runtimes/__init__.py:23-45); the object must look like
`utils/keras/runtimes/runtime.py:19-42`: `__init__(path, *, engine=None, reload=False, **kw)`,
`@staticmethod load_engine(path, **kw)`, `__call__(*args, **kwargs)`, class-level engine cache.
`HIPRuntime` is that object for libocrhip.so; registering it is one line on the reference side
(`_runtimes['hip'] = HIPRuntime` in utils/keras/runtimes/__init__.py).

`BaseOCR.infer` / `BaseDetector.infer` call the runtime as `model(inputs, **kwargs)[0]` with extra
keyword arguments (method=, num_beams=, ...) that a runtime must accept and ignore
(models/ocr/base_ocr.py:186-201, models/detection/base_detector.py:73-76).
"""
import numpy as np
import torch

from .engine import HipContext
from .weights import load_checkpoint, random_crnn_params, random_east_params


class HipEngine:
    """A HipContext with one network loaded."""

    def __init__(self, kind, ctx):
        self.kind = kind            # 'east' or 'crnn'
        self.ctx = ctx


class HIPRuntime:
    _engines = {}
    layers = []                     # describe.py walks `model.layers`; empty = factor 1

    def __init__(self, path, *, engine=None, reload=False, **kwargs):
        if engine is None:
            if path not in self._engines or reload:
                self._engines[path] = self.load_engine(path, **kwargs)
            engine = self._engines[path]
        self.path = path
        self.engine = engine

    def __repr__(self):
        return '<{} path={}>'.format(self.__class__.__name__, self.path)

    @staticmethod
    def load_engine(path, *, device=0, state_dict=None, ctx=None, **kwargs):
        """`path`: a torch checkpoint (SakuraRiven east_vgg16.pth or an EasyOCR recogniser .pth), or
        'random:east[:seed]' / 'random:crnn[:vocab[:seed[:backbone]]]' (backbone 'vgg' or 'resnet') for seeded
        random parameters."""
        if state_dict is None:
            if isinstance(path, str) and path.startswith('random:'):
                parts = path.split(':')
                if parts[1] == 'east':
                    state_dict = random_east_params(int(parts[2]) if len(parts) > 2 else 0)
                elif parts[1] == 'crnn':
                    state_dict = random_crnn_params(int(parts[2]) if len(parts) > 2 else 97,
                                                    seed=int(parts[3]) if len(parts) > 3 else 0,
                                                    backbone=parts[4] if len(parts) > 4 else 'vgg')
                else:
                    raise ValueError('Unknown random model : {}'.format(path))
            else:
                state_dict = load_checkpoint(path)
        keys = [k[len('module.'):] if k.startswith('module.') else k for k in state_dict.keys()]
        ctx = ctx or HipContext(device)
        if 'Prediction.weight' in keys:
            ctx.crnn_load_weights(state_dict)
            return HipEngine('crnn', ctx)
        if 'output.conv1.weight' in keys:
            ctx.east_load_weights(state_dict)
            return HipEngine('east', ctx)
        raise ValueError('Unsupported checkpoint : neither an EAST nor a CRNN state dict ({})'.format(path))

    def __call__(self, *args, **kwargs):
        """EAST: (B,H,W,3) normalised float32 (or raw uint8) -> (B,H/4,W/4,6).
        CRNN: (B,64,W,1) normalised float32 -> (B,T,V) logits.  Extra kwargs are ignored.
        NumPy in -> NumPy out; torch GPU tensor in -> torch GPU tensor out."""
        if not args:
            raise RuntimeError('HIPRuntime called without inputs')
        x = args[0]
        as_numpy = not isinstance(x, torch.Tensor)
        t = torch.from_numpy(np.ascontiguousarray(x)) if as_numpy else x
        t = t.to(self.engine.ctx.device).contiguous()
        if self.engine.kind == 'east':
            if t.dtype not in (torch.uint8, torch.float32):
                t = t.float()
            out = self.engine.ctx.east_forward(t)
        else:
            out = self.engine.ctx.crnn_forward(t.float().contiguous(), want_logits=True)['logits']
        if as_numpy:
            out = out.cpu().numpy()
            self.engine.ctx.check_errors(synchronize=False)     # a failed recurrence raises instead of returning garbage
        return out


_runtimes = {'hip': HIPRuntime}


def build_runtime(runtime, path, *args, **kwargs):
    if runtime not in _runtimes:
        raise ValueError('Unsupported runtime !\n  Accepted : {}\n  Got : {}'.format(tuple(_runtimes.keys()), runtime))
    return _runtimes[runtime](path, *args, **kwargs)
