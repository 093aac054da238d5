"""`EAST` detector with the reference's interface (models/detection/east.py:19-95,
models/detection/base_detector.py:26-81), running on libocrhip.so."""
import numpy as np
import torch

from .. import register_model
from ...runtimes import HIPRuntime
from ...utils import boxes as box_utils
from ...utils.image_sizes import east_input_size


def load_image(filename):
    """utils/image/image_io.py:108-120 for the non-TensorFlow branch: PIL decode, arrays pass through."""
    if isinstance(filename, dict):
        filename = filename['image' if 'image' in filename else 'filename']
    if isinstance(filename, str):
        import PIL.Image
        return np.array(PIL.Image.open(filename).convert('RGB'))
    image = filename.cpu().numpy() if isinstance(filename, torch.Tensor) else np.asarray(filename)
    if image.ndim == 2:
        image = image[:, :, None]
    return image


def get_image_data(data):
    """models/interfaces/base_image_model.py:147-159."""
    filename = None
    if isinstance(data, dict):
        for k in ('tf_image', 'image', 'image_copy', 'filename'):
            if k in data:
                if 'filename' in data:
                    filename = data['filename']
                data = data[k]
                break
    if isinstance(data, str):
        return data, load_image(data)
    return filename, data


class EAST:
    def __init__(self, weights=None, *, name='east', obj_threshold=0.35, nms_threshold=0.2, image_normalization='east',
                 resize_kwargs=None, runtime='hip', device=0, merge_method='union', merge_threshold=0.3, **kwargs):
        if runtime != 'hip':
            raise ValueError('Unsupported runtime !\n  Accepted : {}\n  Got : {}'.format(('hip',), runtime))
        self.name = name
        self.obj_threshold = obj_threshold
        self.nms_threshold = nms_threshold
        self.image_normalization = image_normalization
        self.resize_kwargs = resize_kwargs or {'antialias': True}
        self.merge_method = merge_method
        self.merge_threshold = merge_threshold
        self.runtime = runtime
        if isinstance(weights, dict):
            self.model = HIPRuntime('state_dict:' + name, state_dict=weights, device=device, reload=True)
        else:
            self.model = HIPRuntime(weights if weights is not None else 'random:east:0', device=device)
        self.ctx = self.model.engine.ctx
        register_model(name, self)

    @property
    def compiled_infer(self):
        return self.model                        # base_model.py:365-374 for runtime != 'keras'

    # ---- input -------------------------------------------------------------------------------
    def get_input(self, image):
        """Network input for one image: float32 normalised (H',W',3) with H', W' multiples of 32
        (base_image_model.py:93-131).  uint8 images whose sides already are multiples of 32 are
        returned unchanged: the `/255`, `(x-0.5)/0.5` steps are fused into the first convolution."""
        image = load_image(image)
        h, w = image.shape[:2]
        oh, ow = east_input_size(h, w)
        if image.dtype == np.uint8 and (oh, ow) == (h, w) and image.shape[2] == 3:
            return torch.from_numpy(np.ascontiguousarray(image)).to(self.ctx.device)
        x = torch.from_numpy(np.ascontiguousarray(image)).to(self.ctx.device)
        if x.dtype != torch.uint8:
            x = x.float()
        if x.shape[2] == 1:
            x = x.expand(-1, -1, 3)
        # `/255` (uint8), stretch to the next multiples of 32 (image_processing.py:187-199, antialiased bilinear;
        # the identity when the size already fits) and `(x-0.5)/0.5`, in one kernel of the library
        return self.ctx.image_resize_norm(x[None].contiguous(), oh, ow)[0]

    # ---- inference ---------------------------------------------------------------------------
    def infer(self, image, *, predicted=None, overwrite=False, output_on_device=False, **kwargs):
        """base_detector.py:51-81: returns dict(filename, boxes, image, output).  `output` is the (mh,mw,6) map as a
        NumPy array, or the GPU tensor itself with `output_on_device` (what the stream's frame comparison reads)."""
        if predicted and not overwrite and isinstance(image, str) and image in predicted:
            return predicted[image]
        infos = image.copy() if isinstance(image, dict) else {}
        filename, image = get_image_data(image)
        inputs = self.get_input(image)
        if inputs.dim() == 3:
            inputs = inputs[None]
        output = self.compiled_infer(inputs, **kwargs)[0]
        boxes = self.decode_output(output, inputs=inputs, **kwargs)
        return {**infos, 'filename': filename, 'boxes': boxes, 'image': image,
                'output': output if output_on_device else output.cpu().numpy()}

    def decode_output(self, output, inputs, normalize=True, nms_method='lanms', nms_threshold=None,
                      obj_threshold=None, **kwargs):
        """models/detection/east.py:64-95: RBOX decode then NMS; `{'boxes': (M,4), 'format': 'xyxy'}`."""
        if obj_threshold is None:
            obj_threshold = self.obj_threshold
        if nms_threshold is None:
            nms_threshold = self.nms_threshold
        if not normalize:
            raise NotImplementedError('decode_output(normalize=False) is not supported')
        maps = output if isinstance(output, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(output, dtype=np.float32))
        maps = maps.to(self.ctx.device).float().contiguous()
        if maps.dim() == 3:
            maps = maps[None]
        in_hw = tuple(int(s) for s in inputs.shape[1:-1]) if len(inputs.shape) == 4 else tuple(int(s) for s in inputs.shape[:2])
        use_nms = nms_threshold < 1.
        dec = self.ctx.east_decode(maps, in_hw, obj_threshold, with_polys=not use_nms)
        n = int(dec['counts'][0].item())
        if n == 0:
            return {'boxes': np.zeros((0, 4, 2), np.float32), 'scores': np.zeros((0,), np.float32), 'format': 'poly'}
        if not use_nms:
            return {'boxes': dec['polys'][0, :n].cpu().numpy(), 'scores': dec['scores'][0, :n].cpu().numpy(),
                    'format': 'poly'}
        if nms_method in ('rotated', 'rotated_nms'):
            # opt-in, not in the reference: the quadrilaterals keep their rotation through a greedy NMS with rotated IoU
            dec = self.ctx.east_decode(maps, in_hw, obj_threshold, with_polys=True)
            out = self.ctx.rotated_nms(dec['polys'], dec['scores'], dec['counts'], nms_threshold,
                                       max_out=kwargs.get('max_output_size') or 2048)
            m = int(out['counts'][0].item())
            return {'boxes': out['polys'][0, :m].cpu().numpy(), 'scores': out['scores'][0, :m].cpu().numpy(),
                    'format': 'poly'}
        if nms_method in ('lanms', 'locality_aware_nms'):
            out = self.ctx.lanms(dec['xyxy'], dec['scores'], dec['counts'],
                                 merge_threshold=kwargs.get('merge_threshold', self.merge_threshold),
                                 nms_threshold=nms_threshold, merge_method=kwargs.get('merge_method', self.merge_method),
                                 max_iter=kwargs.get('max_iter') or 0, max_out=kwargs.get('max_output_size') or 0)
            m = int(out['counts'][0].item())
            return {'boxes': out['boxes'][0, :m].cpu().numpy(), 'format': 'xyxy'}
        boxes, _, mask = box_utils.nms(dec['xyxy'][0, :n].cpu().numpy(), dec['scores'][0, :n].cpu().numpy(),
                                       method=nms_method, nms_threshold=nms_threshold, **kwargs)
        return {'boxes': np.asarray(boxes)[np.asarray(mask)], 'format': 'xyxy'}

    def predict(self, inputs, **kwargs):
        if isinstance(inputs, (str, dict)) or (hasattr(inputs, 'shape') and len(inputs.shape) == 3):
            inputs = [inputs]
        return [self.infer(inp, **kwargs) for inp in inputs]
