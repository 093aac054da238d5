"""Network parameters: loading of the original torch checkpoints and seeded random parameters of
the same architectures (for tests and benchmarks; no trained weights are available offline).

The state-dict keys are those of the checkpoints the reference converts from
(architectures/east_arch.py:165-173 -> SakuraRiven's east_vgg16.pth; architectures/crnn_arch.py:
795-813 -> EasyOCR's recogniser .pth files), which the C library consumes directly
(layout facts: models/weights_converter.py:252-322).
"""
import math

import numpy as np
import torch

EAST_VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
EAST_MERGE_CFG = [(1024, 128, 1), (128, 128, 3), (384, 64, 1), (64, 64, 3), (192, 32, 1), (32, 32, 3), (32, 32, 3)]
# (index in the nn.Sequential, cin, cout, kernel, has bias, batch-norm index)
CRNN_VGG_CONVS = [(0, 1, 32, 3, True, None), (3, 32, 64, 3, True, None), (6, 64, 128, 3, True, None),
                  (8, 128, 128, 3, True, None), (11, 128, 256, 3, False, 12), (14, 256, 256, 3, False, 15),
                  (18, 256, 256, 2, True, None)]


def load_checkpoint(path):
    """Reads a torch checkpoint without executing anything from the file."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    if isinstance(sd, dict) and 'state_dict' in sd:
        sd = sd['state_dict']
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _he(rng, cout, cin, k):
    return _t(rng.standard_normal((cout, cin, k, k)) * math.sqrt(2.0 / (cin * k * k)))


def _bn(rng, c, prefix, p):
    p[prefix + '.weight'] = _t(rng.uniform(0.5, 1.5, c))
    p[prefix + '.bias'] = _t(rng.standard_normal(c) * 0.1)
    p[prefix + '.running_mean'] = _t(rng.standard_normal(c) * 0.1)
    p[prefix + '.running_var'] = _t(rng.uniform(0.5, 1.5, c))


def random_east_params(seed=0):
    rng = np.random.default_rng(seed)
    p, cin, idx = {}, 3, 0
    for c in EAST_VGG_CFG:
        if c == 'M':
            idx += 1
            continue
        p['extractor.features.%d.weight' % idx] = _he(rng, c, cin, 3)
        p['extractor.features.%d.bias' % idx] = _t(rng.standard_normal(c) * 0.05)
        _bn(rng, c, 'extractor.features.%d' % (idx + 1), p)
        cin, idx = c, idx + 3
    for k, (ci, co, ks) in enumerate(EAST_MERGE_CFG, start=1):
        p['merge.conv%d.weight' % k] = _he(rng, co, ci, ks)
        p['merge.conv%d.bias' % k] = _t(rng.standard_normal(co) * 0.05)
        _bn(rng, co, 'merge.bn%d' % k, p)
    for k, co in ((1, 1), (2, 4), (3, 1)):
        p['output.conv%d.weight' % k] = _he(rng, co, 32, 1) * 0.5
        p['output.conv%d.bias' % k] = _t(rng.standard_normal(co) * 0.2)
    return p


def resnet_convs():
    """(conv key, bn key, cin, cout, kernel) of the ResNet extractor (architectures/crnn_arch.py:367-425) under
    the keys of EasyOCR's ResNet_FeatureExtractor."""
    pre = 'FeatureExtraction.ConvNet.'
    out = [(pre + 'conv0_1', pre + 'bn0_1', 1, 32, 3), (pre + 'conv0_2', pre + 'bn0_2', 32, 64, 3)]
    cin = 64
    for st, (nb, c) in enumerate(zip((1, 2, 5, 3), (128, 256, 512, 512)), start=1):
        lay = pre + 'layer%d.' % st
        if cin != c:
            out.append((lay + '0.downsample.0', lay + '0.downsample.1', cin, c, 1))
        for b in range(nb):
            out.append((lay + '%d.conv1' % b, lay + '%d.bn1' % b, cin if b == 0 else c, c, 3))
            out.append((lay + '%d.conv2' % b, lay + '%d.bn2' % b, c, c, 3))
        if st < 4:
            out.append((pre + 'conv%d' % st, pre + 'bn%d' % st, c, c, 3))
        cin = c
    return out + [(pre + 'conv4_1', pre + 'bn4_1', 512, 512, 2), (pre + 'conv4_2', pre + 'bn4_2', 512, 512, 2)]


def random_crnn_params(vocab_size=97, hidden=256, seed=0, backbone='vgg'):
    rng = np.random.default_rng(seed)
    p = {}
    if backbone == 'resnet':
        for conv, bn, ci, co, k in resnet_convs():
            p[conv + '.weight'] = _he(rng, co, ci, k)
            _bn(rng, co, bn, p)
    elif backbone == 'vgg':
        for idx, ci, co, k, bias, bn in CRNN_VGG_CONVS:
            p['FeatureExtraction.ConvNet.%d.weight' % idx] = _he(rng, co, ci, k)
            if bias:
                p['FeatureExtraction.ConvNet.%d.bias' % idx] = _t(rng.standard_normal(co) * 0.05)
            if bn is not None:
                _bn(rng, co, 'FeatureExtraction.ConvNet.%d' % bn, p)
    else:
        raise ValueError('unknown backbone %r' % (backbone,))
    cin = 512 if backbone == 'resnet' else 256
    for layer in range(2):
        s = 1.0 / math.sqrt(hidden)
        for suffix in ('', '_reverse'):
            pre = 'SequenceModeling.%d.rnn.' % layer
            p[pre + 'weight_ih_l0' + suffix] = _t(rng.uniform(-s, s, (4 * hidden, cin)))
            p[pre + 'weight_hh_l0' + suffix] = _t(rng.uniform(-s, s, (4 * hidden, hidden)))
            p[pre + 'bias_ih_l0' + suffix] = _t(rng.uniform(-s, s, 4 * hidden))
            p[pre + 'bias_hh_l0' + suffix] = _t(rng.uniform(-s, s, 4 * hidden))
        s2 = 1.0 / math.sqrt(2 * hidden)
        p['SequenceModeling.%d.linear.weight' % layer] = _t(rng.uniform(-s2, s2, (hidden, 2 * hidden)))
        p['SequenceModeling.%d.linear.bias' % layer] = _t(rng.uniform(-s2, s2, hidden))
        cin = hidden
    s = 1.0 / math.sqrt(hidden)
    p['Prediction.weight'] = _t(rng.uniform(-s, s, (vocab_size, hidden)) * 4)
    p['Prediction.bias'] = _t(rng.uniform(-s, s, vocab_size))
    return p
