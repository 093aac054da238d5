"""EAST-VGG16 and CRNN-CTC graphs restated with torch-CPU float32 (test infrastructure).

Follows `architectures/east_arch.py:95-175` and `architectures/crnn_arch.py:38-41,223-226,
306-531` (+ `architectures/current_blocks.py:213-358` for the ResNet blocks).  Keras is not installed here, so the graphs are rebuilt from the layer tables
with torch functional ops: `conv2d` on explicitly zero-padded inputs, inference batch-norm with
eps 1e-5, `interpolate(bilinear, align_corners=True)` for the EAST upsampling (what the layer
itself calls on the torch backend, east_arch.py:44-48), and an explicit LSTM recurrence in Keras
gate order i, f, c, o (= torch's i, f, g, o) with the merged bias b_ih + b_hh
(models/weights_converter.py:305-314).  Parity unpinned against Keras itself; the parameter
counts printed in the reference notebook (15.099 M and 3 778 273) are reproduced.

Parameters are held in dicts keyed by the names of the original torch checkpoints the reference
converts from: SakuraRiven's `east_vgg16.pth` (`extractor.features.N.*`, `merge.convK/bnK.*`,
`output.convK.*`) and EasyOCR's recognisers (`FeatureExtraction.ConvNet.N.*`,
`SequenceModeling.L.rnn.*`, `SequenceModeling.L.linear.*`, `Prediction.*`), all in torch layout
(conv OIHW, linear (out,in), LSTM (4H,in) rows ordered i,f,g,o).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

EAST_VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
# (cin, cout, kernel) of merge.conv1..7 (east_arch.py:135-151)
EAST_MERGE_CFG = [(1024, 128, 1), (128, 128, 3), (384, 64, 1), (64, 64, 3),
                  (192, 32, 1), (32, 32, 3), (32, 32, 3)]
EAST_HEAD_CFG = [('output.conv1', 1), ('output.conv2', 4), ('output.conv3', 1)]

# crnn_arch.py:38-41 : (index in EasyOCR's nn.Sequential, cin, cout, kernel, bias, bn index or None)
CRNN_VGG_CFG = [
    ('conv', 0, 1, 32, 3, True, None), ('pool', (2, 2)),
    ('conv', 3, 32, 64, 3, True, None), ('pool', (2, 2)),
    ('conv', 6, 64, 128, 3, True, None),
    ('conv', 8, 128, 128, 3, True, None), ('pool', (2, 1)),
    ('conv', 11, 128, 256, 3, False, 12),
    ('conv', 14, 256, 256, 3, False, 15), ('pool', (2, 1)),
    ('conv', 18, 256, 256, 2, True, None),
]


def east_vgg_conv_indices():
    """Indices of the conv layers inside torchvision's vgg16_bn `features` Sequential."""
    idx, out = 0, []
    for c in EAST_VGG_CFG:
        if c == 'M':
            idx += 1
        else:
            out.append((idx, c))
            idx += 3
    return out


def _he_conv(rng, cout, cin, k):
    std = math.sqrt(2.0 / (cin * k * k))
    return torch.from_numpy((rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32))


def _bn(rng, c, prefix, params):
    params[prefix + '.weight'] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
    params[prefix + '.bias'] = torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32))
    params[prefix + '.running_mean'] = torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32))
    params[prefix + '.running_var'] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))


def init_east_params(seed=0):
    """Seeded random EAST parameters: He-normal convs, BN gamma in U[0.5,1.5], beta/mean ~ N(0,0.1),
    var in U[0.5,1.5]; the head convs are scaled down so the sigmoids do not saturate."""
    rng = np.random.default_rng(seed)
    p = {}
    cin = 3
    for idx, cout in east_vgg_conv_indices():
        p['extractor.features.%d.weight' % idx] = _he_conv(rng, cout, cin, 3)
        p['extractor.features.%d.bias' % idx] = torch.from_numpy((rng.standard_normal(cout) * 0.05).astype(np.float32))
        _bn(rng, cout, 'extractor.features.%d' % (idx + 1), p)
        cin = cout
    for k, (ci, co, ks) in enumerate(EAST_MERGE_CFG, start=1):
        p['merge.conv%d.weight' % k] = _he_conv(rng, co, ci, ks)
        p['merge.conv%d.bias' % k] = torch.from_numpy((rng.standard_normal(co) * 0.05).astype(np.float32))
        _bn(rng, co, 'merge.bn%d' % k, p)
    for name, co in EAST_HEAD_CFG:
        p[name + '.weight'] = _he_conv(rng, co, 32, 1) * 0.5
        p[name + '.bias'] = torch.from_numpy((rng.standard_normal(co) * 0.2).astype(np.float32))
    return p


def count_params(params):
    return int(sum(int(v.numel()) for v in params.values()))


def _conv_bn_relu(x, w, b, bn, pad):
    if pad:
        x = F.pad(x, (pad, pad, pad, pad))
    x = F.conv2d(x, w, b)
    if bn is not None:
        x = F.batch_norm(x, bn[2], bn[3], bn[0], bn[1], training=False, eps=BN_EPS)
    return F.relu(x)


def _bn_of(p, prefix):
    return (p[prefix + '.weight'], p[prefix + '.bias'], p[prefix + '.running_mean'], p[prefix + '.running_var'])


@torch.no_grad()
def east_forward(params, x_nhwc, return_taps=False):
    """(B,H,W,3) normalised float32 -> (B,H/4,W/4,6) = [score, top, bottom, left, right, angle]."""
    x = torch.as_tensor(x_nhwc, dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    taps = []
    idx = 0
    for c in EAST_VGG_CFG:
        if c == 'M':
            x = F.max_pool2d(x, 2)
            taps.append(x)
            idx += 1
        else:
            x = _conv_bn_relu(x, params['extractor.features.%d.weight' % idx],
                              params['extractor.features.%d.bias' % idx],
                              _bn_of(params, 'extractor.features.%d' % (idx + 1)), 1)
            idx += 3
    residuals = taps[1:-1][::-1]                        # p4, p3, p2
    k = 1
    for i, res in enumerate(residuals):
        x = F.interpolate(x, size=(x.shape[2] * 2, x.shape[3] * 2), mode='bilinear', align_corners=True)
        x = torch.cat([x, res], dim=1)
        for j in range(2 if i < len(residuals) - 1 else 3):
            ks = 1 if j == 0 else 3
            x = _conv_bn_relu(x, params['merge.conv%d.weight' % k], params['merge.conv%d.bias' % k],
                              _bn_of(params, 'merge.bn%d' % k), ks // 2)
            k += 1
    outs = [torch.sigmoid(F.conv2d(x, params[name + '.weight'], params[name + '.bias']))
            for name, _ in EAST_HEAD_CFG]
    out = torch.cat(outs, dim=1).permute(0, 2, 3, 1).contiguous()
    if return_taps:
        return out, [t.permute(0, 2, 3, 1).contiguous() for t in taps], x.permute(0, 2, 3, 1).contiguous()
    return out


# ----------------------------------------------------------------------------- CRNN

# ResNet extractor (crnn_arch.py:367-425 with its defaults): blocks per stage and channels of layer1..4
RESNET_BLOCKS = [1, 2, 5, 3]
RESNET_CHANNELS = [128, 256, 512, 512]
RESNET_BLOCK_BN_EPS = 1e-3        # Conv2DBN default epsilon (current_blocks.py:222); BasicBlock does not override it


def resnet_conv_table():
    """(conv name, bn name, cin, cout, kernel) of every convolution, EasyOCR ResNet_FeatureExtractor keys."""
    pre = 'FeatureExtraction.ConvNet.'
    out = [(pre + 'conv0_1', pre + 'bn0_1', 1, 32, 3), (pre + 'conv0_2', pre + 'bn0_2', 32, 64, 3)]
    cin = 64
    for st, (nb, c) in enumerate(zip(RESNET_BLOCKS, RESNET_CHANNELS), start=1):
        lay = pre + 'layer%d.' % st
        if cin != c:
            out.append((lay + '0.downsample.0', lay + '0.downsample.1', cin, c, 1))
        for b in range(nb):
            out.append((lay + '%d.conv1' % b, lay + '%d.bn1' % b, cin if b == 0 else c, c, 3))
            out.append((lay + '%d.conv2' % b, lay + '%d.bn2' % b, c, c, 3))
        if st < 4:
            out.append((pre + 'conv%d' % st, pre + 'bn%d' % st, c, c, 3))
        cin = c
    out += [(pre + 'conv4_1', pre + 'bn4_1', 512, 512, 2), (pre + 'conv4_2', pre + 'bn4_2', 512, 512, 2)]
    return out


def init_crnn_params(vocab_size=97, hidden=256, seed=0, backbone='vgg'):
    """Seeded random CRNN (VGG or ResNet feature extractor, 2 x BiLSTM(hidden) + Linear, Linear(vocab))."""
    rng = np.random.default_rng(seed)
    p = {}
    if backbone == 'resnet':
        for conv, bn, ci, co, k in resnet_conv_table():
            p[conv + '.weight'] = _he_conv(rng, co, ci, k)
            _bn(rng, co, bn, p)
    else:
        for item in CRNN_VGG_CFG:
            if item[0] != 'conv':
                continue
            _, idx, ci, co, k, bias, bn = item
            p['FeatureExtraction.ConvNet.%d.weight' % idx] = _he_conv(rng, co, ci, k)
            if bias:
                p['FeatureExtraction.ConvNet.%d.bias' % idx] = torch.from_numpy(
                    (rng.standard_normal(co) * 0.05).astype(np.float32))
            if bn is not None:
                _bn(rng, co, 'FeatureExtraction.ConvNet.%d' % bn, p)
    cin = 512 if backbone == 'resnet' else 256
    for layer in range(2):
        for suffix in ('', '_reverse'):
            s = 1.0 / math.sqrt(hidden)
            pre = 'SequenceModeling.%d.rnn.' % layer
            p[pre + 'weight_ih_l0' + suffix] = torch.from_numpy(rng.uniform(-s, s, (4 * hidden, cin)).astype(np.float32))
            p[pre + 'weight_hh_l0' + suffix] = torch.from_numpy(rng.uniform(-s, s, (4 * hidden, hidden)).astype(np.float32))
            p[pre + 'bias_ih_l0' + suffix] = torch.from_numpy(rng.uniform(-s, s, 4 * hidden).astype(np.float32))
            p[pre + 'bias_hh_l0' + suffix] = torch.from_numpy(rng.uniform(-s, s, 4 * hidden).astype(np.float32))
        s = 1.0 / math.sqrt(2 * hidden)
        p['SequenceModeling.%d.linear.weight' % layer] = torch.from_numpy(
            rng.uniform(-s, s, (hidden, 2 * hidden)).astype(np.float32))
        p['SequenceModeling.%d.linear.bias' % layer] = torch.from_numpy(rng.uniform(-s, s, hidden).astype(np.float32))
        cin = hidden
    s = 1.0 / math.sqrt(hidden)
    # a wider head spreads the logits so that greedy argmax is not decided by rounding noise
    p['Prediction.weight'] = torch.from_numpy(rng.uniform(-s, s, (vocab_size, hidden)).astype(np.float32) * 4)
    p['Prediction.bias'] = torch.from_numpy(rng.uniform(-s, s, vocab_size).astype(np.float32))
    return p


def _conv_bn(x, params, conv, bn, eps, pad=0, stride=1):
    if pad:
        x = F.pad(x, (pad, pad, pad, pad))                # ZeroPadding2D + 'valid' (use_manual_padding)
    x = F.conv2d(x, params[conv + '.weight'], None, stride=stride)
    g, b, m, v = _bn_of(params, bn)
    return F.batch_norm(x, m, v, g, b, training=False, eps=eps)


@torch.no_grad()
def crnn_features_resnet(params, x_nhwc):
    """(B,64,W,1) -> (B,W/4+1,512): the ResNet extractor exactly as the reference graph builds it
    (crnn_arch.py:330-425): a skip connection only in the first block of layer1-3 (where a 1x1 downsample
    exists, crnn_arch.py:357), BasicBlock batch-norms with eps 1e-3, stage convolutions with eps 1e-5, zero
    padding of the width before the (2,1)-strided pooling and before conv4_1."""
    pre = 'FeatureExtraction.ConvNet.'
    x = torch.as_tensor(x_nhwc, dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    x = F.relu(_conv_bn(x, params, pre + 'conv0_1', pre + 'bn0_1', BN_EPS, pad=1))
    x = F.relu(_conv_bn(x, params, pre + 'conv0_2', pre + 'bn0_2', BN_EPS, pad=1))
    for st, (nb, c) in enumerate(zip(RESNET_BLOCKS, RESNET_CHANNELS), start=1):
        if st in (1, 2):
            x = F.max_pool2d(x, 2, 2)
        elif st == 3:
            x = F.max_pool2d(F.pad(x, (1, 1, 0, 0)), 2, (2, 1))
        lay = pre + 'layer%d.' % st
        residual = None
        if x.shape[1] != c:
            residual = _conv_bn(x, params, lay + '0.downsample.0', lay + '0.downsample.1', RESNET_BLOCK_BN_EPS)
        for b in range(nb):
            y = F.relu(_conv_bn(x, params, lay + '%d.conv1' % b, lay + '%d.bn1' % b, RESNET_BLOCK_BN_EPS, pad=1))
            y = _conv_bn(y, params, lay + '%d.conv2' % b, lay + '%d.bn2' % b, RESNET_BLOCK_BN_EPS, pad=1)
            if residual is not None and b == 0:
                y = y + residual
            x = F.relu(y)
        if st < 4:
            x = F.relu(_conv_bn(x, params, pre + 'conv%d' % st, pre + 'bn%d' % st, BN_EPS, pad=1))
    x = F.pad(x, (1, 1, 0, 0))
    x = F.relu(_conv_bn(x, params, pre + 'conv4_1', pre + 'bn4_1', BN_EPS, stride=(2, 1)))
    x = F.relu(_conv_bn(x, params, pre + 'conv4_2', pre + 'bn4_2', BN_EPS))
    x = x.mean(dim=2)
    return x.permute(0, 2, 1).contiguous()


@torch.no_grad()
def crnn_features(params, x_nhwc):
    """(B,64,W,1) -> (B,T,C): VGG extractor (C 256, T = W/4 - 1) or, when the parameters are those of the
    ResNet extractor, crnn_features_resnet (C 512, T = W/4 + 1); then the mean over the 3 remaining rows."""
    if 'FeatureExtraction.ConvNet.conv0_1.weight' in params:
        return crnn_features_resnet(params, x_nhwc)
    x = torch.as_tensor(x_nhwc, dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
    for item in CRNN_VGG_CFG:
        if item[0] == 'pool':
            x = F.max_pool2d(x, item[1])
            continue
        _, idx, ci, co, k, bias, bn = item
        w = params['FeatureExtraction.ConvNet.%d.weight' % idx]
        b = params.get('FeatureExtraction.ConvNet.%d.bias' % idx) if bias else None
        bnp = _bn_of(params, 'FeatureExtraction.ConvNet.%d' % bn) if bn is not None else None
        x = _conv_bn_relu(x, w, b, bnp, 1 if k == 3 else 0)
    x = x.mean(dim=2)                                   # GlobalAverage over H (crnn_arch.py:223-226)
    return x.permute(0, 2, 1).contiguous()              # (B,T,C)


def _lstm_direction(x, w_ih, w_hh, bias, reverse):
    """x (B,T,I) -> (B,T,H); gates i,f,g,o; sigmoid recurrent activation, tanh."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    xw = x @ w_ih.t() + bias
    out = x.new_zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        z = xw[:, t] + h @ w_hh.t()
        i, f, g, o = z.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


@torch.no_grad()
def crnn_sequence(params, feats):
    x = feats
    for layer in range(2):
        pre = 'SequenceModeling.%d.rnn.' % layer
        outs = []
        for suffix, rev in (('', False), ('_reverse', True)):
            bias = params[pre + 'bias_ih_l0' + suffix] + params[pre + 'bias_hh_l0' + suffix]
            outs.append(_lstm_direction(x, params[pre + 'weight_ih_l0' + suffix],
                                        params[pre + 'weight_hh_l0' + suffix], bias, rev))
        x = torch.cat(outs, dim=2)
        x = x @ params['SequenceModeling.%d.linear.weight' % layer].t() + params['SequenceModeling.%d.linear.bias' % layer]
    return x


@torch.no_grad()
def crnn_forward(params, x_nhwc):
    """(B,64,W,1) normalised float32 -> (B,T,V) logits (no softmax), T = W/4 - 1."""
    feats = crnn_features(params, x_nhwc)
    x = crnn_sequence(params, feats)
    return x @ params['Prediction.weight'].t() + params['Prediction.bias']
