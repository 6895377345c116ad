"""Oracle: FCN-8s forward (reference models/fcn8.py:16-200).  TEST INFRASTRUCTURE.

`params` is a dict  name -> (W, b)  in the reference's (Lasagne) layouts: Conv2DLayer
W[out,in,kh,kw]; Deconv2DLayer W[in,out,kh,kw].  PARAM_ORDER is the order of
lasagne.layers.get_all_param_values(net['probs']) (P14), i.e. the `arr_%d` order of
fcn8_model.npz (models/fcn8.py:178-180).
"""
from . import nn

PARAM_ORDER = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3',
               'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3',
               'fc6', 'fc7', 'score_fr', 'score2', 'score_pool4', 'score4', 'score_pool3',
               'upsample']


def fcn8_forward(params, x, layer=('probs_dimshuffle',), pad=100, temperature=1.0,
                 concat_h=(), h_list=()):
    """Returns [net[el] for el in layer] (models/fcn8.py:200) for input x (B,C,H,W) in [0,1].

    With `concat_h` / `h_list` this is buildFCN8_DAE (models/fcn8_dae.py:19-171): the same
    FCN-8 applied to y, with h concatenated (h first, P13) at the input and/or after pool k
    (`model_helpers.concatenate` calls at fcn8_dae.py:52-54,63-65,...); score_pool4 /
    score_pool3 read the PRE-concat pools (:141-142,:150-151).

    Deterministic pass: DropoutLayer is the identity (P8, fcn8.py:77,82).  Note P2: score_fr,
    score_pool4 and score_pool3 keep Lasagne's default ReLU (fcn8.py:84-85,92-93,102-103).
    `temperature` divides upsample.W and upsample.b (fcn8.py:194-198).
    """
    net = {'input': x}
    hs = dict(zip(concat_h, h_list))
    cat = lambda name, t: nn.concat_h_first(hs[name], t) if name in hs else t
    c = lambda name, t, p: nn.conv2d(t, params[name][0], params[name][1], pad=p, relu=True)
    t = c('conv1_1', cat('input', x), pad)       # fcn8.py:34-35  pad=100
    t = c('conv1_2', t, 1)                       # :36-37
    net['pool1'] = t = nn.maxpool2(t)            # :38
    t = c('conv2_1', cat('pool1', t), 1)
    t = c('conv2_2', t, 1)
    net['pool2'] = t = nn.maxpool2(t)            # :45
    t = c('conv3_1', cat('pool2', t), 1)
    t = c('conv3_2', t, 1)
    t = c('conv3_3', t, 1)
    net['pool3'] = t = nn.maxpool2(t)            # :54
    t = c('conv4_1', cat('pool3', t), 1)
    t = c('conv4_2', t, 1)
    t = c('conv4_3', t, 1)
    net['pool4'] = t = nn.maxpool2(t)            # :63
    t = c('conv5_1', cat('pool4', t), 1)
    t = c('conv5_2', t, 1)
    t = c('conv5_3', t, 1)
    net['pool5'] = t = nn.maxpool2(t)            # :72
    t = cat('pool5', t)
    t = c('fc6', t, 0)                           # :75-76  7x7 valid (+ReLU), dropout = identity
    t = c('fc7', t, 0)                           # :80-81  1x1
    t = c('score_fr', t, 0)                      # :84-85  1x1, default nonlinearity = ReLU (P2)
    score2 = nn.deconv2d(t, *params['score2'], stride=2)                 # :90-91
    score_pool4 = c('score_pool4', net['pool4'], 0)                      # :92-93 (1x1, ReLU)
    fused = nn.crop_sum(score2, score_pool4)                             # :94-97
    score4 = nn.deconv2d(fused, *params['score4'], stride=2)             # :100-101
    score_pool3 = c('score_pool3', net['pool3'], 0)                      # :102-103
    final = nn.crop_sum(score4, score_pool3)                             # :104-107
    Wu, bu = params['upsample']
    up = nn.deconv2d(final, Wu / temperature, bu / temperature, stride=8)  # :109-110,194-198
    net['score'] = score = nn.crop_like(x, up)                           # :115-119
    net['probs_dimshuffle'] = nn.softmax_channels(score)                 # :122-130,187-191
    return [net[el] for el in layer]
