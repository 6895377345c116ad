"""Oracle: context-module DAE (reference models/contextmod_dae.py:19-138).  TEST INFRASTRUCTURE.

params: 'conv1' (Conv2DLayer W[out,in,3,3]) and 'dilconv1'..'dilconv7' (DilatedConv2DLayer
W[in,out,k,k], P11), each with a bias.  PARAM_ORDER = get_all_param_values order (P14).
"""
import numpy as np

from . import nn

PARAM_ORDER = ['conv1'] + ['dilconv%d' % i for i in range(1, 8)]
DILATIONS = [1, 2, 4, 8, 16, 1]          # contextmod_dae.py:78-101


def contextmod_forward(params, h_list, y, concat_h=('input',), out_softmax=True):
    """pred_dae_fn(h, y) -> r for dae kind 'contextmod'.  concat_h must be all 'input' (:42);
    h (the image, 3 channels, :59) is concatenated first (P13)."""
    assert all(el in ['input'] for el in concat_h)
    t = y
    for h in reversed(list(h_list)):
        t = nn.concat_h_first(h, t)
    W, b = params['conv1']
    t = nn.conv2d(t, W, b, pad=1, relu=True)                       # :74-76
    pad = 32                                                       # PadLayer(width=32), :77 (P12)
    for i, d in enumerate(DILATIONS):                              # :78-101
        W, b = params['dilconv%d' % (i + 1)]
        t = nn.conv2d(t, np.transpose(W, (1, 0, 2, 3)), b, pad=pad, dilation=d, relu=True)
        pad = 0
    W, b = params['dilconv7']                                      # :102-105  1x1 linear
    t = nn.conv2d(t, np.transpose(W, (1, 0, 2, 3)), b, pad=0, relu=False)
    return nn.softmax_channels(t) if out_softmax else t            # :107-122
