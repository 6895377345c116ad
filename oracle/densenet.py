"""Oracle: FC-DenseNet103 forward (reference models/FCDenseNet.py:15-146,196-219).
TEST INFRASTRUCTURE.

The block internals come from the un-vendored, unpinned `FC_DenseNet.layers` (SimJeg/FC-DenseNet,
models/FCDenseNet.py:12); they are restated from upstream and NOT verifiable here:
  BN_ReLU_Conv(x, n, k=3)  = Conv2DLayer(rectify(BatchNormLayer(x)), n, k, pad='same', linear,
                             flip_filters=False) [+ Dropout(0.2) = identity at test time]
  TransitionDown(x, n)     = Pool2DLayer(BN_ReLU_Conv(x, n, k=1), 2, mode='max')
  TransitionUp(skip, blk, keep) = ConcatLayer([Deconv2DLayer(concat(blk), keep, 3, stride=2,
                             crop='valid', linear), skip], cropping center)
  SoftmaxLayer(x, C)       = softmax over channels of Conv2DLayer(x, C, 1, linear)
BatchNorm runs with batch_norm_use_averages=False (iterative_inference.py:187): batch mean and
biased variance over (B,H,W), eps 1e-4 (P10): (x - mean) * (gamma * inv_std) + beta.

`params` is a list of dicts in creation order (see `layer_plan`), each {'kind', ...arrays}.
"""
import numpy as np

from . import nn

GROWTH = 16
N_POOL = 5
LAYERS_PER_BLOCK = [4, 5, 7, 10, 12, 15, 12, 10, 7, 5, 4]      # FCDenseNet.py:208
N_FILTERS_FIRST = 48


def layer_plan(n_layers_per_block=LAYERS_PER_BLOCK, n_pool=N_POOL, growth=GROWTH,
               n_first=N_FILTERS_FIRST, nb_in_channels=3, n_classes=11):
    """Creation-order list of (kind, cin, cout) for every parametrised layer:
    'first' conv3x3, 'brc' (BN + conv3x3 -> growth), 'td' (BN + conv1x1), 'tu' (deconv3x3 s2),
    'softmax' (conv1x1).  103 convolutions in total for the default configuration."""
    plan = [('first', nb_in_channels, n_first)]
    n = n_first
    skips = []
    for i in range(n_pool):                                   # FCDenseNet.py:81-97
        for _ in range(n_layers_per_block[i]):
            plan.append(('brc', n, growth))
            n += growth
        skips.append(n)
        plan.append(('td', n, n))
    skips = skips[::-1]
    nblock = 0
    for _ in range(n_layers_per_block[n_pool]):               # bottleneck, :107-111
        plan.append(('brc', n, growth))
        n += growth
        nblock += 1
    for i in range(n_pool):                                   # :116-127
        keep = growth * n_layers_per_block[n_pool + i]
        plan.append(('tu', growth * nblock, keep))
        n = keep + skips[i]
        nblock = 0
        for _ in range(n_layers_per_block[n_pool + i + 1]):
            plan.append(('brc', n, growth))
            n += growth
            nblock += 1
    plan.append(('softmax', n, n_classes))
    return plan


def _bn_relu(x, beta, gamma, eps=1e-4):
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = x.var(axis=(0, 2, 3), keepdims=True)                # biased (P10)
    inv_std = 1.0 / np.sqrt(var + eps)
    y = (x - mean) * (gamma[None, :, None, None] * inv_std) + beta[None, :, None, None]
    return np.maximum(y, 0)


def densenet_forward(params, x, layer=('pool4',), n_layers_per_block=LAYERS_PER_BLOCK,
                     n_pool=N_POOL, growth=GROWTH, return_probs=True):
    """build_fcdensenet(...): returns hidden_outputs ('input' / 'pool k' stacks, in the order
    they are produced, FCDenseNet.py:73-74,99-100) + [probs (B,C,H,W)]."""
    it = iter(params)
    hidden = []
    if 'input' in layer:
        hidden.append(x)
    ints = [int(h[-1]) for h in layer if h != 'input']
    p = next(it)
    stack = nn.conv2d(x, p['W'], p['b'], pad=1, relu=False)            # :77-78 (linear)

    def brc(t, k):
        q = next(it)
        t = _bn_relu(t, q['beta'], q['gamma'])
        return nn.conv2d(t, q['W'], q['b'], pad=k // 2, relu=False)

    skips = []
    for i in range(n_pool):
        for _ in range(n_layers_per_block[i]):
            l = brc(stack, 3)
            stack = np.concatenate([stack, l], axis=1)                 # :92 stack first
        skips.append(stack)
        stack = nn.maxpool2(brc(stack, 1))                             # TransitionDown
        if i + 1 in ints:
            hidden.append(stack)
    skips = skips[::-1]
    block = []
    for _ in range(n_layers_per_block[n_pool]):
        l = brc(stack, 3)
        block.append(l)
        stack = np.concatenate([stack, l], axis=1)
    for i in range(n_pool):
        q = next(it)                                                   # TransitionUp
        up = nn.deconv2d(np.concatenate(block, axis=1), q['W'], q['b'], stride=2)
        skip = skips[i]
        H, W = min(up.shape[2], skip.shape[2]), min(up.shape[3], skip.shape[3])
        stack = np.concatenate([nn.center_crop(up, H, W), nn.center_crop(skip, H, W)], axis=1)
        block = []
        for _ in range(n_layers_per_block[n_pool + i + 1]):
            l = brc(stack, 3)
            block.append(l)
            stack = np.concatenate([stack, l], axis=1)
    q = next(it)
    score = nn.conv2d(stack, q['W'], q['b'], pad=0, relu=False)        # SoftmaxLayer conv 1x1
    out = list(hidden)
    if return_probs:
        out.append(nn.softmax_channels(score))
    return out
