"""Seeded synthetic weights, images and labels (no datasets / checkpoints are reachable).

Shapes and value ranges follow the reference's data contract: images float32 RGB in [0,1]
(`return_0_255=False`, iterative_inference.py:117-118), labels one-hot with the void channel
LAST (`void = n_classes`, iterative_inference.py:122-125,234).  Weights use the reference's
parameter layouts (Lasagne Conv2DLayer W[out,in,kh,kw]; Deconv2DLayer W[in,out,kh,kw]) so a
real `arr_%d` checkpoint can replace them (see weights.py).
"""
import numpy as np

FCN8_CONVS = [  # name, cin, cout, k   (models/fcn8.py:34-85)
    ('conv1_1', None, 64, 3), ('conv1_2', 64, 64, 3),
    ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3),
    ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3),
    ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv4_3', 512, 512, 3),
    ('conv5_1', 512, 512, 3), ('conv5_2', 512, 512, 3), ('conv5_3', 512, 512, 3),
]


def _he_uniform(rng, shape, fan_in):
    bound = np.sqrt(6.0 / fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _bias(rng, n):
    # small positive biases so ReLU zeros / pooling ties are exercised but units stay alive
    return rng.uniform(0.0, 0.1, size=(n,)).astype(np.float32)


def bilinear_kernel(k):
    """The usual FCN bilinear upsampling filter of size k (symmetric)."""
    f = (k + 1) // 2
    c = f - 1 if k % 2 == 1 else f - 0.5
    og = np.ogrid[:k, :k]
    return ((1 - abs(og[0] - c) / f) * (1 - abs(og[1] - c) / f)).astype(np.float32)


def _deconv(rng, cin, cout, k, jitter):
    """Per-class bilinear filter plus a seeded asymmetric perturbation (so that the P3
    flip convention is observable in parity tests)."""
    W = np.zeros((cin, cout, k, k), dtype=np.float32)
    for c in range(min(cin, cout)):
        W[c, c] = bilinear_kernel(k)
    if jitter > 0:
        W += rng.uniform(-jitter, jitter, size=W.shape).astype(np.float32)
    return W


def make_fcn8_params(nb_in_channels=3, n_classes=11, seed=1234, width_div=1, fc_channels=4096,
                     deconv_jitter=0.02):
    """FCN-8s parameters (models/fcn8.py:30-110).  `width_div` / `fc_channels` shrink the net
    for fast tests (width_div=1, fc_channels=4096 is the real architecture)."""
    rng = np.random.default_rng(seed)
    p = {}
    cprev = nb_in_channels
    for name, _, cout, k in FCN8_CONVS:
        cout = max(cout // width_div, 4)
        p[name] = (_he_uniform(rng, (cout, cprev, k, k), cprev * k * k), _bias(rng, cout))
        cprev = cout
    c3 = p['conv3_3'][0].shape[0]
    c4 = p['conv4_3'][0].shape[0]
    p['fc6'] = (_he_uniform(rng, (fc_channels, cprev, 7, 7), cprev * 49), _bias(rng, fc_channels))
    p['fc7'] = (_he_uniform(rng, (fc_channels, fc_channels, 1, 1), fc_channels),
                _bias(rng, fc_channels))
    p['score_fr'] = (_he_uniform(rng, (n_classes, fc_channels, 1, 1), fc_channels),
                     _bias(rng, n_classes))
    p['score2'] = (_deconv(rng, n_classes, n_classes, 4, deconv_jitter), _bias(rng, n_classes))
    p['score_pool4'] = (_he_uniform(rng, (n_classes, c4, 1, 1), c4), _bias(rng, n_classes))
    p['score4'] = (_deconv(rng, n_classes, n_classes, 4, deconv_jitter), _bias(rng, n_classes))
    p['score_pool3'] = (_he_uniform(rng, (n_classes, c3, 1, 1), c3), _bias(rng, n_classes))
    p['upsample'] = (_deconv(rng, n_classes, n_classes, 16, deconv_jitter),
                     _bias(rng, n_classes))
    return p


def _bn_params(rng, c):
    """(beta, gamma, mean, inv_std) of a BatchNormLayer; a few negative gammas so that the affine
    is not monotone in every channel."""
    gamma = rng.uniform(0.6, 1.4, c).astype(np.float32)
    gamma[::5] *= -1.0
    return (rng.uniform(-0.1, 0.1, c).astype(np.float32), gamma,
            rng.uniform(0.0, 0.3, c).astype(np.float32), rng.uniform(0.7, 1.5, c).astype(np.float32))


def make_dae_params(n_classes=11, h_channels=(512,), concat_h=('pool4',), n_filters=64,
                    conv_before_pool=1, additional_pool=2, unpool_type='trackind', seed=4321,
                    out_gain=4.0, bn=0, dec_gain=1.0):
    """Standard-DAE parameters (models/fcn_down.py:77-136, models/fcn_up.py:26-86).

    Channel bookkeeping mirrors the builders: encoder conv p has n_filters*2^min(p,5) filters
    (fcn_down.py:98-99); h is concatenated (h first) after `pool k` for each name in concat_h
    (:131-134) or at the input; decoder up_conv p outputs the channels of conv p-1
    (fcn_up.py:33-34), n_classes for p == 1 (:30-31).  `out_gain` scales up_conv1 so the
    random-weight reconstruction is not a flat softmax; `dec_gain` scales the other (linear)
    up_convs -- He-uniform bounds are sized for rectified layers, so at 1.0 every decoder level
    doubles the variance it passes on (see DAMPED below).
    """
    rng = np.random.default_rng(seed)
    concat_h = list(concat_h)
    n_pool = int(concat_h[-1][-1]) if 'pool' in concat_h[-1] else 0
    total = n_pool + additional_pool
    hch = dict(zip(concat_h, h_channels))
    p = {}
    cprev = n_classes + hch.get('input', 0)
    enc_out = []
    for lvl in range(total):
        f = n_filters * (2 ** min(lvl, 5))
        for i in range(1, conv_before_pool + 1):
            p['conv%d_%d' % (lvl + 1, i)] = (_he_uniform(rng, (f, cprev, 3, 3), cprev * 9),
                                             _bias(rng, f))
            if bn:
                p['conv%d_%d_bn' % (lvl + 1, i)] = _bn_params(rng, f)
            cprev = f
        enc_out.append(f)
        if lvl < n_pool:
            cprev += hch.get('pool%d' % (lvl + 1), 0)
    # decoder input of level `total` is the (possibly concatenated) last pool
    cin = cprev
    for lvl in range(total, 0, -1):
        cout = n_classes if lvl == 1 else enc_out[lvl - 2]
        if unpool_type == 'standard':
            W = _he_uniform(rng, (cin, cout, 4, 4), cin * 4)
            p['up%d' % lvl] = (W, _bias(rng, cout))
        else:
            gain = out_gain if lvl == 1 else dec_gain
            p['up_conv%d' % lvl] = (gain * _he_uniform(rng, (cout, cin, 3, 3), cin * 9),
                                    _bias(rng, cout))
            if bn:
                p['up_conv%d_bn' % lvl] = _bn_params(rng, cout)
        cin = cout
    return p


# The DAMPED parity workload ("Next round" item 1).  With the default synthetic weights
# the refinement loop is chaotic (profiles/r02_sensitivity.md): y0 is a smooth bilinear upsampling,
# so the four values of almost every 2x2 pooling window nearly tie, a 1e-7 perturbation flips
# DePool2D equality-mask bits (layers/mylayers.py:111-114) and the out_gain-4 / variance-doubling
# decoder amplifies every flip 40x per step.  This second seeded set keeps the architecture, the
# random filters and every layer's contribution, and changes four gains:
#   temperature   the reference's own knob (upsample.W, b / T, models/fcn8.py:194-198): y0 confident
#                 (mean max-probability 0.95 instead of 0.55), so argmax-based checks mean something;
#   deconv_jitter pixel-scale texture on the 8x upsampling filter: the values inside a pooling
#                 window differ at the scale of the activations, near-ties become rare;
#   dec_gain, out_gain   the decoder passes on less than it receives: in float64 a 1e-6 / 1e-5
#                 perturbation of y0 DECAYS over the 10 steps of the bench workload (mean error
#                 x0.9 per step = the (1 - step) factor; profiles/r03_sensitivity.md).
# On it the fast paths carry fixed-tolerance end-to-end checks (tests/test_gpu_damped.py).  The
# chaotic default set stays the stress test and the bench workload.
DAMPED = dict(temperature=0.25, deconv_jitter=0.3, dec_gain=0.35, out_gain=0.25)


def make_damped_set(h_channels=(512,), concat_h=('pool4',), **dae_kw):
    """(fcn8 params, dae params, temperature) of the damped parity workload: pass `temperature`
    to FCN8 / fcn8_forward."""
    fp = make_fcn8_params(deconv_jitter=DAMPED['deconv_jitter'])
    dp = make_dae_params(h_channels=h_channels, concat_h=concat_h, out_gain=DAMPED['out_gain'],
                         dec_gain=DAMPED['dec_gain'], **dae_kw)
    return fp, dp, DAMPED['temperature']


# The damped set of BASELINE configs[2] (FC-DenseNet103 + standard DAE, padding 0, h = pool4): the DAE's two
# gains as above (the loop contracts), and the score layer of the DenseNet scaled by 1 / temperature (the same
# confidence knob, applied to the 1x1 SoftmaxLayer conv of models/FCDenseNet.py:134: y0 confident, argmax-based
# checks mean something).  tests/test_gpu_bf16.py, scripts/parity_c3.py.
def make_damped_densenet_set(plan, seed=2024, dae_seed=4321):
    """(densenet params, dae params) of the damped parity workload of configs[2]."""
    params = make_densenet_params(plan, seed=seed)
    last = dict(params[-1])
    assert last['kind'] == 'softmax'
    last['W'] = (last['W'] / DAMPED['temperature']).astype(np.float32)
    last['b'] = (last['b'] / DAMPED['temperature']).astype(np.float32)
    params = params[:-1] + [last]
    dp = make_dae_params(h_channels=(464,), out_gain=DAMPED['out_gain'], dec_gain=DAMPED['dec_gain'],
                         seed=dae_seed)
    return params, dp


def labels_from_map(y, void_frac=0.05, seed=99, block=16):
    """One-hot labels (N, C+1, H, W), void last, whose class map is the argmax of the probability
    map `y` (N, C, H, W) -- e.g. the float64-refined map of the damped set, so that the mIoU of a
    consistent path is ~1 -- with seeded void blocks as in `make_labels`."""
    y = np.asarray(y)
    n, c, h, w = y.shape
    rng = np.random.default_rng(seed)
    gh, gw = (h + block - 1) // block, (w + block - 1) // block
    void = rng.random((n, gh, gw), dtype=np.float32) < void_frac
    void = np.repeat(np.repeat(void, block, axis=1), block, axis=2)[:, :h, :w]
    cls = np.where(void, c, y.argmax(axis=1))
    onehot = np.zeros((n, c + 1, h, w), dtype=np.float32)
    np.put_along_axis(onehot, cls[:, None, :, :], 1.0, axis=1)
    return onehot


def make_contextmod_params(n_classes=11, h_channels=3, seed=777, jitter=0.05):
    """Context-module parameters (models/contextmod_dae.py:61-105): identity-initialised
    (IdentityInit, :61-72) plus a seeded perturbation so every tap matters.  conv1 is a
    Conv2DLayer W[out,in,3,3]; dilconv* are DilatedConv2DLayer W[in,out,k,k] (P11)."""
    rng = np.random.default_rng(seed)
    p = {}
    cin = n_classes + h_channels
    p['conv1'] = (_he_uniform(rng, (n_classes, cin, 3, 3), cin * 9), _bias(rng, n_classes))
    for i in range(1, 8):
        k = 1 if i == 7 else 3
        W = np.zeros((n_classes, n_classes, k, k), dtype=np.float32)
        for c in range(n_classes):
            W[c, c, k // 2, k // 2] = 1.0
        W += rng.uniform(-jitter, jitter, size=W.shape).astype(np.float32)
        p['dilconv%d' % i] = (W, _bias(rng, n_classes))
    return p


def make_fcn8_dae_params(n_classes=11, concat_h=('input',), h_channels=(3,), seed=555,
                         width_div=1, fc_channels=4096):
    """FCN-8-shaped DAE parameters (models/fcn8_dae.py:56-160): an FCN-8 on y (n_classes
    channels) whose conv after each concat point takes h channels first."""
    hch = dict(zip(concat_h, h_channels))
    p = make_fcn8_params(n_classes + hch.get('input', 0), n_classes, seed=seed,
                         width_div=width_div, fc_channels=fc_channels)
    rng = np.random.default_rng(seed + 1)
    nxt = {'pool1': 'conv2_1', 'pool2': 'conv3_1', 'pool3': 'conv4_1', 'pool4': 'conv5_1'}
    for pool, conv in nxt.items():
        if pool in hch:
            W, b = p[conv]
            cin = W.shape[1] + hch[pool]
            p[conv] = (_he_uniform(rng, (W.shape[0], cin, 3, 3), cin * 9), b)
    return p


def make_densenet_params(plan, seed=2024):
    """FC-DenseNet parameters for a `layer_plan` (densenet.py): list of dicts in creation order.
    Conv2DLayer W[out,in,k,k]; TransitionUp Deconv2DLayer W[in,out,3,3]; BN gamma/beta seeded
    around (1, 0)."""
    rng = np.random.default_rng(seed)
    out = []
    for kind, cin, cout in plan:
        p = {'kind': kind}
        if kind in ('brc', 'td'):
            p['gamma'] = rng.uniform(0.5, 1.5, size=(cin,)).astype(np.float32)
            p['beta'] = rng.uniform(-0.2, 0.2, size=(cin,)).astype(np.float32)
        if kind == 'tu':
            p['W'] = _he_uniform(rng, (cin, cout, 3, 3), cin * 9 / 4.0)
        else:
            k = 1 if kind in ('td', 'softmax') else 3
            p['W'] = _he_uniform(rng, (cout, cin, k, k), cin * k * k)
        p['b'] = _bias(rng, cout)
        out.append(p)
    return out


def make_images(n, h=224, w=224, channels=3, seed=1234):
    """Uniform [0,1) float32 RGB batch (N,C,H,W)."""
    rng = np.random.default_rng(seed)
    return rng.random((n, channels, h, w), dtype=np.float32)


def make_labels(n, h=224, w=224, n_classes=11, void_frac=0.05, seed=99, block=16):
    """Smooth blobby class maps in {0..n_classes} (n_classes == void), as one-hot float32
    (N, n_classes+1, H, W) with the void channel last."""
    rng = np.random.default_rng(seed)
    gh, gw = (h + block - 1) // block, (w + block - 1) // block
    low = rng.random((n, n_classes, gh, gw), dtype=np.float32)
    cls = low.argmax(axis=1)
    cls = np.repeat(np.repeat(cls, block, axis=1), block, axis=2)[:, :h, :w]
    void = rng.random((n, gh, gw), dtype=np.float32) < void_frac
    void = np.repeat(np.repeat(void, block, axis=1), block, axis=2)[:, :h, :w]
    cls = np.where(void, n_classes, cls)
    onehot = np.zeros((n, n_classes + 1, h, w), dtype=np.float32)
    np.put_along_axis(onehot, cls[:, None, :, :], 1.0, axis=1)
    return onehot
