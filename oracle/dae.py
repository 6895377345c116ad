"""Oracle: the standard conditional DAE  r(y | h)  (reference models/DAE_h.py:12-63,
models/fcn_down.py:9-138, models/fcn_up.py:11-172).  TEST INFRASTRUCTURE.

`params`: dict name -> (W, b) (+ name+'_bn' -> (beta, gamma, mean, inv_std) when bn=1) in
Lasagne layouts.  `param_order(...)` is the get_all_param_values order (P14) of
dae_model_best.npz (DAE_h.py:52-57).
"""
from . import nn


def _n_pool(concat_h, additional_pool):
    # DAE_h.py:37-40 / fcn_down.py:43-46: digit of the LAST concat name
    n = int(concat_h[-1][-1]) if 'pool' in concat_h[-1] else 0
    return n, n + additional_pool


def encoder_filters(p, n_filters):
    """fcn_down.py:98-99: n_filters*2^p, frozen after p = 5."""
    return n_filters * (2 ** min(p, 5))


def param_order(concat_h=('pool4',), conv_before_pool=1, additional_pool=2,
                unpool_type='trackind', bn=0):
    n_pool, total = _n_pool(list(concat_h), additional_pool)
    names = []
    for p in range(total):
        for i in range(1, conv_before_pool + 1):
            names.append('conv%d_%d' % (p + 1, i))
            if bn:
                names.append('conv%d_%d_bn' % (p + 1, i))
    for p in range(total, 0, -1):
        names.append(('up%d' if unpool_type == 'standard' else 'up_conv%d') % p)
        if bn and unpool_type != 'standard':
            names.append('up_conv%d_bn' % p)
    return names


def _bn_avg(x, bnp):
    # BatchNormLayer with deterministic=True and stored averages (pred_dae,
    # iterative_inference.py:189): (x - mean) * (gamma * inv_std) + beta   (P10)
    beta, gamma, mean, inv_std = bnp
    return (x - mean[None, :, None, None]) * (gamma * inv_std)[None, :, None, None] \
        + beta[None, :, None, None]


def dae_forward(params, h_list, y, concat_h=('pool4',), padding=100, n_filters=64,
                conv_before_pool=1, additional_pool=2, skip=True, unpool_type='trackind',
                bn=0, out_softmax=True, return_net=False, pad_multi_concat=False, noise=0.0,
                dropout=0.0, hidden_rand=None):
    """pred_dae_fn(h..., y) -> r  (iterative_inference.py:189-190) for dae kind 'standard'.

    Deterministic main path: GaussianNoiseLayer and DropoutLayer are identities (P8, P9).  The
    DePool2D masks come from the same deterministic encoder pass unless `hidden_rand` is given
    and noise > 0 or dropout > 0: then they come from the reference's stochastic hidden
    re-forward (layers/mylayers.py:91-93) with the caller's samples.  Without
    `hidden_rand` the oracle states the DETERMINISTIC-mask semantics, which deviates from the
    reference whenever noise or dropout is non-zero (the reference is then itself random).
    """
    concat_h = list(concat_h)
    h_list = list(h_list)
    assert len(h_list) == len(concat_h)
    assert all(el in ['pool1', 'pool2', 'pool3', 'pool4', 'input', 'pool5'] for el in concat_h)
    n_pool, total = _n_pool(concat_h, additional_pool)
    if concat_h[-1] == 'input' and additional_pool == 0:
        raise ValueError('It seems your DAE will have no conv/pooling layers!')  # fcn_down.py:71
    net = {'input': y}
    pos = 0

    def maybe_concat(name, t, pos):
        # model_helpers.py:72-107: h first (P13)
        if pos < len(concat_h) and concat_h[pos] == name:
            return nn.concat_h_first(h_list[pos], t), pos + 1
        return t, pos

    t, pos = maybe_concat('input', y, pos)
    pre = {}     # input of pool p (post conv/dropout/bn), the `pool2d_layer_in` of DePool2D
    for p in range(total):                                   # fcn_down.py:77
        for i in range(1, conv_before_pool + 1):
            # pad_multi_concat: BUILD-DEFINED generalisation (SURVEY A9', config 5): the pad-100
            # rule also with several concat points, so that h maps of a pad-100 FCN-8 fit
            if p == 0 and i == 1 and (len(concat_h) == 1 or pad_multi_concat) \
                    and concat_h[-1] != 'input' and padding > 0:
                pad = padding                                # fcn_down.py:90-92
            else:
                pad = 1                                      # 'same' for 3x3
            name = 'conv%d_%d' % (p + 1, i)
            t = nn.conv2d(t, params[name][0], params[name][1], pad=pad, relu=True)  # :102-104
            net['_act_' + name] = t
            if bn:
                t = _bn_avg(t, params[name + '_bn'])         # :112-114
        pre[p + 1] = t
        net['pre%d' % (p + 1)] = t
        net['pool%d' % (p + 1)] = t = nn.maxpool2(t)         # :122
        if p < n_pool:
            t, pos = maybe_concat('pool%d' % (p + 1), t, pos)  # :131-134
    # noise > 0 or dropout > 0: every DePool2D re-evaluates the down path up to its pool WITHOUT
    # deterministic=True (layers/mylayers.py:91-93): GaussianNoiseLayer on y (fcn_down.py:60-63)
    # and the DropoutLayers after the convs (:108-111, rescale 1/(1-p)) are active, one fresh
    # sample per DePool2D.  hidden_rand(kind, level, name, shape) supplies the samples.
    hidden = {}
    # The gate is "any stochastic layer live": dropout > 0 alone (noise == 0, the configuration of
    # the golden experiment name of plots.ipynb:84) already makes the reference's masks random.
    if (noise > 0 or dropout > 0) and hidden_rand is not None and unpool_type == 'trackind':
        assert not bn
        for p in range(total, 0, -1):
            u = y + noise * hidden_rand('noise', p, None, y.shape) if noise > 0 else y
            hpos = 0
            u, hpos = maybe_concat('input', u, hpos)
            for q in range(p):
                for i in range(1, conv_before_pool + 1):
                    pad = padding if (q == 0 and i == 1 and (len(concat_h) == 1 or pad_multi_concat)
                                      and concat_h[-1] != 'input' and padding > 0) else 1
                    name = 'conv%d_%d' % (q + 1, i)
                    u = nn.conv2d(u, params[name][0], params[name][1], pad=pad, relu=True)
                    if dropout > 0:
                        u = u * hidden_rand('dropout', p, name, u.shape) / (1.0 - dropout)
                pre_q = u
                pool_q = u = nn.maxpool2(u)
                if q < n_pool:
                    u, hpos = maybe_concat('pool%d' % (q + 1), u, hpos)
            hidden[p] = (pre_q, pool_q)
    # decoder, fcn_up.py:143-151 / UnpoolNet :11-115
    for p in range(total, 0, -1):
        if unpool_type == 'standard':
            name = 'up%d' % p
            u = nn.deconv2d(t, params[name][0], params[name][1], stride=2)   # :41-45
        elif unpool_type in ('trackind', 'inverse'):
            # trackind: DePool2D (:70-75); inverse: InverseLayer of the pool (:76-79) -- the
            # gradient of max-pooling w.r.t. its input with `t` as upstream, the same
            # equality-mask arithmetic when masks are deterministic.
            mpre, mpool = hidden.get(p, (pre[p], net['pool%d' % p]))
            u = nn.depool_eqmask(t, mpre, mpool)
            name = 'up_conv%d' % p
            u = nn.conv2d(u, params[name][0], params[name][1], pad=1, relu=False)  # :83-86
            if bn:
                u = _bn_avg(u, params[name + '_bn'])         # :91-93
        else:
            raise ValueError('Unkown unpool type')
        net['up_out%d' % p] = u
        if skip and p > 1:
            t = nn.crop_sum(u, net['pool%d' % (p - 1)])      # :96-102 (pre-concat pool)
        else:
            t = nn.crop_like(net['pool%d' % (p - 1)] if p > 1 else y, u)  # :104-113
        net['fused_up%d' % p] = t
    net['score'] = t
    r = nn.softmax_channels(t) if out_softmax else t         # :154-169
    net['probs_dimshuffle'] = r
    return (r, net) if return_net else r
