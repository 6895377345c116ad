"""Oracle: gradient of the DAE reconstruction error  E(y) = sum_{c,px} (r(y|h) - y)^2  w.r.t. y, by
hand-written reverse-mode differentiation of oracle/dae.py.  TEST INFRASTRUCTURE.

The reference never differentiates the DAE at inference (its "gradient" is the residual
r - y); BASELINE.json's north_star describes this true-gradient variant, SURVEY section 8(f) rank 4
lists it as an extension with the build's own oracle.  There is therefore no reference parity for
it: this restatement is pinned by finite differences (tests/test_oracle_kat.py) and the HIP path is
checked against it.

Conventions (build-defined where Theano's choice is not observable here):
  * max-pool backward routes the upstream gradient to EVERY position equal to the window maximum
    (the same equality mask DePool2D uses);
  * relu'(0) = 0;
  * the DePool2D equality mask is piecewise constant: no gradient flows through it.
"""
import numpy as np

from . import nn
from .dae import _n_pool, dae_forward


def _conv_bwd_data(g, W, pad, in_hw):
    """Adjoint of nn.conv2d(x, W, pad=pad) w.r.t. x: 'full' correlation of g with the flipped,
    channel-transposed filter, cropped at offset pad.  W is (Cout, Cin, 3, 3)."""
    Wt = np.ascontiguousarray(np.transpose(W, (1, 0, 2, 3))[:, :, ::-1, ::-1])
    full = nn.conv2d(g, Wt, None, pad=W.shape[2] - 1)
    H, Wd = in_hw
    return full[:, :, pad:pad + H, pad:pad + Wd]


def _center(big, small):
    return (big - small) // 2


def dae_sqerr_grad(params, h_list, y, concat_h=('pool4',), padding=100, n_filters=64,
                   conv_before_pool=1, additional_pool=2, skip=True, unpool_type='trackind'):
    """Returns (grad, r): grad = dE/dy for E = sum (r - y)^2 (per image sums are independent),
    r = the DAE output.  Standard DAE, unpool_type in {trackind, inverse}, bn = 0."""
    assert unpool_type in ('trackind', 'inverse')
    concat_h = list(concat_h)
    n_pool, total = _n_pool(concat_h, additional_pool)
    cfg = dict(concat_h=concat_h, padding=padding, n_filters=n_filters,
               conv_before_pool=conv_before_pool, additional_pool=additional_pool, skip=skip,
               unpool_type=unpool_type)
    r, net = dae_forward(params, h_list, y, return_net=True, **cfg)
    d = r - y
    # E = sum d^2;  dE/dr = 2 d;  direct term dE/dy = -2 d
    g_r = 2.0 * d
    g_score = r * (g_r - (r * g_r).sum(axis=1, keepdims=True))        # softmax backward
    # ---- decoder backward: level 1 .. total ------------------------------------------------
    g_pool = {}                                  # gradient w.r.t. pool_p (pre-concat), full size
    g_f = g_score                                # gradient w.r.t. fused_up1 (already cropped)
    for p in range(1, total + 1):
        u_out = net['up_out%d' % p]              # up_conv_p output (pre-pool-p size)
        other = net['pool%d' % (p - 1)] if p > 1 else y
        H = min(u_out.shape[2], other.shape[2])
        Wd = min(u_out.shape[3], other.shape[3])
        # fused_up_p = crop(u_out) [+ crop(other)]
        g_c = np.zeros_like(u_out)
        cy, cx = _center(u_out.shape[2], H), _center(u_out.shape[3], Wd)
        g_c[:, :, cy:cy + H, cx:cx + Wd] = g_f
        if skip and p > 1:
            gp = g_pool.setdefault(p - 1, np.zeros_like(other))
            oy, ox = _center(other.shape[2], H), _center(other.shape[3], Wd)
            gp[:, :, oy:oy + H, ox:ox + Wd] += g_f
        W = params['up_conv%d' % p][0]
        g_u = _conv_bwd_data(g_c, W, 1, u_out.shape[2:])              # d/d(unpooled input)
        # DePool2D backward: sum of the masked gradient over each 2x2 window
        pre, pooled = net['pre%d' % p], net['pool%d' % p]
        h2, w2 = pooled.shape[2], pooled.shape[3]
        mask = pre[:, :, :2 * h2, :2 * w2] == np.repeat(np.repeat(pooled, 2, 2), 2, 3)
        gm = np.where(mask, g_u[:, :, :2 * h2, :2 * w2], 0.0)
        g_f = gm.reshape(gm.shape[0], gm.shape[1], h2, 2, w2, 2).sum(axis=(3, 5))
        # g_f is now the gradient w.r.t. the decoder input of level p: fused_up_{p+1}, or, for
        # p == total, the encoder's last pool
    g_pool[total] = g_pool.get(total, 0) + g_f
    # ---- encoder backward: level total .. 1 --------------------------------------------------
    # channel bookkeeping of the concats (h first): the gradient flows only to the non-h channels
    hch = {name: h.shape[1] for name, h in zip(concat_h, h_list)}
    g_in = None
    for p in range(total, 0, -1):
        pre, pooled = net['pre%d' % p], net['pool%d' % p]
        gp = g_pool.get(p)
        if gp is None:
            gp = np.zeros_like(pooled)
        h2, w2 = pooled.shape[2], pooled.shape[3]
        g_a = np.zeros_like(pre)
        rep = np.repeat(np.repeat(gp, 2, 2), 2, 3)
        mask = pre[:, :, :2 * h2, :2 * w2] == np.repeat(np.repeat(pooled, 2, 2), 2, 3)
        g_a[:, :, :2 * h2, :2 * w2] = np.where(mask, rep, 0.0)        # max-pool backward (all ties)
        for i in range(conv_before_pool, 0, -1):
            name = 'conv%d_%d' % (p, i)
            act = net['_act_' + name]
            g_z = np.where(act > 0, g_a, 0.0)                                 # relu'(0) = 0
            first_pad = (p == 1 and i == 1 and len(concat_h) == 1 and concat_h[-1] != 'input'
                         and padding > 0)
            pad = padding if first_pad else 1
            W = params[name][0]
            in_hw = (g_z.shape[2] + 2 - 2 * pad, g_z.shape[3] + 2 - 2 * pad)
            g_x = _conv_bwd_data(g_z, W, pad, in_hw)
            # strip the h channels if this conv follows a concat point
            at = ('input' if p == 1 else 'pool%d' % (p - 1)) if i == 1 else None
            if at in hch:
                g_x = g_x[:, hch[at]:]
            g_a = g_x
        if p > 1:
            g_pool[p - 1] = g_pool.get(p - 1, 0) + g_a
        else:
            g_in = g_a
    return g_in - 2.0 * d, r


def sqerr(params, h_list, y, **cfg):
    r = dae_forward(params, h_list, y, **cfg)
    return float(((r - y) ** 2).sum())
