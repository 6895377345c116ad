"""CPU baseline: float32 torch-CPU restatement of the configs[1] loop (FCN-8 forward, standard
DAE forward, the refinement loop).  TEST INFRASTRUCTURE -- imported only by bench.py's
`cpu_baseline` leg (the timed "reference CPU path" stand-in) and by the
test that pins it to the numpy oracle.

Why torch and not the numpy oracle: SURVEY 8(d) / BASELINE.md section 4 ask for the same
algorithm class as Theano's CPU convolution (CorrMM = im2col + one BLAS GEMM per image) on ALL
host cores; torch's CPU conv2d (oneDNN / im2col+GEMM) is that, multi-threaded.  Same layer
semantics as oracle/fcn8.py, oracle/dae.py, oracle/refine.py (each function cites the reference
lines it follows); results agree with the float64 numpy oracle to float32 rounding
(tests/test_oracle_torch_cpu.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))


def prepare_params(params):
    """name -> (W, b) numpy -> float32 torch tensors (Deconv2DLayer weights pre-flipped: P3)."""
    out = {}
    for k, v in params.items():
        W, b = v[0], v[1]
        out[k] = (_t(W), None if b is None else _t(b))
    return out


def conv(x, p, pad, relu=True):
    """Conv2DLayer, flip_filters=False, stride 1 (P1); default nonlinearity ReLU."""
    y = F.conv2d(x, p[0], p[1], padding=pad)
    return F.relu_(y) if relu else y


def deconv(x, p, stride):
    """Deconv2DLayer crop='valid', linear: conv_transpose2d with the kernel flipped (P3;
    models/fcn8.py:90,100,109)."""
    return F.conv_transpose2d(x, torch.flip(p[0], (2, 3)), p[1], stride=stride)


def center_crop(x, H, W):
    oh, ow = (x.shape[2] - H) // 2, (x.shape[3] - W) // 2
    return x[:, :, oh:oh + H, ow:ow + W]


def crop_sum(a, b):
    H, W = min(a.shape[2], b.shape[2]), min(a.shape[3], b.shape[3])
    return center_crop(a, H, W) + center_crop(b, H, W)


def fcn8_forward(P, x, pad=100):
    """models/fcn8.py:30-130 -> (pool4, probs).  P = prepare_params(fcn8 params)."""
    c = lambda name, t, p: conv(t, P[name], p)
    pool = lambda t: F.max_pool2d(t, 2)           # ignore_border=True == floor (P4)
    t = c('conv1_1', x, pad); t = c('conv1_2', t, 1); t = pool(t)
    t = c('conv2_1', t, 1); t = c('conv2_2', t, 1); t = pool(t)
    t = c('conv3_1', t, 1); t = c('conv3_2', t, 1); t = c('conv3_3', t, 1); pool3 = t = pool(t)
    t = c('conv4_1', t, 1); t = c('conv4_2', t, 1); t = c('conv4_3', t, 1); pool4 = t = pool(t)
    t = c('conv5_1', t, 1); t = c('conv5_2', t, 1); t = c('conv5_3', t, 1); t = pool(t)
    t = c('fc6', t, 0); t = c('fc7', t, 0); t = c('score_fr', t, 0)       # ReLU on scores (P2)
    fused = crop_sum(deconv(t, P['score2'], 2), c('score_pool4', pool4, 0))
    final = crop_sum(deconv(fused, P['score4'], 2), c('score_pool3', pool3, 0))
    up = deconv(final, P['upsample'], 8)
    score = center_crop(up, min(up.shape[2], x.shape[2]), min(up.shape[3], x.shape[3]))
    return pool4, torch.softmax(score, dim=1)


def depool_eqmask(up, pre, pooled):
    """DePool2D (layers/mylayers.py:88-115): repeat x2, paste top-left, equality mask."""
    h, w = pooled.shape[2], pooled.shape[3]
    rep = lambda t: t.repeat_interleave(2, 2).repeat_interleave(2, 3)
    out = torch.zeros_like(pre)
    out[:, :, :2 * h, :2 * w] = torch.where(pre[:, :, :2 * h, :2 * w] == rep(pooled), rep(up),
                                            torch.zeros((), dtype=pre.dtype))
    return out


def dae_forward(P, h, y, padding=100, n_pool=4, total=6):
    """Standard DAE, concat_h=['pool4'], conv_before_pool=1, skip, trackind
    (models/fcn_down.py:77-136, models/fcn_up.py:143-151) -> r."""
    pre, pool = {}, {0: y}
    t = y
    for p in range(total):
        t = conv(t, P['conv%d_1' % (p + 1)], padding if p == 0 else 1)
        pre[p + 1] = t
        pool[p + 1] = t = F.max_pool2d(t, 2)
        if p + 1 == n_pool:
            t = torch.cat([h, t], dim=1)                              # h first (P13)
    t = pool[total]
    for p in range(total, 0, -1):
        u = conv(depool_eqmask(t, pre[p], pool[p]), P['up_conv%d' % p], 1, relu=False)
        other = pool[p - 1]
        H, W = min(u.shape[2], other.shape[2]), min(u.shape[3], other.shape[3])
        t = center_crop(u, H, W) + center_crop(other, H, W) if p > 1 else center_crop(u, H, W)
    return torch.softmax(t, dim=1)


def refine(P_dae, h, y, step, num_iter, per_image=True, eps=-1.0):
    """iterative_inference.py:258-284.  per_image=True is the reference's schedule (B = 1 DAE
    calls, image after image); False runs the whole batch per call."""
    if not per_image:
        for _ in range(num_iter):
            r = dae_forward(P_dae, h, y)
            y = torch.clamp(y - step * (y - r), 0.0, 1.0)
        return y
    outs = []
    for im in range(y.shape[0]):
        yi, hi = y[im:im + 1], h[im:im + 1]
        for _ in range(num_iter):
            grad = yi - dae_forward(P_dae, hi, yi)
            yi = torch.clamp(yi - step * grad, 0.0, 1.0)
            if float(torch.linalg.vector_norm(grad, dim=1).mean()) < eps:
                break
        outs.append(yi)
    return torch.cat(outs, dim=0)


def run_batch(P_fcn, P_dae, x, step, num_iter, per_image):
    with torch.no_grad():
        h, y = fcn8_forward(P_fcn, _t(x) if not isinstance(x, torch.Tensor) else x)
        return refine(P_dae, h, y, step, num_iter, per_image=per_image)
