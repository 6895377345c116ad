"""Standard conditional DAE  r(y | h)  on the HIP kernels (mirror of reference
models/DAE_h.py:12-63, models/fcn_down.py:9-138, models/fcn_up.py:11-172).

The layer plan is static per (B, H, W): every launch is a fused kernel --
  encoder level p :  conv3x3 (+bias +ReLU; two-source gather where h is concatenated) , maxpool
  decoder level p :  conv3x3 whose input gather IS the DePool2D equality-mask unpool, with the
                     skip sum (ElemwiseSumLayer) and the final center crop fused in the epilogue
so the unpooled tensors, the concatenated tensor and the cropped score are never materialised.
"""
import os

import numpy as np

import torch

from . import ops
from .weights import load_param_list

# float64 decoder layers on the halo-tile kernel apply DePool2D while staging their patches
# (conv_halo_f64.hip); 0: materialise the unpooled window first (pool_unpool.hip) and run the plain conv
F64_FUSE_UNPOOL = os.environ.get('IISEG_F64_FUSE_UNPOOL', '1') != '0'


def _n_pool(concat_h, additional_pool):
    n = int(concat_h[-1][-1]) if 'pool' in concat_h[-1] else 0   # DAE_h.py:37-40
    return n, n + additional_pool


def param_order(concat_h=('pool4',), conv_before_pool=1, additional_pool=2,
                unpool_type='trackind', bn=0):
    """get_all_param_values order of dae_model_best.npz; with bn=1 every conv is
    followed by its BatchNormLayer entry `<name>_bn` (fcn_down.py:112-114, fcn_up.py:91-93)."""
    _, total = _n_pool(list(concat_h), additional_pool)
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


# mma='bf16c8': decoder levels with at least this many input channels materialise DePool2D (ops.unpool_c8) and run
# their conv as a plain layer (0: every level unpools in its conv's patch staging)
C8_UNPOOL_MIN_CIN = int(os.environ.get('IISEG_C8_UNPOOL_MIN_CIN', '1024'))


def _center(big, small):
    return (big - small) // 2


class StandardDAE:
    """Callable with (h_1..h_k, y) like the compiled pred_dae_fn (iterative_inference.py:189-190)."""

    def __init__(self, params, n_classes, concat_h=('pool4',), padding=100, n_filters=64,
                 conv_before_pool=1, additional_pool=2, skip=True, unpool_type='trackind', bn=0,
                 device='cuda', dtype=torch.float32, pad_multi_concat=False, noise=0.0,
                 dropout=0.0, emulate_noise=False, seed=0, mma=None):
        """mma: matrix-pipe operand precision of the float32 path's convolutions ('f32' default,
        'bf16' = 16-bit MFMA operands with fp32 accumulation; ops.Conv)."""
        concat_h = list(concat_h)
        mma = mma or ops.DEFAULT_MMA
        self.mma = mma
        # bf16 C8 activations between the layers (ops.Conv mma='bf16c8', `_scores_c8`)
        # ('bf16x3': the same plan on hi / lo pairs, the fp32-class mode of that kernel)
        self.c8 = mma in ('bf16c8', 'bf16x3') and dtype == torch.float32
        self.x3 = self.c8 and mma == 'bf16x3'
        assert all(el in ['pool1', 'pool2', 'pool3', 'pool4', 'input', 'pool5']
                   for el in concat_h)                                   # fcn_down.py:39-41
        if concat_h[-1] == 'input' and additional_pool == 0:
            raise ValueError('It seems your DAE will have no conv/pooling layers!')  # :71-72
        if unpool_type not in ('standard', 'trackind', 'inverse'):
            raise ValueError('Unkown unpool type')                       # fcn_up.py:114-115
        if unpool_type == 'standard' and dtype != torch.float32:
            raise NotImplementedError("unpool_type='standard' (4x4/2 transposed conv) is float32 "
                                      "only")
        self.bn = bool(bn)
        self.enc_bn = {}
        # stochastic-mask emulation (optional): DePool2D's masks from a hidden down-path
        # re-forward with GaussianNoiseLayer / DropoutLayer active, one fresh sample per level
        # (whenever noise > 0 OR dropout > 0: layers/mylayers.py:91-93 drops `deterministic` for
        # every stochastic layer).  Off by default: the deterministic masks are the build's
        # reference semantics -- a documented deviation from the reference whenever noise or
        # dropout is non-zero.
        self.noise, self.dropout = float(noise), float(dropout)
        self.emulate_noise = bool(emulate_noise)
        self.random_source = None     # callable(kind, level, name, shape) -> tensor, or None: torch RNG
        self._seed, self._gen = int(seed), None
        self.dtype = dtype
        self._bwd, self._saved = None, None
        self.unpool_type = unpool_type
        self.concat_h, self.padding, self.skip = concat_h, padding, skip
        self.conv_before_pool = conv_before_pool
        self.n_classes = n_classes
        self.n_pool, self.total = _n_pool(concat_h, additional_pool)
        self.device = device
        self.enc, self.dec = {}, {}
        for p in range(self.total):
            for i in range(1, conv_before_pool + 1):
                # pad-100 rule of fcn_down.py:90-94
                # pad_multi_concat: build-defined generalisation for several concat points
                # (SURVEY A9', config 5); the reference applies pad-100 only with ONE concat point
                first_pad = (p == 0 and i == 1 and (len(concat_h) == 1 or pad_multi_concat)
                             and concat_h[-1] != 'input' and padding > 0)
                name = 'conv%d_%d' % (p + 1, i)
                self.enc[name] = ops.Conv(params[name][0], params[name][1],
                                          pad=padding if first_pad else 1, relu=True,
                                          device=device, dtype=dtype, mma=mma)        # :102-104
                if bn:   # BatchNormLayer on the rectified conv, stored averages (:112-114)
                    self.enc_bn[name] = tuple(
                        torch.as_tensor(np.asarray(a)).to(dtype).contiguous().to(device)
                        for a in params[name + '_bn'])
        # The conv that follows a concat point computes W_h*h + W_y*features.  h does not change
        # during a refinement loop, so the h half (a per-pixel bias map) is loop-invariant
        # (SURVEY section 7): it is kept as its own linear conv whose output the y half adds in its
        # epilogue.  Same association with or without a session, float32 only (float64 keeps the
        # single-sum form of the oracle).
        self.hsplit = {}
        if dtype == torch.float32 and os.environ.get('IISEG_H_SPLIT', '1') != '0':
            prev = n_classes
            for p in range(self.total):
                name = 'conv%d_1' % (p + 1)
                W, b = params[name]
                ch = W.shape[1] - prev
                at = 'input' if p == 0 else 'pool%d' % p
                if ch > 0 and at in concat_h:
                    Wt = torch.as_tensor(W)
                    conv_h = ops.Conv(Wt[:, :ch].contiguous(), b, pad=self.enc[name].pad,
                                      relu=False, device=device, dtype=dtype, mma=mma)
                    conv_y = ops.Conv(Wt[:, ch:].contiguous(), None, pad=self.enc[name].pad,
                                      relu=True, device=device, dtype=dtype, mma=mma)
                    self.hsplit[name] = (conv_h, conv_y)
                prev = params['conv%d_%d' % (p + 1, conv_before_pool)][0].shape[0]
        for p in range(self.total, 0, -1):
            if unpool_type == 'standard':                                # fcn_up.py:41-45
                name = 'up%d' % p
                self.dec[name] = ops.Conv(params[name][0], params[name][1], pad=0, relu=False,
                                          layout='iohw', transposed=True, device=device,
                                          dtype=dtype, mma=mma)
                continue
            name = 'up_conv%d' % p
            W, b = params[name]
            if bn:
                # fcn_up.py:91-93: BatchNormLayer (stored averages) on the LINEAR up_conv, before
                # the skip sum -- an affine per output channel, folded into W and b in float64
                beta, gamma, mean, inv_std = (np.asarray(a, np.float64) for a in params[name + '_bn'])
                sc = gamma * inv_std
                W = np.asarray(W, np.float64) * sc[:, None, None, None]
                b = (np.asarray(b, np.float64) - mean) * sc + beta
            self.dec[name] = ops.Conv(W, b, pad=1, relu=False, device=device,
                                      dtype=dtype, mma=mma)                           # fcn_up.py:83-86
        self.conv_log = None
        # DePool2D fused into the conv's input load (halo kernel: 3 loads per PATCH element;
        # Winograd: applied by the input transform) or materialised first.  None (auto): fused in
        # float32, materialised in float64; True / False force either form.
        env = os.environ.get('IISEG_FUSE_UNPOOL', 'auto')
        self.fuse_unpool = None if env == 'auto' else env != '0'
        # compute each decoder level only on the window that reaches the final crop
        self.dce = os.environ.get('IISEG_DECODER_DCE', '1') != '0'
        # inside a refinement loop recompute only the y-dependent part of the encoder maps
        self.licm = os.environ.get('IISEG_ENCODER_LICM', '1') != '0'
        # keep the weights-only border of the encoder maps across batches (see new_session)
        self.fold_border = os.environ.get('IISEG_DAE_BORDER_FOLD', '1') != '0'
        self._store = None
        self._scratch_sessions = {}   # C8, no provenance records: buffer-stable sessions per geometry (new_session)
        self.trace = None   # set to a dict to keep intermediates (debug / parity tests)
        # DePool2D masks as bytes: where the encoder conv that pools and the decoder conv that
        # unpools both run on halo kernels, the pre-pool map is never stored -- the encoder writes
        # pool + one mask byte per pooled element, the decoder reads up + that byte (`_mask_levels`)
        self.use_masks = os.environ.get('IISEG_DEPOOL_MASKS', '1') != '0'
        self.keep_pre = False   # True: the pre-pool maps are needed afterwards (backward_y)

    def _mask_levels(self, overridden):
        """Levels (1-based) whose DePool2D mask travels as bytes in this call."""
        if not self.use_masks or overridden or self.keep_pre or self.bn or self.trace is not None or \
                self.unpool_type == 'standard' or self.dtype not in (torch.float32, torch.float64) or \
                self.fuse_unpool is False:
            return frozenset()
        levels = set()
        for L in range(1, self.total + 1):
            enc = self.enc['conv%d_%d' % (L, self.conv_before_pool)]
            fed_by_h = self.conv_before_pool == 1 and \
                ('input' if L == 1 else 'pool%d' % (L - 1)) in self.concat_h
            # (the levels of the fp32-NCHW form of `scores`: a Conv built for C8 input answers for the
            # form that runs on NCHW input)
            if not fed_by_h and enc.pool_fusable(False) and enc.mask_ok(False) and \
                    self.dec['up_conv%d' % L].mask_ok(False):
                levels.add(L)
        return frozenset(levels)

    def new_session(self, h_list=None, y=None, tags=None):
        """State of one refinement loop (h fixed, y evolving): see `scores`.

        `tags`: one provenance record per h (FCN8.last_provenance: ((store id, geometry),
        image-dependent region)) or None.  When EVERY h comes with the record of a border-folding
        FCN-8 (outside the recorded region the map depends on that net's weights and the geometry
        only), the encoder maps of this DAE have a batch-independent border too: the session of
        the previous batch of the same geometry is handed out again and the first step recomputes
        only the region that y or the image-dependent part of h can reach.  No records (the
        default): a fresh session, every map computed in full on the first step."""
        tags = list(tags) if tags is not None else [None] * len(h_list or [])
        if self.licm and self.fold_border and y is not None and tags and \
                all(t is not None for t in tags):
            key = (tuple(t[0] for t in tags), tuple(y.shape), y.dtype)
            st = self._store
            if st is not None and st.get('key') == key and st.get('primed'):
                st['y8_fresh'] = False       # (a C8 copy of the PREVIOUS loop's last y may be there)
                st['h_fresh'] = True
                st['h_dep'] = [t[1] for t in tags]
                return st
            self._store = {'primed': False, 'key': key}
            return self._store
        if self.c8 and y is not None and h_list is not None and \
                all(isinstance(h, torch.Tensor) for h in h_list):
            # No records, C8 path: nothing of the previous batch may be REUSED, but its buffers can be written
            # again -- the session of this geometry is handed out unprimed (the first step computes every map in
            # full, into the same tensors), so a refinement step captured as a HIP graph on it stays valid for
            # the next batch (api._refine_graph; e.g. the FC-DenseNet host, whose h carries no record).
            key = (tuple(tuple(h.shape) for h in h_list), tuple(y.shape), y.dtype)
            st = self._scratch_sessions.get(key)
            if st is None:
                while len(self._scratch_sessions) >= 2:
                    self._scratch_sessions.pop(next(iter(self._scratch_sessions)))
                st = self._scratch_sessions[key] = {'stable': True}
            else:
                self._scratch_sessions[key] = self._scratch_sessions.pop(key)
            st.update(primed=False, h_stale=True, y8_fresh=False, h_fresh=False)
            return st
        return {'primed': False}

    @property
    def stable_sessions(self):
        """Sessions without provenance records keep their buffers from batch to batch (C8 path): a captured
        refinement step stays valid, so api.refine replays it for short loops too."""
        return bool(self.c8)

    def conv_layers(self):
        d = dict(self.enc)
        d.update(self.dec)
        return d

    def scores(self, h_list, y, mask_override=None, session=None):
        """Runs the DAE up to the pre-softmax score map already cropped to y's size
        (fused_up1 of fcn_up.py:104-113).  Returns score (B, n_classes, H, W).
        `mask_override` {level: (pre, pooled)} substitutes the tensors whose equality defines
        the DePool2D mask of that level (parity tests use it to inject reference masks).
        `session` (a dict from `new_session()`) makes consecutive calls with the SAME h recompute
        only what depends on y (used by the refinement loop)."""
        h_list = list(h_list)
        if len(h_list) != len(self.concat_h):
            raise ValueError('expected %d h tensors, got %d' % (len(self.concat_h), len(h_list)))
        if self.c8:
            if mask_override is not None:
                raise NotImplementedError("mask injection needs fp32 activations (mma='bf16' / 'f32')")
            return self._scores_c8(h_list, y, session)
        pos = 0
        pending_h = None
        h_fresh = session is not None and session.get('h_fresh', False)
        if self.concat_h[pos] == 'input':                # model_helpers.py:86-94 at the input
            pending_h, pos = h_list[pos], pos + 1
        t = y
        pre, pool = {}, {0: y}
        # Loop-invariant code motion for the refinement loop: between two steps only y changes, so
        # only the part of every encoder map that y can reach has to be recomputed (the pad-100
        # border and everything fed by h alone keep the values of the first step).  `session`
        # carries the full-size buffers; `dep` is the y-dependent region (y0, x0, h, w) of `t`.
        primed = session is not None and session.get('primed', False) and self.licm
        will_override = mask_override is not None or (
            self.emulate_noise and (self.noise > 0 or self.dropout > 0) and
            self.unpool_type == 'trackind')
        masked = self._mask_levels(will_override)
        if session is not None:
            if primed and session.get('masked', frozenset()) != masked:
                primed = False               # buffers of the other form: compute everything again
            session['masked'] = masked
            if not primed:
                # the session's buffers are (re)allocated by this call: anything that captured
                # pointers into the previous ones (api._refine_graph) must notice
                session['gen'] = session.get('gen', 0) + 1
        masks = {}
        dep = (0, 0, y.shape[2], y.shape[3])
        ydep = dep    # the region y alone reaches: its origin parity anchors the Winograd tiles

        def clip(lo, hi, size):
            lo, hi = max(lo, 0), min(hi, size)
            return lo, max(hi - lo, 0)

        for p in range(self.total):                      # fcn_down.py:77-136
            fused_pool = None
            for i in range(1, self.conv_before_pool + 1):
                name = 'conv%d_%d' % (p + 1, i)
                conv = self.enc[name]
                fh, fw = conv.out_hw(t.shape[2], t.shape[3])
                ydep = (clip(ydep[0] + conv.pad - (conv.KH - 1), ydep[0] + ydep[2] + conv.pad, fh) +
                        clip(ydep[1] + conv.pad - (conv.KW - 1), ydep[1] + ydep[3] + conv.pad, fw))
                ydep = (ydep[0], ydep[2], ydep[1], ydep[3])
                kw = dict(anchor=(ydep[0], ydep[1]))
                if primed and pending_h is not None and h_fresh:
                    # a new batch in a reused session: h changed inside its tagged region
                    hd = session['h_dep'][pos - 1]
                    y1 = max(dep[0] + dep[2], hd[0] + hd[2])
                    x1 = max(dep[1] + dep[3], hd[1] + hd[3])
                    dep = (min(dep[0], hd[0]), min(dep[1], hd[1]), 0, 0)
                    dep = (dep[0], dep[1], y1 - dep[0], x1 - dep[1])
                if primed:
                    buf = session[name]
                    fh, fw = buf.shape[2], buf.shape[3]
                    wy0, wh = clip(dep[0] + conv.pad - (conv.KH - 1), dep[0] + dep[2] + conv.pad, fh)
                    wx0, ww = clip(dep[1] + conv.pad - (conv.KW - 1), dep[1] + dep[3] + conv.pad, fw)
                    dep = (wy0, wx0, wh, ww)
                    kw.update(window=dep, out=buf, place=(wy0, wx0))
                if pending_h is not None and name in self.hsplit:
                    conv_h, conv_y = self.hsplit[name]
                    keep = session is not None and self.licm
                    hb = session.get('hb_' + name) if keep else None
                    if hb is None:                       # loop-invariant: once per refine()
                        hb = conv_h(pending_h)
                        if keep:
                            session['hb_' + name] = hb
                    elif h_fresh:                        # reused session: only where h changed
                        hd = session['h_dep'][pos - 1]
                        hy0, hh = clip(hd[0] + conv_h.pad - (conv_h.KH - 1), hd[0] + hd[2] + conv_h.pad,
                                       hb.shape[2])
                        hx0, hw = clip(hd[1] + conv_h.pad - (conv_h.KW - 1), hd[1] + hd[3] + conv_h.pad,
                                       hb.shape[3])
                        conv_h(pending_h, window=(hy0, hx0, hh, hw), out=hb, place=(hy0, hx0))
                    off = kw['place'] if 'place' in kw else (0, 0)
                    t = conv_y(t, add=hb, add_off=off, **kw)
                    pending_h = None
                elif pending_h is not None:              # h first, then features (P13)
                    t = conv(pending_h, x2=t, **kw)
                    pending_h = None
                else:
                    # last conv of the level on the halo kernel: its epilogue also writes pool_p
                    # (the window is widened to whole pooling windows; the extra row / column is
                    # recomputed to the values it already has)
                    pw_ = None
                    if i == self.conv_before_pool and not self.bn:
                        pw_ = conv.pool_window(t.shape[2], t.shape[3], dep if primed else None, c8=False)
                    if pw_ is not None:
                        fh, fw = conv.out_hw(t.shape[2], t.shape[3])
                        if primed:
                            pooled_t = session['pool%d' % (p + 1)]
                            kw.update(window=pw_, place=(pw_[0], pw_[1]))
                        else:
                            pooled_t = torch.empty((t.shape[0], conv.Cout, fh // 2, fw // 2),
                                                   dtype=t.dtype, device=t.device)
                        fused_pool = pooled_t
                        kw['pool_out'] = pooled_t
                    if (p + 1) in masked:
                        if pw_ is None:
                            raise RuntimeError('internal: level %d was planned for byte masks but '
                                               'its pool is not fused' % (p + 1))
                        m = session['mask%d' % (p + 1)] if primed else \
                            torch.empty(pooled_t.shape, dtype=torch.uint8, device=t.device)
                        masks[p + 1] = m
                        kw.update(mask_out=m, store_out=False)
                        conv(t, **kw)
                        # the pre-pool map is not stored: a shape-only stand-in from here on
                        t = torch.empty((t.shape[0], conv.Cout, fh, fw), dtype=t.dtype, device='meta')
                    else:
                        t = conv(t, **kw)
                if self.bn:
                    ops.bn_affine(t, self.enc_bn[name], window=dep if primed else None)
                if session is not None and not primed:
                    session[name] = t
                    if (p + 1) in masks and i == self.conv_before_pool:
                        session['mask%d' % (p + 1)] = masks[p + 1]
                self._count(name, conv, t, computed=(dep[2], dep[3]) if primed else None)
            pre[p + 1] = t
            ydep = (ydep[0] // 2, ydep[1] // 2,
                    min((ydep[0] + ydep[2] + 1) // 2, t.shape[2] // 2) - ydep[0] // 2,
                    min((ydep[1] + ydep[3] + 1) // 2, t.shape[3] // 2) - ydep[1] // 2)
            if primed:
                buf = session['pool%d' % (p + 1)]
                qy0, qh = clip(dep[0] // 2, (dep[0] + dep[2] + 1) // 2, buf.shape[2])
                qx0, qw = clip(dep[1] // 2, (dep[1] + dep[3] + 1) // 2, buf.shape[3])
                dep = (qy0, qx0, qh, qw)
                t = buf if fused_pool is not None else ops.maxpool2x2(t, out=buf, window=dep)
            else:
                t = fused_pool if fused_pool is not None else ops.maxpool2x2(t)   # :122
                if session is not None:
                    session['pool%d' % (p + 1)] = t
            pool[p + 1] = t
            if p < self.n_pool and pos < len(self.concat_h) and \
                    self.concat_h[pos] == 'pool%d' % (p + 1):   # :131-134
                pending_h, pos = h_list[pos], pos + 1
        if session is not None:
            session['primed'] = True
            session['h_fresh'] = False
        if pending_h is not None:
            raise NotImplementedError('h concatenated at the last pool feeds DePool2D directly '
                                      '(additional_pool=0); not shape-consistent in the reference')
        # the hidden re-forward is stochastic whenever ANY stochastic layer is live: Gaussian noise
        # (noise > 0) or the DropoutLayers (dropout > 0, even at noise == 0 -- the configuration of
        # the reference's golden experiment name, plots.ipynb:84: dropout 0.5, z0)
        if self.emulate_noise and (self.noise > 0 or self.dropout > 0) and \
                self.unpool_type == 'trackind' and mask_override is None:
            mask_override = self.hidden_masks(h_list, y)
        if self.unpool_type == 'standard':
            # fcn_up.py:37-63: up_p = Deconv2DLayer(prev, n_cl, 4, stride=2, crop='valid', linear),
            # then ElemwiseSumLayer with pool_{p-1} (center crop) or CroppingLayer.  The 4x4/2
            # transposed conv runs on the static-tap conv kernel; crop + sum are its window + add.
            for p in range(self.total, 0, -1):
                name = 'up%d' % p
                conv = self.dec[name]
                other = pool[p - 1]
                uh, uw = conv.out_hw(t.shape[2], t.shape[3])
                oh, ow = min(uh, other.shape[2]), min(uw, other.shape[3])
                kw = dict(window=(_center(uh, oh), _center(uw, ow), oh, ow))
                if self.skip and p > 1:
                    kw.update(add=other, add_off=(_center(other.shape[2], oh),
                                                  _center(other.shape[3], ow)))
                t = conv(t, **kw)
                self._count(name, conv, t, full=(uh, uw))
            return t
        # ---- decoder, fcn_up.py:143-151 / UnpoolNet ------------------------------------------
        # Only the final center crop (fused_up1) is an output, so each level is computed just on
        # the window that reaches it (dead-code elimination, bit-identical results): level p
        # needs fused_up_{p+1} on [floor((lo-1)/2), ceil((hi+1)/2)).  Tensors keep their full-size
        # addressing; windows are written in place (`place`), the rest is never read.
        geom = {}
        for p in range(self.total, 0, -1):
            ph, pw = pre[p].shape[2], pre[p].shape[3]    # up_conv 'same' keeps the pre-pool size
            other = pool[p - 1]                          # pre-concat pool (or the input for p=1)
            oh, ow = min(ph, other.shape[2]), min(pw, other.shape[3])
            geom[p] = (ph, pw, oh, ow, _center(ph, oh), _center(pw, ow))
        win = {1: (0, 0, geom[1][2], geom[1][3])}        # (y0, x0, h, w) in fused_up_p coords
        for p in range(1, self.total):
            ph, pw, oh, ow, cy, cx = geom[p]
            y0, x0, nh, nw = win[p]
            uy0, ux0 = max(cy + y0 - 1, 0), max(cx + x0 - 1, 0)          # unpooled input rows/cols
            uy1, ux1 = min(cy + y0 + nh + 1, ph), min(cx + x0 + nw + 1, pw)
            qh, qw = geom[p + 1][2], geom[p + 1][3]                      # fused_up_{p+1} dims
            ny0, nx0 = uy0 // 2, ux0 // 2
            ny1, nx1 = min((uy1 + 1) // 2, qh), min((ux1 + 1) // 2, qw)
            win[p + 1] = (ny0, nx0, ny1 - ny0, nx1 - nx0)
        # with DCE off the full maps are computed, but the Winograd tiles stay anchored at the
        # parity of the DCE windows so that both modes agree bit for bit
        need = win if self.dce else {p: (0, 0, geom[p][2], geom[p][3]) for p in geom}
        for p in range(self.total, 0, -1):
            name = 'up_conv%d' % p
            conv = self.dec[name]
            ph, pw, oh, ow, cy, cx = geom[p]
            other = pool[p - 1]
            y0, x0, nh, nw = need[p]
            window = (cy + y0, cx + x0, nh, nw)
            mpre, mpool = pre[p], pool[p]
            if mask_override and p in mask_override:
                mpre, mpool = mask_override[p]
            fuse = self.fuse_unpool
            if fuse is None:
                # fused in float32; in float64 where the layer runs in Winograd form (its input
                # transform applies the mask) or on the halo-tile kernel (its patch staging does);
                # materialised for the static-tap float64 kernel
                fuse = conv.dtype == torch.float32 or getattr(conv, 'wino_f64', False) or \
                    (F64_FUSE_UNPOOL and (conv.KH, conv.KW, conv.dil) == (3, 3, 1) and not conv.transposed)
            if not fuse:
                # materialise DePool2D with the HBM-bound kernel and run the plain conv
                uy0, ux0 = max(cy + y0 - 1, 0), max(cx + x0 - 1, 0)
                uy1, ux1 = min(cy + y0 + nh + 1, ph), min(cx + x0 + nw + 1, pw)
                u = torch.empty_like(mpre)
                t = ops.unpool_eqmask(t, mpre, mpool, out=u, window=(uy0, ux0, uy1 - uy0, ux1 - ux0))
                mpre = mpool = None
            full = (nh, nw) == (oh, ow)
            out = None if full else torch.empty((y.shape[0], conv.Cout, oh, ow), dtype=y.dtype,
                                                device=y.device)
            kw = dict(pre=mpre, pooled=mpool, window=window, out=out,
                      place=None if full else (y0, x0),
                      anchor=(cy + win[p][0], cx + win[p][1]))
            if p in masks:
                kw.update(pre=None, pooled=None, mask_in=masks[p], unpool_hw=(ph, pw))
            if self.skip and p > 1:                      # :96-102 ElemwiseSumLayer, center crop
                kw.update(add=other, add_off=(_center(other.shape[2], oh) + y0,
                                              _center(other.shape[3], ow) + x0))
            t = conv(t, **kw)                            # else :104-113 CroppingLayer
            self._count(name, conv, t, full=(ph, pw), computed=(nh, nw))
            if self.trace is not None:
                self.trace['fused_up%d' % p] = t
                self.trace['need%d' % p] = need[p]
        if self.trace is not None:
            self.trace.update({'pre%d' % k: v for k, v in pre.items()})
            self.trace.update({'pool%d' % k: v for k, v in pool.items() if k > 0})
        self._saved = (mask_override, pre, pool)      # what backward_y needs (masks only)
        return t

    def c8_feed(self, session):
        """The C8 buffer a fused refinement update may write the new y into for the NEXT `scores`
        call of this session (ops.refine_update(..., y8=)), or None; `c8_fed(session)` afterwards."""
        if not self.c8 or self.x3 or not isinstance(session, dict):
            return None                  # (x3: y is converted to its pair per call, 0.05 ms)
        return session.get('y8')

    def c8_fed(self, session):
        session['y8_fresh'] = True

    def _scores_c8(self, h_list, y, session):
        """`scores` with bf16 C8 activations between the layers (csrc/conv_c8_bf16.hip): the same
        layer plan, windows (decoder dead-code elimination, loop-invariant encoder maps, border
        stores) and fusions, in the form every level takes here --
          encoder level p : conv3x3 + ReLU whose epilogue writes pool_p (C8) and the DePool2D mask
                            bytes of its windows, taken from the fp32 results; the pre-pool map is
                            never stored (fcn_down.py:102-122);
          decoder level p : conv3x3 whose input staging IS DePool2D (fused_up_{p+1} chunk + mask
                            bytes, layers/mylayers.py:88-115), skip sum with pool_{p-1} and the crop
                            in the epilogue (fcn_up.py:64-113); the last one writes fp32 NCHW scores.
        y arrives fp32 NCHW and is converted once per call; the h half of the conv behind a concat
        point is a cached fp32 C8 addend (loop-invariant, `hsplit`)."""
        if self.conv_before_pool != 1 or self.bn or self.unpool_type == 'standard' or \
                self.trace is not None or self.keep_pre or \
                (self.emulate_noise and (self.noise > 0 or self.dropout > 0)):
            raise NotImplementedError("mma='bf16c8' runs the plain trackind / inverse DAE "
                                      "(conv_before_pool=1, bn=0, no trace / gradient mode / noise "
                                      "emulation): use mma='bf16' for those")
        B, dev = y.shape[0], y.device
        pos, pending_h = 0, None
        h_fresh = session is not None and session.get('h_fresh', False)
        if self.concat_h[pos] == 'input':
            pending_h, pos = h_list[pos], pos + 1
        primed = session is not None and session.get('primed', False) and self.licm
        if session is not None:
            if primed and session.get('masked') != 'c8':
                primed = False
            session['masked'] = 'c8'
        # (`gen` tells a captured graph that buffers were allocated anew; a buffer-stable session -- new_session
        # without records -- writes the tensors it already has)
        stable = session is not None and session.get('stable', False)
        realloc = session is not None and not primed and not stable
        h_stale = stable and session.get('h_stale', False)
        # y as a C8 tensor: converted here, unless the refinement step that produced y has already
        # written it (`c8_feed`: api passes the session's buffer to ops.refine_update)
        if session is not None and session.get('y8_fresh') and session.get('y8') is not None and \
                tuple(session['y8'].shape[2:4]) == tuple(y.shape[2:4]):
            t = session['y8']
        else:
            t = ops.nchw_to_c8(y, out=session.get('y8') if session is not None and
                               session.get('y8') is not None and
                               tuple(session['y8'].shape[2:4]) == tuple(y.shape[2:4]) and
                               session['y8'].shape[0] == B else None, x3=self.x3)
            if session is not None:
                session['y8'] = t
        if session is not None:
            session['y8_fresh'] = False
        pre_hw, pool8, masks = {}, {}, {}
        pool_hw = {0: (y.shape[2], y.shape[3])}
        dep = (0, 0, y.shape[2], y.shape[3])

        def clip(lo, hi, size):
            lo, hi = max(lo, 0), min(hi, size)
            return lo, max(hi - lo, 0)

        for p in range(self.total):                      # fcn_down.py:77-136
            name = 'conv%d_1' % (p + 1)
            conv = self.enc[name]
            fh, fw = conv.out_hw(t.shape[2], t.shape[3])
            pre_hw[p + 1] = (fh, fw)
            if primed and pending_h is not None and h_fresh:
                hd = session['h_dep'][pos - 1]           # a new batch: h changed inside its region
                y1 = max(dep[0] + dep[2], hd[0] + hd[2])
                x1 = max(dep[1] + dep[3], hd[1] + hd[3])
                dep = (min(dep[0], hd[0]), min(dep[1], hd[1]), 0, 0)
                dep = (dep[0], dep[1], y1 - dep[0], x1 - dep[1])
            kw = {}
            if primed:
                wy0, wh = clip(dep[0] + conv.pad - 2, dep[0] + dep[2] + conv.pad, fh)
                wx0, ww = clip(dep[1] + conv.pad - 2, dep[1] + dep[3] + conv.pad, fw)
                dep = (wy0, wx0, wh, ww)
                kw['window'] = conv.pool_window(t.shape[2], t.shape[3], dep)
                pooled_t, m = session['pool%d' % (p + 1)], session['mask%d' % (p + 1)]
            else:
                pooled_t = m = None
                if stable:
                    pooled_t, m = session.get('pool%d' % (p + 1)), session.get('mask%d' % (p + 1))
                    nch = ops.c8_chunks(conv.Cout)
                    if pooled_t is None or m is None or \
                            tuple(pooled_t.shape) != (B, nch * (2 if self.x3 else 1), fh // 2, fw // 2, 8) or \
                            tuple(m.shape) != (B, nch, fh // 2, fw // 2, 8):
                        pooled_t = m = None
                if pooled_t is None:
                    pooled_t = ops.empty_c8(B, conv.Cout, fh // 2, fw // 2, dev, x3=self.x3)
                    m = ops.empty_c8(B, conv.Cout, fh // 2, fw // 2, dev, dtype=torch.uint8)
                    realloc = realloc or stable
            kw.update(pool_out=pooled_t, mask_out=m, store_out=False)
            if pending_h is not None and name in self.hsplit:
                conv_h, conv_y = self.hsplit[name]
                keep = session is not None and self.licm
                hb = session.get('hb_' + name) if keep else None
                if hb is None:                           # loop-invariant: once per refine()
                    hb = conv_h(ops.nchw_to_c8(pending_h, x3=self.x3), out_format='c8f32')
                    if keep:
                        session['hb_' + name] = hb
                    realloc = realloc or stable
                elif h_stale:                            # buffer-stable session, new batch: all of it again
                    conv_h(ops.nchw_to_c8(pending_h, x3=self.x3), window=(0, 0, hb.shape[2], hb.shape[3]),
                           out=hb, place=(0, 0), out_format='c8f32')
                elif h_fresh:                            # reused session: only where h changed
                    hd = session['h_dep'][pos - 1]
                    hy0, hh = clip(hd[0] + conv_h.pad - 2, hd[0] + hd[2] + conv_h.pad, hb.shape[2])
                    hx0, hw = clip(hd[1] + conv_h.pad - 2, hd[1] + hd[3] + conv_h.pad, hb.shape[3])
                    conv_h(ops.nchw_to_c8(pending_h, x3=self.x3), window=(hy0, hx0, hh, hw), out=hb,
                           place=(hy0, hx0), out_format='c8f32')
                off = (kw['window'][0], kw['window'][1]) if 'window' in kw else (0, 0)
                conv_y(t, add=hb, add_off=off, **kw)
                pending_h = None
            elif pending_h is not None:                  # h first, then features (P13)
                if self.x3:
                    raise NotImplementedError("mma='bf16x3' needs the h-split form of a concat "
                                              "point (hsplit=True)")
                conv(ops.nchw_to_c8(pending_h), x2=t, **kw)
                pending_h = None
            else:
                conv(t, **kw)
            if session is not None and not primed:
                session['pool%d' % (p + 1)], session['mask%d' % (p + 1)] = pooled_t, m
            if self.conv_log is not None:
                cw_ = kw['window'] if 'window' in kw else (0, 0, fh, fw)
                self.conv_log.append((name, conv.flops(B, fh, fw), conv.flops(B, cw_[2], cw_[3])))
            pool8[p + 1], masks[p + 1] = pooled_t, m
            pool_hw[p + 1] = (fh // 2, fw // 2)
            if primed:
                qy0, qh = clip(dep[0] // 2, (dep[0] + dep[2] + 1) // 2, fh // 2)
                qx0, qw = clip(dep[1] // 2, (dep[1] + dep[3] + 1) // 2, fw // 2)
                dep = (qy0, qx0, qh, qw)
            t = pooled_t
            if p < self.n_pool and pos < len(self.concat_h) and \
                    self.concat_h[pos] == 'pool%d' % (p + 1):   # :131-134
                pending_h, pos = h_list[pos], pos + 1
        if session is not None:
            session['primed'] = True
            session['h_fresh'] = False
            session['h_stale'] = False
            if realloc:
                session['gen'] = session.get('gen', 0) + 1
        if pending_h is not None:
            raise NotImplementedError('h concatenated at the last pool feeds DePool2D directly '
                                      '(additional_pool=0); not shape-consistent in the reference')
        # ---- decoder, fcn_up.py:143-151 (windows as in `scores`) -------------------------------
        geom = {}
        for p in range(self.total, 0, -1):
            ph, pw = pre_hw[p]
            oh, ow = min(ph, pool_hw[p - 1][0]), min(pw, pool_hw[p - 1][1])
            geom[p] = (ph, pw, oh, ow, _center(ph, oh), _center(pw, ow))
        win = {1: (0, 0, geom[1][2], geom[1][3])}
        for p in range(1, self.total):
            ph, pw, oh, ow, cy, cx = geom[p]
            y0, x0, nh, nw = win[p]
            uy0, ux0 = max(cy + y0 - 1, 0), max(cx + x0 - 1, 0)
            uy1, ux1 = min(cy + y0 + nh + 1, ph), min(cx + x0 + nw + 1, pw)
            qh, qw = geom[p + 1][2], geom[p + 1][3]
            ny0, nx0 = uy0 // 2, ux0 // 2
            ny1, nx1 = min((uy1 + 1) // 2, qh), min((ux1 + 1) // 2, qw)
            win[p + 1] = (ny0, nx0, ny1 - ny0, nx1 - nx0)
        need = win if self.dce else {p: (0, 0, geom[p][2], geom[p][3]) for p in geom}
        for p in range(self.total, 0, -1):
            name = 'up_conv%d' % p
            conv = self.dec[name]
            ph, pw, oh, ow, cy, cx = geom[p]
            y0, x0, nh, nw = need[p]
            full = (nh, nw) == (oh, ow)
            out = None
            if not full:
                out = torch.empty((B, conv.Cout, oh, ow), dtype=torch.float32, device=dev) if p == 1 \
                    else ops.empty_c8(B, conv.Cout, oh, ow, dev, x3=self.x3)
            kw = dict(mask_in=masks[p], unpool_hw=(ph, pw), window=(cy + y0, cx + x0, nh, nw),
                      out=out, place=None if full else (y0, x0),
                      out_format='nchw' if p == 1 else 'c8')
            if self.skip and p > 1:                      # :96-102 ElemwiseSumLayer, center crop
                oth = pool_hw[p - 1]
                kw.update(add=pool8[p - 1], add_off=(_center(oth[0], oh) + y0, _center(oth[1], ow) + x0))
            if 0 < C8_UNPOOL_MIN_CIN <= conv.Cin and not self.x3 and conv.Cout > 16:
                # Deep levels: DePool2D materialised first (a few tens of MB at these sizes), the conv then runs
                # as a PLAIN layer -- LDS-DMA patch staging instead of up chunk + mask bytes selected through
                # registers by every output-channel tile of every pixel tile (same values: bit-identical).  The
                # map is kept per session (rows / columns past the last pooling window stay zero).
                skey = 'unp%d' % p
                u = session.get(skey) if session is not None else None
                if u is None or tuple(u.shape) != (B, t.shape[1], ph, pw, 8):
                    u = torch.zeros((B, t.shape[1], ph, pw, 8), dtype=torch.bfloat16, device=dev)
                    if session is not None:
                        session[skey] = u
                wy0, wx0 = max(cy + y0 - 1, 0), max(cx + x0 - 1, 0)            # input rows / columns the window reads
                wy1, wx1 = min(cy + y0 + nh + 1, 2 * (ph // 2)), min(cx + x0 + nw + 1, 2 * (pw // 2))
                qy0, qx0 = wy0 // 2, wx0 // 2
                ops.unpool_c8(t, masks[p], u, window=(qy0, qx0, (wy1 + 1) // 2 - qy0, (wx1 + 1) // 2 - qx0))
                kw.pop('mask_in'); kw.pop('unpool_hw')
                t = conv(u, **kw)
            else:
                t = conv(t, **kw)                        # else :104-113 CroppingLayer
            if self.conv_log is not None:
                self.conv_log.append((name, conv.flops(B, ph, pw), conv.flops(B, nh, nw)))
        self._saved = None
        return t

    # ---- true-gradient mode (not in the reference, F1) -----------------------
    def _bwd_convs(self):
        """Backward-data of a 3x3 stride-1 conv = the forward conv of the gradient with the
        channel-transposed, spatially flipped filter ('same' for pad 1; for the pad-100 first layer
        the window at offset pad-1 of that 'same' result).  After a concat only the filters of the
        non-h channels are kept: h is a constant of the loop."""
        if self._bwd is None:
            if self.conv_before_pool != 1 or self.bn or self.unpool_type == 'standard':
                raise NotImplementedError('gradient mode: conv_before_pool=1, bn=0, unpool_type in '
                                          '{trackind, inverse}')
            bwd = {}
            prev = self.n_classes
            for p in range(self.total):
                name = 'conv%d_1' % (p + 1)
                W = self.enc[name].W                     # (Cout, ch + prev, 3, 3), on the device
                ch = W.shape[1] - prev
                Wt = W[:, ch:].transpose(0, 1).flip(2, 3).contiguous()
                bwd[name] = ops.Conv(Wt, None, pad=1, relu=False, device=W.device, dtype=self.dtype)
                prev = W.shape[0]
            for p in range(self.total, 0, -1):
                name = 'up_conv%d' % p
                W = self.dec[name].W
                bwd[name] = ops.Conv(W.transpose(0, 1).flip(2, 3).contiguous(), None, pad=1,
                                     relu=False, device=W.device, dtype=self.dtype)
            self._bwd = bwd
        return self._bwd

    def backward_y(self, g_score, y_shape):
        """dE/dy THROUGH the DAE for an upstream gradient g_score w.r.t. the (cropped) score map
        of the latest `scores()` call; adjoint of that forward, level by level (oracle/dae_grad.py).
        Full maps (the decoder / encoder windows of the forward are not exploited here)."""
        bwd = self._bwd_convs()
        override, pre, pool = self._saved
        if any(t.device.type == 'meta' for t in pre.values()):
            raise RuntimeError('backward_y needs the pre-pool maps of the forward pass: set '
                               'dae.keep_pre = True before calling scores()')
        B = g_score.shape[0]
        dev, dt = g_score.device, g_score.dtype
        g_pool = {}
        g_f = g_score
        for p in range(1, self.total + 1):               # decoder, output to input
            mpre, mpool = pre[p], pool[p]
            if override and p in override:
                mpre, mpool = override[p]
            ph, pw = pre[p].shape[2], pre[p].shape[3]
            other_hw = (pool[p - 1].shape[2], pool[p - 1].shape[3]) if p > 1 else \
                (y_shape[2], y_shape[3])
            oh, ow = min(ph, other_hw[0]), min(pw, other_hw[1])
            cy, cx = _center(ph, oh), _center(pw, ow)
            g_c = torch.zeros((B, g_f.shape[1], ph, pw), dtype=dt, device=dev)
            g_c[:, :, cy:cy + oh, cx:cx + ow].copy_(g_f)           # adjoint of the center crop
            if self.skip and p > 1:                                 # adjoint of the skip sum
                gp = torch.zeros_like(pool[p - 1])
                oy, ox = _center(other_hw[0], oh), _center(other_hw[1], ow)
                gp[:, :, oy:oy + oh, ox:ox + ow].copy_(g_f)
                g_pool[p - 1] = gp
            g_u = bwd['up_conv%d' % p](g_c)
            g_f = ops.depool_bwd(g_u, mpre, mpool)
        g_pool[self.total] = g_f
        g_in = None
        for p in range(self.total, 0, -1):               # encoder, deep to shallow
            gp = g_pool.get(p)
            if gp is None:
                gp = torch.zeros_like(pool[p])
            g_z = ops.pool_relu_bwd(gp, pre[p], pool[p])
            conv = bwd['conv%d_1' % p]
            if p > 1:
                acc = g_pool.get(p - 1)
                g_pool[p - 1] = conv(g_z) if acc is None else conv(g_z, add=acc, out=acc)
            else:
                fwd = self.enc['conv1_1']
                if fwd.pad != 1:      # pad-100 first layer: crop at offset pad - 1
                    g_in = conv(g_z, window=(fwd.pad - 1, fwd.pad - 1, y_shape[2], y_shape[3]))
                else:
                    g_in = conv(g_z)
        return g_in

    def _rand(self, kind, level, name, shape, like):
        if self.random_source is not None:
            t = self.random_source(kind, level, name, tuple(shape))
            return torch.as_tensor(t).to(like.dtype).contiguous().to(like.device)
        if self._gen is None:
            self._gen = torch.Generator(device=like.device)
            self._gen.manual_seed(self._seed)
        if kind == 'noise':
            return torch.randn(shape, generator=self._gen, device=like.device, dtype=like.dtype)
        u = torch.rand(shape, generator=self._gen, device=like.device, dtype=like.dtype)
        return (u >= self.dropout).to(like.dtype)     # keep mask

    def hidden_masks(self, h_list, y):
        """{level: (pre, pooled)} as DePool2D sees them when dae_dict['noise'] > 0 or
        dae_dict['dropout'] > 0: every DePool2D
        calls lasagne.layers.get_output([pool_in, pool]) WITHOUT deterministic=True
        (layers/mylayers.py:91-93), i.e. a fresh forward of the down path up to its level with
        GaussianNoiseLayer (fcn_down.py:60-63) and the DropoutLayers (:108-111) active.  The
        main path (the values that are unpooled) stays deterministic."""
        if self.bn:
            raise NotImplementedError('noise emulation with bn=1 (batch statistics in the hidden '
                                      'forward) is not supported')
        masks = {}
        for p in range(self.total, 0, -1):
            # GaussianNoiseLayer(sigma=0) adds nothing (and draws nothing worth emulating)
            t = ops.add_noise(y, self._rand('noise', p, None, y.shape, y), self.noise) \
                if self.noise > 0 else y
            pos, pending = 0, None
            if self.concat_h[0] == 'input':
                pending, pos = h_list[0], 1
            for q in range(p):
                for i in range(1, self.conv_before_pool + 1):
                    name = 'conv%d_%d' % (q + 1, i)
                    conv = self.enc[name]
                    t = conv(pending, x2=t) if pending is not None else conv(t)
                    pending = None
                    if self.dropout > 0:
                        ops.dropout_apply(t, self._rand('dropout', p, name, t.shape, t), self.dropout)
                pre_q = t
                t = ops.maxpool2x2(t)
                if q < self.n_pool and pos < len(self.concat_h) and \
                        self.concat_h[pos] == 'pool%d' % (q + 1):
                    pending, pos = h_list[pos], pos + 1
            masks[p] = (pre_q, t)
        return masks

    def _count(self, name, conv, out, full=None, computed=None):
        # (name, nominal FLOPs of the FULL layer output, FLOPs of the computed window)
        if self.conv_log is not None:
            oh, ow = full if full is not None else (out.shape[2], out.shape[3])
            ch, cw = computed if computed is not None else (out.shape[2], out.shape[3])
            self.conv_log.append((name, conv.flops(out.shape[0], oh, ow),
                                  conv.flops(out.shape[0], ch, cw)))

    def __call__(self, *args):
        """pred_dae_fn(h..., y) -> r  (softmax output, models/fcn_up.py:154-169)."""
        h_list, y = args[:-1], args[-1]
        score = self.scores(h_list, y)
        return ops.crop_softmax(score, score.shape[2], score.shape[3], off=(0, 0))

    def residual(self, *args):
        """de_fn(h..., y) = -(r - y) = y - r  (iterative_inference.py:203-204)."""
        h_list, y = args[:-1], args[-1]
        score = self.scores(h_list, y)
        return ops.crop_softmax(score, score.shape[2], score.shape[3], off=(0, 0), minuend=y)


def buildDAE(input_concat_h_vars=None, input_mask_var=None, n_classes=11,
             nb_features_to_concat=None, padding=100, ae_h=False, void_labels=(),
             path_weights=None, model_name='dae_model.npz', trainable=False, load_weights=False,
             out_nonlin='softmax', concat_h=('input',), noise=0.1, n_filters=64,
             conv_before_pool=1, additional_pool=0, dropout=0., skip=False,
             unpool_type='standard', bn=0, params=None, device='cuda', dtype=torch.float32,
             emulate_noise=False):
    """Mirror of models/DAE_h.py:12-20.  Inference only: `noise`, `dropout` are identities at
    deterministic=True (P8, P9); the DePool2D masks are the deterministic ones unless
    `emulate_noise` asks for the reference's noisy hidden re-forward; the symbolic
    `input_*_var`, `trainable`, `ae_h`, `void_labels` are accepted and ignored."""
    import os
    if params is None:
        if not (load_weights and path_weights):
            raise ValueError('buildDAE needs `params` or `path_weights`')
        order = param_order(concat_h, conv_before_pool, additional_pool, unpool_type, bn)
        params = load_param_list(os.path.join(path_weights, model_name), order)  # DAE_h.py:52-57
    if out_nonlin not in ('softmax',):
        raise NotImplementedError('inference uses out_nonlin=softmax (iterative_inference.py:158)')
    return StandardDAE(params, n_classes, concat_h=concat_h, padding=padding, n_filters=n_filters,
                       conv_before_pool=conv_before_pool, additional_pool=additional_pool,
                       skip=skip, unpool_type=unpool_type, bn=bn, device=device, dtype=dtype,
                       noise=noise, dropout=dropout, emulate_noise=emulate_noise)
