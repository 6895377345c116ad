"""Context-module DAE on the HIP kernels (mirror of reference models/contextmod_dae.py:19-138):
conv3x3(+ReLU) on [h=image, y] -> pad 32 -> six dilated 3x3 convs (dilation 1,2,4,8,16,1, ReLU)
-> 1x1 linear -> softmax, 11 channels throughout.  The PadLayer is the `pad` of the first dilated
conv; the 3x3 layers (dilation 1..16) run on the 16-channel halo kernel (`conv_halo16`, 16x16x4
MFMA; DilatedConv2DLayer weight layout W[in,out,k,k], P11); inside a refinement loop the concat is
a buffer whose y channels are refreshed per step, outside it the two-source gather."""
import os

import torch

from . import ops
from .weights import load_param_list

# inside a refinement loop the image half of the first layer (3 of its 14 input channels; h is the image and does not
# change) is computed once per batch and the y half CONTINUES its FMA chain from that map (bit-identical: the
# chain is bias, then the channels in order, h first -- P13); 0: the whole layer every step
CTX_HSPLIT = os.environ.get('IISEG_CTX_HSPLIT', '1') != '0'

PARAM_ORDER = ['conv1'] + ['dilconv%d' % i for i in range(1, 8)]   # get_all_param_values (P14)
DILATIONS = [1, 2, 4, 8, 16, 1]                                     # contextmod_dae.py:78-101


class ContextModDAE:
    def __init__(self, params, n_classes, concat_h=('input',), device='cuda',
                 dtype=torch.float32):
        assert all(el in ['input'] for el in concat_h)               # contextmod_dae.py:42
        if len(concat_h) != 1:
            raise NotImplementedError('one h (the image) is concatenated at the input')
        self.concat_h = list(concat_h)
        self.conv1 = ops.Conv(params['conv1'][0], params['conv1'][1], pad=1, relu=True,
                              device=device, dtype=dtype)                         # :74-76
        self.dil = []
        pad = 32                                                     # PadLayer(width=32), :77
        for i, d in enumerate(DILATIONS):
            W, b = params['dilconv%d' % (i + 1)]
            self.dil.append(ops.Conv(W, b, pad=pad, relu=True, dil=d, layout='iohw', device=device,
                                     dtype=dtype))
            pad = 0
        W, b = params['dilconv7']
        self.last = ops.Conv(W, b, pad=0, relu=False, layout='iohw', device=device,
                             dtype=dtype)                                # :102-105
        # the same two padded layers as 'valid' convolutions of zero-bordered buffers (a refinement loop keeps
        # those buffers, `new_session`): zero padding = a border that is written once and never again, and a
        # 'valid' layer between <= 16 channels runs on the vector-ALU kernel (csrc/conv_small.hip)
        self.conv1_valid = ops.Conv(params['conv1'][0], params['conv1'][1], pad=0, relu=True,
                                    device=device, dtype=dtype)
        W, b = params['dilconv1']
        self.dil1_valid = ops.Conv(W, b, pad=0, relu=True, dil=DILATIONS[0], layout='iohw', device=device,
                                   dtype=dtype)
        self._sessions = {}
        self._conv1_params = (params['conv1'][0], params['conv1'][1], device, dtype)
        self._hsplit = {}           # h channels -> (image-half conv, y-half conv) of the first layer

    def conv_layers(self):
        d = {'conv1': self.conv1, 'dilconv7': self.last}
        d.update({'dilconv%d' % (i + 1): c for i, c in enumerate(self.dil)})
        return d

    def new_session(self, h_list=None, y=None, tags=None):
        """State of one refinement loop (h fixed, y evolving): the ConcatLayer((h, y)) buffer of
        contextmod_dae.py:55-59 with h copied in once; each step only refreshes the y channels
        (plain device copies), so conv1 runs single-source."""
        if not h_list or y is None or len(h_list) != 1:
            return None
        h = h_list[0]
        B, ch, H, W = h.shape[0], h.shape[1], y.shape[2], y.shape[3]
        # The buffers are kept per geometry and handed out again (a captured refinement step points into them:
        # a new batch replays the same graph); what a call changes: the h channels (copied here) and the y
        # channels (every step).  The zero borders are written once.
        split = CTX_HSPLIT and y.dtype == torch.float32
        key = (B, ch, y.shape[1], H, W, y.dtype, str(y.device), split)
        sess = self._sessions.get(key)
        if sess is None:
            # [h, y] with the one-pixel zero border of conv1's pad, and conv1's output inside PadLayer(32)'s zeros;
            # split form: h and y in buffers of their own (both halves are dense single-source layers)
            zeros = lambda c, hh, ww: torch.zeros((B, c, hh, ww), dtype=y.dtype, device=y.device)
            pad32 = zeros(self.conv1.Cout, H + 64, W + 64)
            while len(self._sessions) >= 4:
                self._sessions.pop(next(iter(self._sessions)))
            if split:
                sess = {'hpad': zeros(ch, H + 2, W + 2), 'cat': zeros(y.shape[1], H + 2, W + 2), 'ch': 0,
                        'hb': torch.empty((B, self.conv1.Cout, H, W), dtype=y.dtype, device=y.device),
                        'pad32': pad32, 'split': self._split_convs(ch)}
            else:
                sess = {'cat': zeros(ch + y.shape[1], H + 2, W + 2), 'ch': ch, 'pad32': pad32, 'split': None}
            self._sessions[key] = sess
        if sess['split'] is not None:
            # the loop-invariant image half: bias + the h channels' taps, linear, once per batch
            sess['hpad'][:, :, 1:-1, 1:-1].copy_(h)
            sess['split'][0](sess['hpad'], out=sess['hb'])
        else:
            sess['cat'][:, :ch, 1:-1, 1:-1].copy_(h)
        sess['y_in_cat'] = False        # the y channels hold another loop's map
        return sess

    def _split_convs(self, ch):
        pair = self._hsplit.get(ch)
        if pair is None:
            W, b, device, dtype = self._conv1_params
            W = torch.as_tensor(W)
            pair = self._hsplit[ch] = (
                ops.Conv(W[:, :ch].contiguous(), b, pad=0, relu=False, device=device, dtype=dtype),
                ops.Conv(W[:, ch:].contiguous(), None, pad=0, relu=True, device=device, dtype=dtype))
        return pair

    def _first_layer(self, session):
        """conv1 of a session step into the PadLayer(32) buffer: the whole layer on [h, y], or its y half
        continuing from the cached image half."""
        if session['split'] is not None:
            session['split'][1](session['cat'], add=session['hb'], add_off=(0, 0), out=session['pad32'],
                                place=(32, 32))
        else:
            self.conv1_valid(session['cat'], out=session['pad32'], place=(32, 32))

    def scores(self, h_list, y, mask_override=None, session=None):
        if len(h_list) != 1:
            raise ValueError('expected 1 h tensor, got %d' % len(h_list))
        if session is not None:
            if not session.get('y_in_cat'):
                session['cat'][:, session['ch']:, 1:-1, 1:-1].copy_(y)
            session['y_in_cat'] = False      # (the caller's update changes y, not the buffer: see `fused_step`)
            self._first_layer(session)
            t = self.dil1_valid(session['pad32'])
            rest = self.dil[1:]
        else:
            t = self.conv1(h_list[0], x2=y)                          # h first (P13)
            rest = self.dil
        for conv in rest:
            t = conv(t)
        return self.last(t)

    def y_updated(self, session, y):
        """The caller has just updated y outside `fused_step` (the first step of a loop, which also hands out
        the score map): refresh the y channels of the concat buffer, so that every later step -- eager or
        replayed from a captured graph -- starts with the buffer equal to y."""
        session['cat'][:, session['ch']:, 1:-1, 1:-1].copy_(y)
        session['y_in_cat'] = True

    def fused_step(self, h_list, y, state, step, session):
        """One refinement step (scores + softmax + update of y, in place) with the last two layers and the
        update as one launch that also refreshes the y channels of the concat buffer (csrc/conv_small.hip
        ctx_tail_kernel): bit-identical y, three launches and the per-step copy of y fewer.  Returns the number
        of norm partials per image written (for ops.refine_finalize), or None when this geometry / dtype has no
        fused form (the caller then runs scores + refine_update)."""
        if session is None or not ops.ctx_tail_supported(self.dil[-1], self.last, y):
            return None
        if not session.get('y_in_cat'):
            session['cat'][:, session['ch']:, 1:-1, 1:-1].copy_(y)
        self._first_layer(session)
        t = self.dil1_valid(session['pad32'])
        for conv in self.dil[1:-1]:
            t = conv(t)
        nblk = ops.ctx_tail(self.dil[-1], self.last, t, y, state, step, ycat=session['cat'],
                            cat_c0=session['ch'], cat_off=(1, 1))
        session['y_in_cat'] = True
        return nblk

    def __call__(self, *args):
        score = self.scores(args[:-1], args[-1])
        return ops.crop_softmax(score, score.shape[2], score.shape[3], off=(0, 0))

    def residual(self, *args):
        score = self.scores(args[:-1], args[-1])
        return ops.crop_softmax(score, score.shape[2], score.shape[3], off=(0, 0),
                                minuend=args[-1])


def buildDAE_contextmod(input_concat_h_vars=None, input_mask_var=None, n_classes=11,
                        path_weights=None, model_name='dae_model.npz', trainable=False,
                        load_weights=False, out_nonlin='softmax', concat_h=('input',), noise=0.1,
                        params=None, device='cuda', dtype=torch.float32):
    """Mirror of models/contextmod_dae.py:19-23 (inference only: noise is the identity)."""
    if params is None:
        if not (load_weights and path_weights):
            raise ValueError('buildDAE_contextmod needs `params` or `path_weights`')
        params = load_param_list(os.path.join(path_weights, model_name), PARAM_ORDER)  # :127-132
    return ContextModDAE(params, n_classes, concat_h=concat_h, device=device, dtype=dtype)
