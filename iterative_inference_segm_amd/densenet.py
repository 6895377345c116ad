"""FC-DenseNet103 host segmentation network on the HIP kernels (mirror of reference
models/FCDenseNet.py:15-146,196-219; block internals restated from the un-vendored
FC_DenseNet.layers, see oracle/densenet.py -- unverifiable here).

MI355X mapping:
  * a dense block's stack is ONE preallocated (B, C_final, H, W) buffer; every BN_ReLU_Conv
    writes its `growth` new channels into the next channel slice (conv `out_c0`), so the
    ConcatLayer([stack, l]) of models/FCDenseNet.py:92 costs no copy;
  * BatchNorm uses batch statistics at inference (batch_norm_use_averages=False,
    iterative_inference.py:187; P10).  A channel's statistics never change once it is in the
    stack, so they are reduced once per produced slice (`bn_stats`), and each consumer applies its
    own gamma/beta with `bn_relu` (HBM-bound) before its convolution;
  * TransitionUp's 3x3 stride-2 Deconv2DLayer runs on the same static-tap conv kernel
    (IISEG_CONV_TRANSPOSED2: the stride is just a different tap-validity pattern), written
    center-cropped straight into the new stack; the skip stack is copied behind it.
Because BN couples the images of a batch, a reference batch must stay on one GPU.
"""
import os

import numpy as np
import torch

from . import ops

# mma='bf16c8': TransitionDown and the 1x1 class-score layer as one C8 kernel each (0: the 'bf16' forms on
# fp32 NCHW copies of the stack, through layout converters)
C8_1X1 = os.environ.get('IISEG_DENSENET_C8_1X1', '1') != '0'

# dense-block layers: BatchNorm fold and the statistics of the produced slice inside the conv launch (0: separate
# bn_fold / bn_stats_c8 launches)
M16_FUSE_BN = os.environ.get('IISEG_M16_FUSE_BN', '1') != '0'

# mma='bf16c8': the forward of one input geometry replayed from a captured HIP graph from its third call on (0: its
# ~430 launches issued from Python every time -- 7-9 ms of host time per forward against 6.5 ms of kernels)
FWD_GRAPH = os.environ.get('IISEG_DENSENET_GRAPH', '1') != '0'

GROWTH = 16
N_POOL = 5
LAYERS_PER_BLOCK = [4, 5, 7, 10, 12, 15, 12, 10, 7, 5, 4]      # FCDenseNet.py:208
N_FILTERS_FIRST = 48
BN_EPS = 1e-4


def layer_plan(n_layers_per_block=LAYERS_PER_BLOCK, n_pool=N_POOL, growth=GROWTH,
               n_first=N_FILTERS_FIRST, nb_in_channels=3, n_classes=11):
    """Creation-order (kind, cin, cout) of every parametrised layer ('first', 'brc', 'td', 'tu',
    'softmax'); 103 convolutions for the default configuration."""
    plan = [('first', nb_in_channels, n_first)]
    n = n_first
    skips = []
    for i in range(n_pool):
        for _ in range(n_layers_per_block[i]):
            plan.append(('brc', n, growth))
            n += growth
        skips.append(n)
        plan.append(('td', n, n))
    skips = skips[::-1]
    nblock = 0
    for _ in range(n_layers_per_block[n_pool]):
        plan.append(('brc', n, growth))
        n += growth
        nblock += 1
    for i in range(n_pool):
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


class _Stack:
    """A growing (B, cap, H, W) feature stack with per-channel batch statistics."""

    def __init__(self, B, cap, H, W, device, dtype):
        self.buf = torch.empty((B, cap, H, W), dtype=dtype, device=device)
        self.mean = torch.empty(cap, dtype=dtype, device=device)
        self.inv_std = torch.empty(cap, dtype=dtype, device=device)
        self.n = 0

    def added(self, k):
        ops.bn_stats(self.buf, self.n, k, self.mean, self.inv_std, BN_EPS)
        self.n += k

    def view(self):
        return self.buf if self.n == self.buf.shape[1] else self.buf[:, :self.n].contiguous()


class _Peek:
    """Iterator over the layer entries with a look at the next one."""

    def __init__(self, items):
        self.items, self.i = items, 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.i >= len(self.items):
            raise StopIteration
        self.i += 1
        return self.items[self.i - 1]

    def peek(self):
        return self.items[self.i] if self.i < len(self.items) else None


class _Level8:
    """The bf16 C8 buffer of ONE resolution of the C8 forward, (B, ctot / 8, H, W, 8), with its per-channel
    vectors (batch statistics, the folded BatchNorm of the current consumer).  It IS the stack of the up
    path's dense block at this resolution: [TransitionUp output | skip stack | the block's new layers]
    (models/FCDenseNet.py:119-127) -- and the down path's dense block at the same resolution builds its
    stack in place in the `skip` slice, so the ConcatLayer([deconv, skip]) of TransitionUp costs no copy
    (the block's channels, their statistics and the consumers' pointers are all slices of these tensors)."""

    def __init__(self, B, ctot, H, W, device):
        assert ctot % 16 == 0
        self.buf = torch.zeros((B, ctot // 8, H, W, 8), dtype=torch.bfloat16, device=device)
        self.mean = torch.zeros(ctot, dtype=torch.float32, device=device)
        self.inv_std = torch.zeros(ctot, dtype=torch.float32, device=device)
        self.a = torch.zeros(ctot, dtype=torch.float32, device=device)     # folded BN of the consumer
        self.b = torch.zeros(ctot, dtype=torch.float32, device=device)


class _Stack8:
    """A growing dense-block stack as bf16 C8 chunks -- mma='bf16c8': the format the dense-block layers read
    and write (csrc/conv_c8_m16.hip) --: channels [c0, c0 + cap) of a `_Level8`; the statistics (fp32) are
    those of the stored bf16 values, reduced once per produced slice."""

    def __init__(self, level, c0, cap):
        assert cap % 16 == 0 and c0 % 16 == 0
        self.buf = level.buf[:, c0 // 8:(c0 + cap) // 8]        # (a view: same image stride as the level)
        self.mean, self.inv_std = level.mean[c0:c0 + cap], level.inv_std[c0:c0 + cap]
        self.a, self.b = level.a[c0:c0 + cap], level.b[c0:c0 + cap]
        self.n = 0
        self.folded_for = None      # the layer entry (a, b) currently hold the folded BatchNorm of

    def added(self, k, stats=True):
        if stats:
            ops.bn_stats_c8(self.buf, self.n, k, self.mean, self.inv_std, BN_EPS)
        self.n += k


class FCDenseNet:
    def __init__(self, params, n_classes=11, layer=('pool4',), n_layers_per_block=LAYERS_PER_BLOCK,
                 n_pool=N_POOL, growth=GROWTH, device='cuda', dtype=torch.float32, mma=None):
        """mma: 'f32' (default) | 'bf16' (bf16 MFMA operands, fp32 NCHW activations) | 'bf16c8'
        (configs[2] as BASELINE names it: the dense-block stacks as bf16 C8 tensors, every
        BN_ReLU_Conv of a dense block one launch of the 16-row C8 kernel with BatchNorm + ReLU applied
        while the input is staged; the first conv and TransitionUp on the 64-channel C8 kernel;
        TransitionDown (BN + ReLU + 1x1 conv + pool) and the 1x1 score layer one launch each of
        csrc/conv1x1_c8.hip on the stack -- all 103 convolutions read bf16 C8)."""
        mma = mma or ops.DEFAULT_MMA
        self.c8 = mma == 'bf16c8' and dtype == torch.float32 and growth == 16
        if mma in ('bf16c8', 'bf16x3'):
            mma_other = 'bf16' if mma == 'bf16c8' else 'f32'
        else:
            mma_other = mma
        self.layer = list(layer)
        assert all(h in ['input', 'pool1', 'pool2', 'pool3', 'pool4', 'pool5'] for h in self.layer)
        self.nlpb, self.n_pool, self.growth = list(n_layers_per_block), n_pool, growth
        self.n_classes, self.device, self.dtype = n_classes, device, dtype
        dev = lambda a: torch.as_tensor(a).to(dtype).contiguous().to(device)
        self._c8_levels = {}          # mma='bf16c8': the forward's buffers per input geometry (_levels_c8)
        self._c8_graphs = {}          # ... and its captured launch sequence (_forward_c8_graph)
        self.layers = []
        for p in params:
            e = {'kind': p['kind']}
            if p['kind'] in ('brc', 'td'):
                e['beta'], e['gamma'] = dev(p['beta']), dev(p['gamma'])
            if p['kind'] == 'tu':
                e['conv'] = ops.Conv(p['W'], p['b'], pad=0, relu=False, layout='iohw',
                                     transposed=True, device=device, dtype=dtype, mma=mma_other)
                if self.c8 and p['W'].shape[0] % 16 == 0 and p['W'].shape[1] % 16 == 0 and \
                        tuple(p['W'].shape[2:]) == (3, 3):
                    # C8 form of the 3x3 stride-2 'valid' transposed convolution (Deconv2DLayer W[in,out,3,3],
                    # P3: out[c, 2 i + a] += x[o, i] W[o, c, 2 - a], oracle/nn.py deconv2d)  ==  a plain
                    # 'valid' 3x3 correlation of the zero-inserted map z[2 i + 2] = x[i] (size 2 H + 3) with
                    # the filter's in / out axes swapped: out[oy] = sum_k z[oy + k] W[k] (the two flips
                    # cancel).  Three of four products multiply a zero -- on a kernel that runs 15 x the
                    # fp32 static-tap kernel's rate on this shape.
                    Wf = np.ascontiguousarray(np.asarray(p['W']).transpose(1, 0, 2, 3))
                    e['conv8'] = ops.Conv(Wf, p['b'], pad=0, relu=False, device=device, dtype=dtype,
                                          mma='bf16c8')
            else:
                k = p['W'].shape[2]
                on_c8 = self.c8 and k == 3 and (p['kind'] == 'brc' or (p['kind'] == 'first' and
                                                                       p['W'].shape[0] % 16 == 0))
                e['conv'] = ops.Conv(p['W'], p['b'], pad=k // 2, relu=False, device=device,
                                     dtype=dtype, mma='bf16c8' if on_c8 else mma_other)
                if self.c8 and k == 1 and C8_1X1 and p['kind'] in ('td', 'softmax') and p['W'].shape[1] % 16 == 0:
                    # TransitionDown / the class-score layer on the C8 stack itself (csrc/conv1x1_c8.hip)
                    e['conv8'] = ops.Conv1x1C8(p['W'], p['b'], device=device)
            self.layers.append(e)

    def __call__(self, x):
        return self.forward(x)

    def _brc(self, it, stack, out=None, out_c0=None):
        e = next(it)
        if out is not None and out_c0 is not None:
            # dense-block layer (3x3, 16 filters): BatchNorm + ReLU applied while the conv stages
            # its input -- the normalised copy of the whole stack is never written
            r = e['conv'].bnrelu_conv(stack.buf, stack.n,
                                      (e['beta'], e['gamma'], stack.mean, stack.inv_std), out, out_c0)
            if r is not None:
                return r
        t = ops.bn_relu(stack.buf, stack.n, e['beta'], e['gamma'], stack.mean, stack.inv_std)
        return e['conv'](t, out=out, out_c0=out_c0)

    def _brc8(self, it, stack):
        """Dense-block layer on the C8 stack: BN + ReLU of the first n channels on the way in, 16 new
        channels into the next slice (models/FCDenseNet.py:88-92)."""
        e = next(it)
        if not M16_FUSE_BN:
            ops.bn_fold(e['beta'], e['gamma'], stack.mean, stack.inv_std, stack.n, a=stack.a, b=stack.b)
            e['conv'](stack.buf, in_c=stack.n, bn=(stack.a, stack.b), out=stack.buf, out_c0=stack.n)
            stack.added(self.growth)
            return
        # the statistics of the new slice come out of the conv's epilogue; the one-workgroup reduction that finishes
        # them also folds the NEXT consumer's BatchNorm (a dense-block layer or the TransitionDown of this stack)
        if stack.folded_for is not e:
            ops.bn_fold(e['beta'], e['gamma'], stack.mean, stack.inv_std, stack.n, a=stack.a, b=stack.b)
        nxt = it.peek()
        fold = None
        if nxt is not None and nxt['kind'] in ('brc', 'td') and stack.n + self.growth <= stack.a.numel():
            fold = (nxt['beta'], nxt['gamma'], stack.a, stack.b, stack.n + self.growth)
        e['conv'](stack.buf, in_c=stack.n, bn=(stack.a, stack.b), out=stack.buf, out_c0=stack.n,
                  stats=(stack.mean, stack.inv_std, BN_EPS), fold=fold)
        stack.added(self.growth, stats=False)
        stack.folded_for = nxt if fold is not None else None

    def _levels_c8(self, B, H, W, n_first):
        """The buffers of the C8 forward for one input geometry, kept across calls (every element a forward
        reads is written earlier in the same forward; the zero rows of padding are written once): one
        `_Level8` per resolution + the bottleneck's."""
        key = (B, H, W)
        lv = self._c8_levels.get(key)
        if lv is None:
            g, dev = self.growth, self.device
            n, hw, skip_n = n_first, (H, W), []
            for i in range(self.n_pool):
                n += g * self.nlpb[i]
                skip_n.append((n, hw))
                hw = (hw[0] // 2, hw[1] // 2)
            levels = []
            for L in range(self.n_pool):                 # up block i = n_pool - 1 - L runs at resolution L
                i = self.n_pool - 1 - L
                keep = g * self.nlpb[self.n_pool + i]
                ctot = keep + skip_n[L][0] + g * self.nlpb[self.n_pool + i + 1]
                levels.append((_Level8(B, ctot, skip_n[L][1][0], skip_n[L][1][1], dev), keep))
            bott = _Level8(B, n + g * self.nlpb[self.n_pool], hw[0], hw[1], dev)
            while len(self._c8_levels) >= 2:
                self._c8_levels.pop(next(iter(self._c8_levels)))
            lv = self._c8_levels[key] = (levels, bott)
        return lv

    def _forward_c8(self, x):
        B, _, H, W = x.shape
        g = self.growth
        it = _Peek(self.layers)
        hidden = [x] if 'input' in self.layer else []
        ints = [int(h[-1]) for h in self.layer if h != 'input']
        first = self.layers[0]['conv']
        n = first.Cout
        if not first.c8:
            raise NotImplementedError("mma='bf16c8': the first convolution's filter count must be a multiple of 16")
        levels, bott = self._levels_c8(B, H, W, n)
        # the down path's dense block of resolution L lives in the skip slice of that resolution's buffer
        stack = _Stack8(levels[0][0], levels[0][1], n + g * self.nlpb[0])
        next(it)['conv'](ops.nchw_to_c8(x), out=stack.buf, out_c0=0)  # first conv (linear)
        stack.added(n)
        skips = []
        for i in range(self.n_pool):                                  # FCDenseNet.py:81-100
            for _ in range(self.nlpb[i]):
                self._brc8(it, stack)
            skips.append(stack)
            # TransitionDown (BN -> ReLU -> 1x1 conv -> pool)
            e = next(it)
            n = stack.n
            if i + 1 < self.n_pool:
                nxt = _Stack8(levels[i + 1][0], levels[i + 1][1], n + g * self.nlpb[i + 1])
            else:
                nxt = _Stack8(bott, 0, n + g * self.nlpb[self.n_pool])
            if 'conv8' in e:
                # one kernel on the C8 stack: BN + ReLU on the way in, the 2x2 max-pool in the epilogue, bf16
                # C8 straight into the next block's stack
                if stack.folded_for is not e:
                    ops.bn_fold(e['beta'], e['gamma'], stack.mean, stack.inv_std, n, a=stack.a, b=stack.b)
                e['conv8'](stack.buf, n, bn=(stack.a, stack.b), pool=True, out=nxt.buf, out_c0=0)
                t = None
            else:
                # on the fp32-NCHW forms through layout converters
                xs = ops.c8_slice_to_nchw(stack.buf, 0, n)
                t = ops.bn_relu(xs, n, e['beta'], e['gamma'], stack.mean, stack.inv_std, out=xs)
                t = ops.maxpool2x2(e['conv'](t))
                del xs
                ops.nchw_to_c8_slice(t, nxt.buf, 0)
            H, W = H // 2, W // 2
            stack = nxt
            stack.added(n)
            if i + 1 in ints:
                hidden.append(t if t is not None else ops.c8_slice_to_nchw(stack.buf, 0, n))
        skips = skips[::-1]
        nblock = self.nlpb[self.n_pool]
        block0 = stack.n
        for _ in range(nblock):                                       # bottleneck, :107-111
            self._brc8(it, stack)
        for i in range(self.n_pool):                                  # :116-127
            e = next(it)                                              # TransitionUp
            skip = skips[i]
            h_in, w_in = H, W
            uh, uw = e['conv'].out_hw(H, W)
            H, W = min(uh, skip.buf.shape[2]), min(uw, skip.buf.shape[3])
            if (H, W) != tuple(skip.buf.shape[2:4]):
                raise NotImplementedError('skip larger than the upsampled map')
            keep = e['conv'].Cout
            nlay = self.nlpb[self.n_pool + i + 1]
            level, lkeep = levels[self.n_pool - 1 - i]
            assert lkeep == keep and skip.n == level.buf.shape[1] * 8 - keep - g * nlay
            new = _Stack8(level, 0, keep + skip.n + g * nlay)
            blk = up = None
            if 'conv8' in e and block0 % 8 == 0:
                # concat(block_to_upsample) = chunk planes [block0 / 8, n / 8) of the stack, read in place and
                # zero-inserted by the kernel's own patch staging (IISEG_CONV_ZINS), straight into the new
                # stack's first slice, center-cropped to the skip's size
                e['conv8'](stack.buf[:, block0 // 8:stack.n // 8], zins=True,
                           window=((uh - H) // 2, (uw - W) // 2, H, W), out=new.buf, out_c0=0)
            else:
                blk = ops.c8_slice_to_nchw(stack.buf, block0, stack.n - block0)
                up = e['conv'](blk, window=((uh - H) // 2, (uw - W) // 2, H, W))
                ops.nchw_to_c8_slice(up, new.buf, 0)
            new.added(keep)
            # the skip stack is already behind it, with its statistics (same tensors)
            new.added(skip.n, stats=False)
            stack, block0 = new, new.n
            del blk, up
            for _ in range(nlay):
                self._brc8(it, stack)
        e = next(it)                                                  # SoftmaxLayer's 1x1 conv
        if 'conv8' in e:
            score = e['conv8'](stack.buf, stack.n)
        else:
            score = e['conv'](ops.c8_slice_to_nchw(stack.buf, 0, stack.n))
        probs = ops.crop_softmax(score, H, W, off=(0, 0))
        return hidden + [probs]

    def _forward_c8_graph(self, x):
        """The C8 forward from a captured HIP graph.  Everything in it is static per input geometry: the level
        buffers (`_levels_c8`), the launch sequence (the Python-side BatchNorm-fold bookkeeping starts from the
        same state every forward) and, under one engine's workspace tag, the scratch the launches point into.
        First call of a geometry: eager (weights get packed, scratch comes into being); second: capture; then one
        copy of x into the graph's input + one replay.  Same kernels in the same order: same bits.  The results
        are handed out as copies -- a later forward does not overwrite what an earlier one returned."""
        dev = x.device
        key = (tuple(x.shape), ops._WS_TAG[0])
        ctx = self._c8_graphs.get(key)
        if ctx is None:
            while len(self._c8_graphs) >= 2:
                self._c8_graphs.pop(next(iter(self._c8_graphs)))
            self._c8_graphs[key] = {'graph': None}
            return self._forward_c8(x)
        self._c8_graphs[key] = self._c8_graphs.pop(key)          # most recently used last
        if ctx['graph'] is not None and ctx['ws'] != ops.workspace_ptrs(dev):
            ctx['graph'] = None          # (scratch regrown elsewhere since the capture)
        if ctx['graph'] is None:
            xs = x.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = self._forward_c8(xs)
            B, _, H, W = x.shape
            ctx.update(graph=g, x=xs, outs=outs, ws=ops.workspace_ptrs(dev),
                       keep=(ops.workspace_refs(dev), self._c8_levels.get((B, H, W))))
        else:
            ctx['x'].copy_(x)
        ctx['graph'].replay()
        return [o.clone() for o in ctx['outs']]

    def forward(self, x):
        if self.c8:
            if FWD_GRAPH and x.is_cuda and ops.CONV_PROFILE is None and not torch.cuda.is_current_stream_capturing():
                return self._forward_c8_graph(x)
            return self._forward_c8(x)
        B, _, H, W = x.shape
        g, dt, dev = self.growth, self.dtype, self.device
        it = iter(self.layers)
        hidden = [x] if 'input' in self.layer else []
        ints = [int(h[-1]) for h in self.layer if h != 'input']
        n = self.layers[0]['conv'].Cout
        stack = _Stack(B, n + g * self.nlpb[0], H, W, dev, dt)
        next(it)['conv'](x, out=stack.buf, out_c0=0)                  # first conv (linear)
        stack.added(n)
        skips = []
        for i in range(self.n_pool):                                  # FCDenseNet.py:81-100
            for _ in range(self.nlpb[i]):
                self._brc(it, stack, out=stack.buf, out_c0=stack.n)
                stack.added(g)
            skips.append(stack)
            t = ops.maxpool2x2(self._brc(it, stack))                  # TransitionDown
            H, W = H // 2, W // 2
            n = stack.n
            stack = _Stack(B, n + g * self.nlpb[i + 1], H, W, dev, dt)
            stack.buf[:, :n].copy_(t)                                 # plumbing: place the pool
            stack.added(n)
            if i + 1 in ints:
                hidden.append(t)
        skips = skips[::-1]
        nblock = self.nlpb[self.n_pool]
        block0 = stack.n
        for _ in range(nblock):                                       # bottleneck, :107-111
            self._brc(it, stack, out=stack.buf, out_c0=stack.n)
            stack.added(g)
        for i in range(self.n_pool):                                  # :116-127
            e = next(it)                                              # TransitionUp
            blk = stack.buf[:, block0:stack.n].contiguous()           # concat(block_to_upsample)
            skip = skips[i]
            uh, uw = e['conv'].out_hw(H, W)
            H, W = min(uh, skip.buf.shape[2]), min(uw, skip.buf.shape[3])
            if (H, W) != tuple(skip.buf.shape[2:]):
                raise NotImplementedError('skip larger than the upsampled map')
            keep = e['conv'].Cout
            nlay = self.nlpb[self.n_pool + i + 1]
            new = _Stack(B, keep + skip.n + g * nlay, H, W, dev, dt)
            e['conv'](blk, window=((uh - H) // 2, (uw - W) // 2, H, W), out=new.buf, out_c0=0)
            new.buf[:, keep:keep + skip.n].copy_(skip.buf[:, :skip.n])
            new.added(keep + skip.n)
            stack, block0 = new, new.n
            for _ in range(nlay):
                self._brc(it, stack, out=stack.buf, out_c0=stack.n)
                stack.added(g)
        score = next(it)['conv'](stack.view())                        # SoftmaxLayer's 1x1 conv
        probs = ops.crop_softmax(score, H, W, off=(0, 0))
        return hidden + [probs]


def build_fcdensenet(input_var=None, layer=('pool4',), nb_in_channels=3, n_classes=11,
                     output_d='4d', from_gt=False, weight_path=None, params=None, device='cuda',
                     dtype=torch.float32):
    """Mirror of models/FCDenseNet.py:196-219 (DenseNet103: 48 first filters, 5 pools, growth 16,
    blocks [4,5,7,10,12,15,12,10,7,5,4]).  Weights: `params` (list in creation order) or an
    `arr_%d` .npz (BN: beta, gamma, mean, inv_std; conv: W, b -- P10/P14)."""
    if params is None:
        if not weight_path:
            raise ValueError('build_fcdensenet needs `params` or `weight_path`')
        params = load_params(weight_path, layer_plan(nb_in_channels=nb_in_channels,
                                                     n_classes=n_classes))
    return FCDenseNet(params, n_classes, layer=layer, device=device, dtype=dtype)


def load_params(path, plan):
    with np.load(path) as f:
        vals = [f['arr_%d' % i] for i in range(len(f.files))]
    out, i = [], 0
    for kind, _, _ in plan:
        p = {'kind': kind}
        if kind in ('brc', 'td'):
            p['beta'], p['gamma'] = vals[i], vals[i + 1]      # mean, inv_std (i+2, i+3) unused:
            i += 4                                            # batch statistics at inference
        p['W'], p['b'] = vals[i], vals[i + 1]
        i += 2
        out.append(p)
    if i != len(vals):
        raise ValueError('%s holds %d arrays, the plan consumes %d' % (path, len(vals), i))
    return out
