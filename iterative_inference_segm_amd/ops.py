"""Operator layer: torch tensors in, HIP kernels of libiiseg_hip.so out.

torch is plumbing here (device memory + the current HIP stream); every arithmetic op is a
call through the C ABI (include/iiseg.h).  All ops enqueue on torch's current stream and
never synchronise.  Two arithmetic modes: float32 (throughput path) and float64 (strict-parity
path: the reference's CPU numerics), selected by the tensors' dtype.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, DeconvDesc, CONV_RELU, CONV_UNPOOL, CONV_TRANSPOSED2, CONV_X3, CONV_ZINS, check

# When set to a list, every conv launch appends (kernel, executed_flops, start_event, end_event):
# HIP events recorded on the launch stream right around the kernel (bench.py's roofline leg).
CONV_PROFILE = None
# With CONV_PROFILE: when set to a list, every C8 launch also appends its geometry (one dict per launch,
# same order as the 'conv_c8_kernel' entries of CONV_PROFILE) -- scripts/c8_step_profile.py
CONV_PROFILE_INFO = None
# ... and for kernels that are bound by HBM traffic rather than the matrix pipe: algorithmic bytes (input planes
# read once + output planes written once) summed per kernel name while CONV_PROFILE is on
KERNEL_BYTES = {}

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64'}

# float32 3x3 layers with at least this many input / output channels (and Cin % 16 == 0) run in
# Winograd F(2x2,3x3) form (conv_wino.hip); IISEG_WINO_MIN_CIN=0 switches the path off.  Below
# these widths the HBM-bound transforms cost more than the saved MFMAs (scripts/bench_wino.py).
WINO_MIN_CIN = int(os.environ.get('IISEG_WINO_MIN_CIN', '128'))
# float64 (strict-parity) path: Winograd F(2x2,3x3) on v_mfma_f64_16x16x4_f64 from this many input /
# output channels (conv_wino_f64.hip); IISEG_WINO_F64=0 keeps every layer on the direct kernel
WINO_F64 = os.environ.get('IISEG_WINO_F64', '1') != '0'
WINO_F64_MIN_CIN = int(os.environ.get('IISEG_WINO_F64_MIN_CIN', '128'))
# (a 128 -> 64 layer at 113^2 is HBM-bound on the V / M round trips: 1.35 ms against 1.1 ms on the halo kernel)
WINO_F64_MIN_COUT = int(os.environ.get('IISEG_WINO_F64_MIN_COUT', '128'))
WINO_MIN_COUT = int(os.environ.get('IISEG_WINO_MIN_COUT', '128'))
# Layers with at most this many input channels use the kernel that also applies the output
# transform (products stay in registers, no M round trip through HBM); deeper layers are
# MFMA-bound and run faster on the plain 256x128 GEMM + separate output transform.  0: never.
WINO_FUSED_MAX_CIN = int(os.environ.get('IISEG_WINO_FUSED_MAX_CIN', '256'))
# fuse the 2x2 max-pool behind a halo-kernel conv into that conv's epilogue
POOL_FUSE = os.environ.get('IISEG_POOL_FUSE', '1') != '0'
# float64 path: the same in the halo-tile float64 kernel's epilogue (conv_halo_f64.hip)
F64_POOL_FUSE = os.environ.get('IISEG_F64_POOL_FUSE', '1') != '0'
# ... and DePool2D masks as bytes between its encoder / decoder layers (the float64 pre-pool map is never stored)
F64_MASKS = os.environ.get('IISEG_F64_MASKS', '1') != '0'
# BN_ReLU_Conv of FC-DenseNet's dense blocks as one kernel.  Off by default: measured 6 % SLOWER
# end to end on config 3 (361 vs 383 images/s) -- the per-element parameter loads in the conv's
# staging cost more than the separate HBM-bound normalisation pass saves.
BNRELU_FUSE = os.environ.get('IISEG_BNRELU_FUSE', '0') != '0'
# 'valid' KxK layers without a static-tap variant (7x7 fc6) as im2col + split-K GEMM
CONV_GEMM = os.environ.get('IISEG_CONV_GEMM', '1') != '0'
# Matrix-pipe operand precision of the float32 path's wide 3x3 layers: 'f32' (exact fp32 MFMA, the
# default: the path with tolerance claims) or 'bf16' (bf16 operands, fp32 accumulation -- statistical
# parity only; BASELINE north_star's 16-bit MFMA target).  Per-Conv `mma=` overrides the default.
DEFAULT_MMA = os.environ.get('IISEG_MMA', 'f32')
# bf16 mode: 3x3 layers at least this wide (input AND output channels) are candidates for the bf16
# Winograd kernels (conv_wino_bf16.hip); below, the direct bf16 kernel won every measured case
#  and also fuses the pool and the DePool2D byte masks
BF16_WINO_MIN_CIN = int(os.environ.get('IISEG_BF16_WINO_MIN_CIN', '256'))
# Both 16-bit forms possible for a layer (Winograd / direct): the choice is DETERMINISTIC -- an optional
# table of measured winners per launch geometry (bf16_picks.json beside this file, keys without the
# batch size; NONE is shipped: the static rule decides everywhere) and, for a geometry the table does
# not hold, a static rule on the geometry alone (`_bf16_static_pick`).  Two processes, two ranks, two runs make the same choices,
# so bf16 outputs are reproducible and identical across the ranks of a data-parallel evaluation, and a
# loop-invariant border and its recomputed window come from the same form.  IISEG_BF16_TUNE=1 (the
# generating script only) times both forms for keys the table lacks and records them in BF16_PICKS.
BF16_TUNE = os.environ.get('IISEG_BF16_TUNE', '0') == '1'
# 'wino' / 'halo': force one form wherever both apply (tests, A/B timing); '' = table, then rule
BF16_FORCE = os.environ.get('IISEG_BF16_FORM', '')
BF16_PICKS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bf16_picks.json')


def _bf16_key(Cin, Cout, C1, C2, unpool, OH, OW):
    return '%d,%d,%d,%d,%d,%d,%d' % (Cin, Cout, C1, C2, 1 if unpool else 0, OH, OW)


def _load_bf16_picks():
    import json
    try:
        with open(BF16_PICKS_FILE) as f:
            return {k: v for k, v in json.load(f).get('picks', {}).items() if v in ('wino', 'halo')}
    except OSError:
        return {}


BF16_PICKS = _load_bf16_picks()
BF16_TIMES = {}      # key -> [ms Winograd, ms direct] of the keys timed in this process (tuning runs)


def _bf16_static_pick(Cin, OW):
    """Rule for geometries outside the table, from the round-2 measurements: the
    direct kernel's 32-pixel column tiles waste (1 - OW / (32 ceil(OW / 32))) of its MFMAs, and from
    1024 input channels the long-K Winograd GEMM wins on any window."""
    fill = OW / (32.0 * ((OW + 31) // 32))
    return 'wino' if (Cin >= 1024 or fill < 0.7) else 'halo'


# C8 layers with at most 16 output channels on the 16-row MFMA kernel (conv_c8_m16.hip); 0: the
# 32-row form of conv_c8_bf16.hip (timing experiments)
C8_M16 = os.environ.get('IISEG_C8_M16', '1') != '0'
BF16_UPCONV1 = os.environ.get('IISEG_BF16_UPCONV1', '0') != '0'
BF16_WINO_MIN_COUT = int(os.environ.get('IISEG_BF16_WINO_MIN_COUT', '256'))
_wino_ws = {}   # device -> workspace tensor shared by all layers (launches are stream-ordered)
_wino_ws64 = {}
_bn_ws = {}     # device -> BN partial-sum workspace (stream-ordered reuse)


# Scratch is reused by consecutive launches of ONE stream (stream-ordered).  Engines that run
# concurrently on different streams (bench.py --streams 2) must not share it: every
# IterativeInference tags its launches (`workspace_tag`), and the caches are keyed (device, tag).
_WS_TAG = [0]


class workspace_tag:
    """Context manager: launches inside use the scratch buffers of `tag` (an engine's id)."""

    def __init__(self, tag):
        self.tag, self.prev = tag, None

    def __enter__(self):
        self.prev, _WS_TAG[0] = _WS_TAG[0], self.tag
        return self

    def __exit__(self, *exc):
        _WS_TAG[0] = self.prev
        return False


def _ws_key(device):
    return (device, _WS_TAG[0])


def _wino_workspace(n, device):
    key = _ws_key(device)
    ws = _wino_ws.get(key)
    if ws is None or ws.numel() < n:
        _wino_ws[key] = None
        ws = _wino_ws[key] = torch.empty(int(n), dtype=torch.float32, device=device)
    return ws


def _wino_workspace64(n, device):
    """float64 scratch of the current workspace tag (Winograd V / M, the GEMM form's V / partial sums)."""
    key = _ws_key(device)
    ws = _wino_ws64.get(key)
    if ws is None or ws.numel() < n:
        _wino_ws64[key] = None
        ws = _wino_ws64[key] = torch.empty(int(n), dtype=torch.float64, device=device)
    return ws


_m16_ws = {}


def _m16_workspace(nbytes, device):
    """Scratch of the split-K launches of the 16-row C8 kernel (fp32 slabs of partial sums)."""
    key = _ws_key(device)
    ws = _m16_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        _m16_ws[key] = None
        ws = _m16_ws[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return ws


def workspace_refs(device):
    """The scratch tensors launches under the current workspace tag point into (fp32 / float64
    Winograd workspaces, BN partial sums): a captured HIP graph keeps them alive."""
    key = _ws_key(device)
    return tuple(c.get(key) for c in (_wino_ws, _wino_ws64, _bn_ws, _m16_ws))


def workspace_ptrs(device):
    """Their device addresses (None where not allocated): part of what a captured graph is valid for."""
    return tuple(None if t is None else t.data_ptr() for t in workspace_refs(device))


# Dispatch timing (bench.py's roofline pass, `profile_begin` / `profile_end`): the two "events" around a
# conv call are positions in the library's launch sequence, and the time between them is the sum of the
# execution times of the kernels launched in between -- each measured by the start / stop HIP events its
# own dispatch carries (include/iiseg.h iiseg_profile_begin).  Off: torch events recorded on the stream.
_DISPATCH_TIMING = [False]
PROFILE_MS = []


class _LaunchMark:
    def __init__(self, n):
        self.n = n

    def elapsed_time(self, other):
        return float(sum(PROFILE_MS[self.n:other.n]))


_PROFILE_CAP = [16384]


def profile_begin(capacity=16384):
    check(_lib.load().iiseg_profile_begin(int(capacity)), 'iiseg_profile_begin')
    _DISPATCH_TIMING[0] = True
    _PROFILE_CAP[0] = int(capacity)
    del PROFILE_MS[:]


def profile_end():
    """Ends the pass and fills PROFILE_MS (one entry per launch of the library, in launch order).  A pass
    that filled the event table raises: launches past the capacity carry no events, and a roofline built
    on such a pass would silently under-report kernel time."""
    _DISPATCH_TIMING[0] = False
    cap = _PROFILE_CAP[0]
    buf = (C.c_float * cap)()
    n = _lib.load().iiseg_profile_end(buf, cap)
    if n < 0:
        raise RuntimeError('iiseg_profile_end: %d' % n)
    if n >= cap:
        raise RuntimeError('profiling pass reached its capacity of %d launches: later launches were not '
                           'timed (raise profile_begin(capacity))' % cap)
    PROFILE_MS[:] = list(buf[:n])
    return n


def _ev():
    """A mark on the launch stream right now (bench.py's roofline leg): a HIP event recorded on it, or,
    under dispatch timing, the position in the library's launch sequence."""
    if _DISPATCH_TIMING[0]:
        return _LaunchMark(_lib.load().iiseg_profile_count())
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _stream():
    if not torch.cuda.is_available():
        raise RuntimeError('iiseg ops need device tensors on a HIP GPU; none is available and '
                           'there is no CPU fallback')
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('iiseg ops need device tensors (got %s)' % t.device)
    if t.dtype != dtype:
        raise RuntimeError('expected %s, got %s' % (dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('iiseg ops need C-contiguous NCHW tensors')
    return C.c_void_p(t.data_ptr())


def _fn(name, dtype):
    """C entry point `iiseg_<name>_<f32|f64>` for a tensor dtype."""
    if dtype not in _SUFFIX:
        raise RuntimeError('iiseg ops support float32 and float64, not %s' % dtype)
    return getattr(_lib.load(), 'iiseg_%s_%s' % (name, _SUFFIX[dtype]))


class Conv:
    """One convolution layer bound to its weights: packed-weight (+ gather-table) cache per
    input geometry.  Weight layouts: 'oihw' (Lasagne Conv2DLayer W[out,in,kh,kw], P1) or
    'iohw' (DilatedConv2DLayer W[in,out,kh,kw], P11)."""

    def __init__(self, W, b, pad, relu, dil=1, layout='oihw', device='cuda', dtype=torch.float32,
                 transposed=False, mma=None):
        """transposed=True: 3x3 stride-2 transposed convolution, crop='valid' (Deconv2DLayer
        W[in,out,3,3] -> layout 'iohw'); output (2H+1, 2W+1).  mma: 'f32' | 'bf16' operand
        precision on the matrix pipe (float32 tensors only; None: ops.DEFAULT_MMA)."""
        self.lib = _lib.load()
        self.mma = (mma or DEFAULT_MMA) if dtype == torch.float32 else 'f32'
        if self.mma not in ('f32', 'bf16', 'bf16c8', 'bf16x3'):
            raise ValueError("mma must be 'f32', 'bf16', 'bf16c8' or 'bf16x3'")
        # 'bf16c8': bf16 MFMA operands AND bf16 C8 activations between the 3x3 layers (conv_c8_bf16.hip);
        # a layer takes that form when it is handed a C8 tensor, any other call is the 'bf16' mode.
        # 'bf16x3': the fp32-class mode of the same kernel -- activations and weights as bf16 hi / lo
        # pairs, three products per term (IISEG_CONV_X3); any other call is the 'f32' mode
        self.x3 = self.mma == 'bf16x3'
        self.c8 = self.mma in ('bf16c8', 'bf16x3')
        if self.c8:
            self.mma = 'f32' if self.x3 else 'bf16'
        self.transposed = bool(transposed)
        self.dtype = dtype
        # (the conv kernels do not promise to propagate a NaN -- build.py EXTRA_FLAGS -- so non-finite
        # parameters are refused where they enter, once, when the layer is built)
        if not bool(torch.isfinite(torch.as_tensor(W)).all()) or \
                (b is not None and not bool(torch.isfinite(torch.as_tensor(b)).all())):
            raise ValueError('non-finite convolution parameters (NaN / Inf in W or b)')
        self.W = torch.as_tensor(W).to(dtype).contiguous().to(device)
        self.b = None if b is None else torch.as_tensor(b).to(dtype).contiguous().to(device)
        self.layout = layout
        if layout == 'oihw':
            self.Cout, self.Cin, self.KH, self.KW = self.W.shape
            self.so, self.sc = self.Cin * self.KH * self.KW, self.KH * self.KW
        elif layout == 'iohw':
            self.Cin, self.Cout, self.KH, self.KW = self.W.shape
            self.so, self.sc = self.KH * self.KW, self.Cout * self.KH * self.KW
        else:
            raise ValueError(layout)
        self.pad, self.dil, self.relu = int(pad), int(dil), bool(relu)
        taps = (self.KH, self.KW) in ((1, 1), (3, 3)) or \
            ((self.KH, self.KW) == (4, 4) and dtype == torch.float32)
        # which kernel family the C ABI dispatches this filter shape to
        if dtype == torch.float64:
            self.kernel = 'conv_taps_f64_kernel'
            # other filter shapes (7x7 fc6): im2col + the same weights as a 1x1 convolution
            self.via_im2col = not taps
            if self.via_im2col and (layout != 'oihw' or self.pad != 0 or self.dil != 1):
                raise NotImplementedError('float64 path: only valid, undilated oihw KxK filters '
                                          'go through im2col')
        else:
            self.kernel = 'conv_taps_f32_kernel' if taps else 'conv_igemm_f32_kernel'
            # direct 3x3 layers below 256 output channels run on the halo-tile kernel (the C side
            # dispatches: conv_igemm.hip, IISEG_CONV_HALO)
            # (dilated 3x3 layers with at most 16 output channels -- the context module -- too)
            if (self.KH, self.KW) == (3, 3) and not self.transposed and \
                    os.environ.get('IISEG_CONV_HALO', '1') != '0' and \
                    ((self.dil == 1 and self.Cout < 256) or
                     (self.dil in (2, 4, 8, 16) and self.Cout <= 16)):
                self.kernel = 'conv_halo_f32_kernel'
            self.via_im2col = False
        self.wino = (dtype == torch.float32 and (self.KH, self.KW) == (3, 3) and self.dil == 1 and
                     not self.transposed and self.Cin % 16 == 0 and
                     0 < WINO_MIN_CIN <= self.Cin and self.Cout >= WINO_MIN_COUT)
        self.wino_f64 = (dtype == torch.float64 and WINO_F64 and (self.KH, self.KW) == (3, 3) and
                         self.dil == 1 and not self.transposed and self.Cin % 16 == 0 and
                         self.Cin >= WINO_F64_MIN_CIN and self.Cout >= WINO_F64_MIN_COUT)
        self.wino_bf16 = (self.mma == 'bf16' and (self.KH, self.KW) == (3, 3) and self.dil == 1 and
                          not self.transposed and self.Cin >= BF16_WINO_MIN_CIN and
                          self.Cout >= BF16_WINO_MIN_COUT)
        # bf16 mode: every other plain 3x3 layer on the bf16 halo kernel (conv_halo_bf16.hip)
        self.halo_bf16 = (self.mma == 'bf16' and (self.KH, self.KW) == (3, 3) and self.dil == 1 and
                          not self.transposed and not self.wino_bf16)
        self.c8 = self.c8 and (self.KH, self.KW) == (3, 3) and self.dil == 1 and not self.transposed
        self._U = None
        self._U16 = None
        self._W16 = None
        self._W16c8 = None
        self._form_cache = {}
        self._plans = {}
        self._packs = {}

    def out_hw(self, H, W):
        if self.transposed:
            return (H - 1) * 2 + self.KH, (W - 1) * 2 + self.KW
        return (H + 2 * self.pad - self.dil * (self.KH - 1),
                W + 2 * self.pad - self.dil * (self.KW - 1))

    def flops(self, B, OH, OW):
        """Nominal 2*Cin*Cout*k*k*OH*OW*B (SURVEY 6.2 convention)."""
        return 2.0 * self.Cin * self.Cout * self.KH * self.KW * OH * OW * B

    def _plan(self, B, C1, C2, H, W, window, add_geom, unpool, out_slice=None, place=None,
              anchor=(0, 0)):
        key = (B, C1, C2, H, W, window, add_geom, unpool, out_slice, place, anchor)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        if C1 + C2 != self.Cin:
            raise RuntimeError('conv expects %d input channels, got %d+%d' % (self.Cin, C1, C2))
        fullH, fullW = self.out_hw(H, W)
        oy0, ox0, OH, OW = window if window is not None else (0, 0, fullH, fullW)
        d = ConvDesc()
        d.B, d.C1, d.C2, d.H, d.W = B, C1, C2, H, W
        d.Cout, d.KH, d.KW, d.pad, d.dil = self.Cout, self.KH, self.KW, self.pad, self.dil
        d.oy0, d.ox0, d.OH, d.OW = oy0, ox0, OH, OW
        if add_geom is not None:
            d.AH, d.AW, d.ay0, d.ax0 = add_geom
        d.flags = (CONV_RELU if self.relu else 0) | (CONV_UNPOOL if unpool else 0) | \
            (CONV_TRANSPOSED2 if self.transposed else 0)
        if out_slice is not None:
            d.out_ctot, d.out_c0 = out_slice
        if place is not None:
            d.out_H, d.out_W, d.out_y0, d.out_x0 = place
        d.tile_y0, d.tile_x0 = int(anchor[0]) & 1, int(anchor[1]) & 1
        so, sc = self.so, self.sc
        if self.via_im2col:
            # logical layer: 1x1 over C*KH*KW channels of the im2col'd tensor (OH x OW)
            d.C1, d.C2, d.H, d.W = C1 * self.KH * self.KW, 0, fullH, fullW
            d.KH = d.KW = 1
            so, sc = self.Cin * self.KH * self.KW, 1
        # the packed weights (and the gather table of the table-driven kernel) depend on the input
        # geometry only: one copy per (C1, C2, H, W), shared by every window / placement variant
        pkey = (d.C1, d.C2, d.H, d.W)
        packed = self._packs.get(pkey)
        if self.dtype == torch.float64:
            check(self.lib.iiseg_conv_plan_f64(C.byref(d)), 'iiseg_conv_plan_f64')
            if packed is None:
                wp = torch.empty(d.Kpad * d.Mpad, dtype=self.dtype, device=self.W.device)
                check(self.lib.iiseg_conv_pack_f64(_stream(), C.byref(d), _ptr(self.W, self.dtype),
                                                   so, sc, _ptr(wp, self.dtype)),
                      'iiseg_conv_pack_f64')
                packed = self._packs[pkey] = (wp, None)
        else:
            check(self.lib.iiseg_conv_plan(C.byref(d)), 'iiseg_conv_plan')
            if packed is None:
                wp = torch.empty(d.Kpad * d.Mpad, dtype=self.dtype, device=self.W.device)
                ktab = torch.empty(d.Kpad * 4, dtype=torch.int32, device=self.W.device)
                check(self.lib.iiseg_conv_pack_f32(_stream(), C.byref(d), _ptr(self.W), so, sc,
                                                   _ptr(wp), _ptr(ktab, torch.int32)),
                      'iiseg_conv_pack_f32')
                packed = self._packs[pkey] = (wp, ktab)
        wp, ktab = packed
        plan = (d, wp, ktab)
        self._plans[key] = plan
        return plan

    def __call__(self, x1, x2=None, pre=None, pooled=None, add=None, add_off=(0, 0),
                 window=None, out=None, out_c0=None, place=None, anchor=(0, 0), pool_out=None,
                 mask_in=None, unpool_hw=None, mask_out=None, store_out=True, out_format=None,
                 in_c=None, bn=None, stats=None, fold=None, zins=False):
        """x1 (B,C1,H,W) [+ x2 (B,C2,H,W): channel concat, x1 first].  With `pre`/`pooled`
        the logical input is the equality-mask unpool of x1 (DePool2D) at pre's size.
        `add` (B,Cout,AH,AW) is summed into the result starting at `add_off`;
        `window` = (oy0, ox0, OH, OW) restricts the computed output region.  With `out_c0`
        the result is written into channels [out_c0, out_c0+Cout) of the wider tensor `out`; with
        `place` = (y0, x0) the (OH, OW) result is written at that offset of the larger planes of
        `out` (everything else in `out` is left untouched).  `anchor` = parity of the output
        row / column where the Winograd 2x2 tiles start (a per-layer constant for the caller:
        launches of one layer agree bit for bit only under the same anchor).  `pool_out`: the FULL
        pooled tensor (B, Cout, fullH//2, fullW//2) of this layer; the 2x2 max-pool of the computed
        window is written into it by the conv's epilogue (see `pool_window`).

        DePool2D masks as bytes (`mask_ok()` layers, include/iiseg.h iiseg_conv_mask_f32):
        `mask_out` (uint8, pool_out's shape) receives, with the pool, one byte per pooled element
        whose bit (y&1)*2+(x&1) says pre == pooled; `store_out=False` then skips the pre-pool map
        itself (returns None; `out`, if given, only carries the placement geometry).  `mask_in`
        (uint8, x1's shape) + `unpool_hw` = (H, W) replace `pre` / `pooled` for the unpooled
        input."""
        if is_c8(x1):
            return self._call_c8(x1, x2, add, add_off, window, out, place, pool_out, mask_in,
                                 unpool_hw, mask_out, store_out, out_format, out_c0, in_c, bn, stats, fold,
                                 zins)
        if zins or in_c is not None or bn is not None or stats is not None or fold is not None:
            raise RuntimeError('in_c / bn / stats: C8 input only (layers with at most 16 output channels)')
        dt = self.dtype
        unpool = pre is not None or mask_in is not None
        masked = mask_in is not None or mask_out is not None
        B, C1 = x1.shape[0], x1.shape[1]
        if masked:
            if not self.mask_ok(False):
                raise RuntimeError('DePool2D byte masks need a halo-kernel layer (Conv.mask_ok)')
            if mask_in is not None and (pre is not None or mask_in.dtype != torch.uint8 or
                                        mask_in.shape != x1.shape or unpool_hw is None or
                                        (unpool_hw[0] // 2, unpool_hw[1] // 2) != tuple(x1.shape[2:])):
                raise RuntimeError('mask_in: uint8 of x1\'s shape %s with unpool_hw, without pre'
                                   % (tuple(x1.shape),))
            if mask_out is not None and (pool_out is None or mask_out.dtype != torch.uint8 or
                                         mask_out.shape != pool_out.shape):
                raise RuntimeError('mask_out: uint8 of pool_out\'s shape')
        if not store_out and (mask_out is None or out_c0 is not None):
            raise RuntimeError('store_out=False needs pool_out and mask_out')
        if mask_in is not None:
            H, W = int(unpool_hw[0]), int(unpool_hw[1])
        elif unpool:
            H, W = pre.shape[2], pre.shape[3]
            if pooled.shape != x1.shape or pre.shape[:2] != x1.shape[:2] or \
                    (H // 2, W // 2) != tuple(x1.shape[2:]):
                raise RuntimeError('unpool shapes: up %s pre %s pooled %s'
                                   % (tuple(x1.shape), tuple(pre.shape), tuple(pooled.shape)))
        else:
            H, W = x1.shape[2], x1.shape[3]
        C2 = 0
        if x2 is not None:
            if x2.shape[0] != B or tuple(x2.shape[2:]) != (H, W):
                raise RuntimeError('concat shapes %s vs %s' % (tuple(x1.shape), tuple(x2.shape)))
            C2 = x2.shape[1]
        add_geom = None
        if add is not None:
            add_geom = (add.shape[2], add.shape[3], add_off[0], add_off[1])
            if add.shape[0] != B or add.shape[1] != self.Cout:
                raise RuntimeError('add tensor shape %s' % (tuple(add.shape),))
        if self.via_im2col:
            if x2 is not None or unpool or window is not None:
                raise NotImplementedError('float64 im2col path: plain single-source conv only')
            fullH, fullW = self.out_hw(H, W)
            cols = torch.empty((B, C1 * self.KH * self.KW, fullH, fullW), dtype=dt, device=x1.device)
            check(self.lib.iiseg_im2col_f64(_stream(), _ptr(x1, dt), _ptr(cols, dt), B, C1, H, W,
                                            self.KH, self.KW), 'iiseg_im2col_f64')
            d, wp, ktab = self._plan(B, C1, 0, H, W, None, add_geom, False)
            x1 = cols
        else:
            out_slice = None if out_c0 is None else (out.shape[1], int(out_c0))
            pl = None if place is None else (out.shape[2], out.shape[3], int(place[0]), int(place[1]))
            d, wp, ktab = self._plan(B, C1, C2, H, W, window, add_geom, unpool, out_slice, pl,
                                     (int(anchor[0]) & 1, int(anchor[1]) & 1))
        if not store_out:
            out = None                     # geometry taken above; nothing is written
        elif place is not None:
            if self.via_im2col or out is None or out.shape[0] != B or out.dtype != dt or \
                    (out_c0 is None and out.shape[1] != self.Cout):
                raise RuntimeError('bad placement target %s' % (None if out is None else tuple(out.shape),))
        elif out_c0 is not None:
            if self.via_im2col or tuple(out.shape[2:]) != (d.OH, d.OW) or out.shape[0] != B or \
                    out.dtype != dt or out_c0 + self.Cout > out.shape[1]:
                raise RuntimeError('bad output slice %s @%d' % (tuple(out.shape), out_c0))
        elif out is None:
            out = torch.empty((B, self.Cout, d.OH, d.OW), dtype=dt, device=x1.device)
        elif tuple(out.shape) != (B, self.Cout, d.OH, d.OW) or out.dtype != dt:
            raise RuntimeError('out shape %s != %s' % (tuple(out.shape), (B, self.Cout, d.OH, d.OW)))
        # profiling: every launch is bracketed by events recorded IMMEDIATELY around the ctypes
        # call (after all planning / workspace work), so a bracket holds the kernel and nothing else
        prof = CONV_PROFILE
        wino16 = not masked and pool_out is None and self.wino_bf16 and \
            bool(self.lib.iiseg_conv_wino_bf16_supported(C.byref(d)))
        if wino16 and self.lib.iiseg_conv_halo_bf16_supported(C.byref(d)):
            # both 16-bit forms can run this layer: table of measured winners, else the static rule
            # (see BF16_PICKS above).  The key carries no batch size: an image gets the same form
            # alone and in a batch.
            key = _bf16_key(self.Cin, self.Cout, C1, C2, unpool, d.OH, d.OW)
            pick = BF16_FORCE or BF16_PICKS.get(key)
            if not pick and BF16_TUNE and not torch.cuda.is_current_stream_capturing():
                forms = (lambda: self._call_wino_bf16(d, x1, x2, pre, pooled, add, out, None),
                         lambda: self._call_halo_bf16(d, x1, x2, pre, pooled, add, out, None, None,
                                                      None, None, B, H, W))
                ms = []
                for f in forms:
                    f()                                       # packs the weights, warms up
                    e0 = _ev(); f(); f(); f(); e1 = _ev()
                    e1.synchronize()
                    ms.append(e0.elapsed_time(e1))
                pick = BF16_PICKS[key] = 'wino' if ms[0] <= ms[1] else 'halo'
                BF16_TIMES[key] = ms
            if not pick:
                pick = _bf16_static_pick(self.Cin, d.OW)
            if pick == 'halo':
                return self._call_halo_bf16(d, x1, x2, pre, pooled, add, out, None, None, None, prof,
                                            B, H, W)
        if wino16:
            return self._call_wino_bf16(d, x1, x2, pre, pooled, add, out, prof)
        # (the DAE's last layer -- DePool2D input, <= 16 output channels: reading pre / pooled it is
        # read-bound and faster on the 16-row fp32 kernel, 0.60 vs 0.74 ms at configs[1]; from mask
        # bytes the bf16 kernel wins, 0.45 vs 0.58 ms)
        if self.halo_bf16 and \
                not (unpool and self.Cout <= 16 and mask_in is None and not BF16_UPCONV1) and \
                self.lib.iiseg_conv_halo_bf16_supported(C.byref(d)):
            return self._call_halo_bf16(d, x1, x2, pre, pooled, add, out, pool_out, mask_in,
                                        mask_out, prof, B, H, W)
        if not masked and pool_out is None and self.wino and \
                self.lib.iiseg_conv_wino_supported(C.byref(d)) and \
                self._form_by_full_map(self.lib.iiseg_conv_wino_supported, d):
            return self._call_wino(d, x1, x2, pre, pooled, add, out, prof)
        if masked and dt == torch.float64:
            if pool_out is not None:
                fh, fw = self.out_hw(H, W)
                if tuple(pool_out.shape) != (B, self.Cout, fh // 2, fw // 2) or pool_out.dtype != dt:
                    raise RuntimeError('pool_out shape %s' % (tuple(pool_out.shape),))
            ev0 = _ev() if prof is not None else None
            check(self.lib.iiseg_conv_mask_f64(_stream(), C.byref(d), _ptr(x1, dt), _ptr(x2, dt), _ptr(pre, dt),
                                               _ptr(pooled, dt), _ptr(mask_in, torch.uint8), _ptr(wp, dt),
                                               _ptr(self.b, dt), _ptr(add, dt), _ptr(out, dt), _ptr(pool_out, dt),
                                               _ptr(mask_out, torch.uint8)), 'iiseg_conv_mask_f64')
            if prof is not None:
                prof.append(('conv_halo_f64_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        if masked:
            if pool_out is not None:
                fh, fw = self.out_hw(H, W)
                if tuple(pool_out.shape) != (B, self.Cout, fh // 2, fw // 2):
                    raise RuntimeError('pool_out shape %s' % (tuple(pool_out.shape),))
            ev0 = _ev() if prof is not None else None
            check(self.lib.iiseg_conv_mask_f32(_stream(), C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre),
                                               _ptr(pooled), _ptr(mask_in, torch.uint8), _ptr(wp),
                                               _ptr(ktab, torch.int32), _ptr(self.b), _ptr(add),
                                               _ptr(out), _ptr(pool_out),
                                               _ptr(mask_out, torch.uint8)), 'iiseg_conv_mask_f32')
            if prof is not None:
                prof.append((self.kernel, self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        if pool_out is not None and dt == torch.float64:
            fh, fw = self.out_hw(H, W)
            if tuple(pool_out.shape) != (B, self.Cout, fh // 2, fw // 2) or pool_out.dtype != dt or \
                    not self.lib.iiseg_conv_pool_f64_supported(C.byref(d)) or add is not None:
                raise RuntimeError('conv + pool fusion is not available for this launch')
            ev0 = _ev() if prof is not None else None
            check(self.lib.iiseg_conv_pool_f64(_stream(), C.byref(d), _ptr(x1, dt), _ptr(x2, dt), _ptr(pre, dt),
                                               _ptr(pooled, dt), _ptr(wp, dt), _ptr(self.b, dt), None,
                                               _ptr(out, dt), _ptr(pool_out, dt)), 'iiseg_conv_pool_f64')
            if prof is not None:
                prof.append(('conv_halo_f64_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        if pool_out is not None:
            fh, fw = self.out_hw(H, W)
            if dt != torch.float32 or tuple(pool_out.shape) != (B, self.Cout, fh // 2, fw // 2) or \
                    not self.lib.iiseg_conv_pool_supported(C.byref(d)):
                raise RuntimeError('conv + pool fusion is not available for this launch')
            ev0 = _ev() if prof is not None else None
            check(self.lib.iiseg_conv_pool_f32(_stream(), C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre),
                                               _ptr(pooled), _ptr(wp), _ptr(ktab, torch.int32),
                                               _ptr(self.b), _ptr(add), _ptr(out), _ptr(pool_out)),
                  'iiseg_conv_pool_f32')
            if prof is not None:
                prof.append((self.kernel, self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        gemm_shape = self.kernel == 'conv_igemm_f32_kernel' or (
            # deep 1x1 layers on few pixels (fc7, score_fr): the same split-K GEMM beats the 1x1 tap
            # kernel
            dt == torch.float32 and (self.KH, self.KW) == (1, 1) and self.Cin >= 1024)
        if gemm_shape and CONV_GEMM and self.mma == 'bf16' and x2 is None and not unpool and \
                add is None and self.lib.iiseg_conv_gemm_bf16_supported(C.byref(d)):
            lib = self.lib
            if self._W16 is None:
                self._W16 = torch.empty(lib.iiseg_conv_gemm_bf16_weight_bytes(C.byref(d)) // 2,
                                        dtype=torch.bfloat16, device=self.W.device)
                check(lib.iiseg_conv_gemm_bf16_pack(_stream(), C.byref(d), _ptr(self.W), self.so,
                                                    self.sc, _ptr(self._W16, torch.bfloat16)),
                      'iiseg_conv_gemm_bf16_pack')
            ws = _wino_workspace((lib.iiseg_conv_gemm_bf16_workspace_bytes(C.byref(d)) + 3) // 4,
                                 x1.device)
            ev0 = _ev() if prof is not None else None
            check(lib.iiseg_conv_gemm_bf16(_stream(), C.byref(d), _ptr(x1),
                                           _ptr(self._W16, torch.bfloat16), _ptr(self.b), _ptr(ws),
                                           _ptr(out)), 'iiseg_conv_gemm_bf16')
            if prof is not None:
                prof.append(('wino_gemm_bf16_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        if gemm_shape and CONV_GEMM and x2 is None and not unpool and \
                add is None and self.lib.iiseg_conv_gemm_supported(C.byref(d)):
            ws = _wino_workspace(self.lib.iiseg_conv_gemm_workspace_elems(C.byref(d)), x1.device)
            args = (C.byref(d), _ptr(x1), _ptr(wp), _ptr(self.b), _ptr(ws), _ptr(out))
            if prof is None:
                check(self.lib.iiseg_conv_gemm_f32(_stream(), *args, 7), 'iiseg_conv_gemm_f32')
                return out
            names = ('gemm_im2col_kernel', 'wino_gemm_kernel', 'gemm_output_kernel')
            ev0 = _ev()
            for i, stage in enumerate((1, 2, 4)):        # the three kernels separately
                check(self.lib.iiseg_conv_gemm_f32(_stream(), *args, stage), 'iiseg_conv_gemm_f32')
                ev1 = _ev()
                prof.append((names[i], self.flops(B, d.OH, d.OW) if i == 1 else 0.0, ev0, ev1))
                ev0 = ev1
            return out
        if dt == torch.float64 and CONV_GEMM and (d.KH, d.KW) == (1, 1) and d.C1 >= 1024 and x2 is None and \
                not unpool and add is None and self.lib.iiseg_conv_gemm_f64_supported(C.byref(d)):
            # deep 1x1 layers (fc6 after im2col, fc7, score_fr): split-K GEMM on the Winograd path's kernel
            ws = _wino_workspace64(self.lib.iiseg_conv_gemm_f64_workspace_elems(C.byref(d)), x1.device)
            ev0 = _ev() if prof is not None else None
            check(self.lib.iiseg_conv_gemm_f64(_stream(), C.byref(d), _ptr(x1, dt), _ptr(wp, dt),
                                               _ptr(self.b, dt), _ptr(ws, dt), _ptr(out, dt)),
                  'iiseg_conv_gemm_f64')
            if prof is not None:
                prof.append(('wino64_gemm_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
            return out
        if dt == torch.float64 and self.wino_f64 and not self.via_im2col and \
                self.lib.iiseg_conv_wino_f64_supported(C.byref(d)) and \
                self._form_by_full_map(self.lib.iiseg_conv_wino_f64_supported, d):
            return self._call_wino_f64(d, x1, x2, pre, pooled, add, out, prof, B)
        ev0 = _ev() if prof is not None else None
        if dt == torch.float64:
            check(self.lib.iiseg_conv_f64(_stream(), C.byref(d), _ptr(x1, dt), _ptr(x2, dt),
                                          _ptr(pre, dt), _ptr(pooled, dt), _ptr(wp, dt),
                                          _ptr(self.b, dt), _ptr(add, dt), _ptr(out, dt)),
                  'iiseg_conv_f64')
        else:
            check(self.lib.iiseg_conv_f32(_stream(), C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre),
                                          _ptr(pooled), _ptr(wp), _ptr(ktab, torch.int32),
                                          _ptr(self.b), _ptr(add), _ptr(out)), 'iiseg_conv_f32')
        if prof is not None:
            ev1 = _ev()
            kern = self.kernel
            if dt == torch.float64 and self.lib.iiseg_conv_halo_f64_supported(C.byref(d)):
                kern = 'conv_halo_f64_kernel'
            if dt == torch.float32 and (add is None or self.b is None) and \
                    self.lib.iiseg_conv_small_supported(C.byref(d)):
                kern = 'conv_small_f32_kernel'         # vector ALU, HBM-bound: algorithmic bytes for the roofline
                KERNEL_BYTES[kern] = KERNEL_BYTES.get(kern, 0.0) + \
                    4.0 * B * (self.Cin * d.H * d.W + (2 if add is not None else 1) * self.Cout * d.OH * d.OW)
            if kern == 'conv_halo_f32_kernel' and C2 > 0 and C1 % 4:
                kern = 'conv_taps_f32_kernel'      # a k-tile would straddle the two sources
            prof.append((kern, self.flops(B, d.OH, d.OW), ev0, ev1))
        return out


    def bnrelu_conv(self, stack, n, bn, out, out_c0):
        """BN_ReLU_Conv as one kernel (iiseg_conv_bnrelu_f32): the conv reads the first `n`
        channels of `stack` (B, cap, H, W) and normalises + rectifies them while staging
        (bn = (beta, gamma, mean, inv_std), batch statistics); the result goes to channels
        [out_c0, out_c0 + Cout) of `out`.  Returns None when the layer has no such kernel (the
        caller then runs bn_relu + the plain conv)."""
        if self.dtype != torch.float32 or not BNRELU_FUSE or (self.KH, self.KW) != (3, 3) or \
                self.Cout > 16 or self.dil != 1 or self.transposed:
            return None
        B, cap, H, W = stack.shape
        pl = None
        out_slice = (out.shape[1], int(out_c0))
        d, wp, ktab = self._plan(B, n, 0, H, W, None, None, False, out_slice, pl)
        if not self.lib.iiseg_conv_bnrelu_supported(C.byref(d)) or \
                tuple(out.shape[2:]) != (d.OH, d.OW) or out_c0 + self.Cout > out.shape[1]:
            return None
        prof = CONV_PROFILE
        beta, gamma, mean, inv_std = bn
        ev0 = _ev() if prof is not None else None
        check(self.lib.iiseg_conv_bnrelu_f32(_stream(), C.byref(d), _ptr(stack), cap * H * W,
                                             _ptr(beta), _ptr(gamma), _ptr(mean), _ptr(inv_std),
                                             _ptr(wp), _ptr(ktab, torch.int32), _ptr(self.b),
                                             _ptr(out)), 'iiseg_conv_bnrelu_f32')
        if prof is not None:
            prof.append((self.kernel, self.flops(B, d.OH, d.OW), ev0, _ev()))
        return out

    def _form_by_full_map(self, supported, d):
        """A layer's kernel FORM (Winograd or direct) is decided on its FULL-MAP geometry, not on the
        window a launch covers: a form with a size limit (workspace within 32-bit offsets) that takes a
        small window but not the full map would otherwise compute the loop-invariant border with one
        kernel and the recomputed window with another -- and the two differ in their last bits, which the
        exact work eliminations rule out.  Cached per input geometry."""
        key = (id(supported), d.B, d.C1, d.C2, d.H, d.W, d.flags)
        ok = self._form_cache.get(key)
        if ok is None:
            full = ConvDesc()
            C.memmove(C.byref(full), C.byref(d), C.sizeof(ConvDesc))
            fh, fw = self.out_hw(d.H, d.W)
            full.oy0, full.ox0, full.OH, full.OW = 0, 0, fh, fw
            full.out_H = full.out_W = full.out_y0 = full.out_x0 = 0
            if full.AH:
                full.AH, full.AW, full.ay0, full.ax0 = max(full.AH, fh), max(full.AW, fw), 0, 0
            ok = self._form_cache[key] = bool(supported(C.byref(full)))
        return ok

    def _f64_halo_runs(self):
        """float64: does the LIBRARY run this layer on the halo-tile kernel (the only float64 kernel with the
        fused pool and the mask bytes)?  Asked once per layer on a representative request -- the switch
        IISEG_F64_HALO, the packed-weight shape (Kpad % 36, Mpad % 64) -- so that `pool_fusable` / `mask_ok`
        never promise what iiseg_conv_pool_f64 / iiseg_conv_mask_f64 would refuse."""
        ok = self._form_cache.get('f64_halo')
        if ok is None:
            d = ConvDesc()
            d.B, d.C1, d.C2, d.H, d.W = 1, self.Cin, 0, 16, 16
            d.Cout, d.KH, d.KW, d.pad, d.dil = self.Cout, self.KH, self.KW, self.pad, self.dil
            d.OH, d.OW = self.out_hw(16, 16)
            d.flags = CONV_RELU if self.relu else 0
            ok = self.lib.iiseg_conv_plan_f64(C.byref(d)) == 0 and \
                bool(self.lib.iiseg_conv_halo_f64_supported(C.byref(d)))
            self._form_cache['f64_halo'] = ok
        return ok

    def pool_fusable(self, c8=None):
        """True if this layer runs on a halo kernel whose epilogue can do the 2x2 max-pool.  `c8`:
        whether the call will hand it a C8 tensor (None: the form the layer was built for) -- a
        layer built with mma='bf16c8' and called on fp32 NCHW input runs the 'bf16' forms, and
        those decide."""
        if not POOL_FUSE:
            return False
        if self.dtype == torch.float64:
            # float64: the halo-tile kernel's epilogue (conv_halo_f64.hip), layers it runs with > 16 output channels
            return F64_POOL_FUSE and (self.KH, self.KW) == (3, 3) and self.dil == 1 and not self.transposed and \
                not self.wino_f64 and self.Cout > 16 and self._f64_halo_runs()
        if self.dtype != torch.float32:
            return False
        if self.c8 and c8 is not False:
            return True
        if self.mma == 'bf16':
            return self.halo_bf16
        return not (self.wino or self.kernel != 'conv_halo_f32_kernel' or not 16 < self.Cout < 256)

    def mask_ok(self, c8=None):
        """True if this layer can take / produce DePool2D masks as bytes (halo kernels only, and
        only where the byte form runs the very kernel the pre / pooled form runs).  `c8` as in
        `pool_fusable`."""
        if self.c8 and c8 is not False:
            return True
        if self.dtype == torch.float64:
            # float64: layers of the halo-tile kernel (the mask bytes hold float64 comparisons)
            return F64_MASKS and (self.KH, self.KW) == (3, 3) and self.dil == 1 and not self.transposed and \
                not self.wino_f64 and self._f64_halo_runs()
        if self.dtype != torch.float32 or (self.KH, self.KW) != (3, 3) or self.dil != 1 or \
                self.transposed or self.kernel != 'conv_halo_f32_kernel' or self.wino_bf16:
            return False
        return self.halo_bf16 if self.mma == 'bf16' else not self.wino

    def pool_window(self, H, W, region=None, c8=None):
        """If the 2x2 max-pool that follows this layer can be fused into its epilogue: the conv
        window (y0, x0, h, w) to launch so that every pooling window touching `region` (of the
        conv output; None = the whole map) is whole -- even origin, even extent unless it ends at
        the map's last row / column.  None if the layer does not run on the halo kernel."""
        if not self.pool_fusable(c8):
            return None
        fh, fw = self.out_hw(H, W)
        if region is None:
            return (0, 0, fh, fw)
        y0, x0 = region[0] & ~1, region[1] & ~1
        y1, x1 = min((region[0] + region[2] + 1) & ~1, fh), min((region[1] + region[3] + 1) & ~1, fw)
        if region[0] + region[2] == fh:
            y1 = fh
        if region[1] + region[3] == fw:
            x1 = fw
        return (y0, x0, y1 - y0, x1 - x0)

    def c8_tiling(self, d, pool):
        """(kind, th, tw, quad) of a C8 launch: include/iiseg.h iiseg_conv_c8_tiling."""
        t = (C.c_int32 * 4)()
        check(self.lib.iiseg_conv_c8_tiling(C.byref(d), 1 if pool else 0, t), 'iiseg_conv_c8_tiling')
        return tuple(t)

    def _call_c8_m16(self, x1, window, out, place, mask_in, unpool_hw, out_format, out_c0, in_c, bn,
                     stats=None, fold=None):
        """Layers with at most 16 output channels on bf16 C8 activations (include/iiseg.h,
        iiseg_conv_c8_m16).  in_c: convolve only the first in_c channels of x1 (a dense block's stack);
        out + out_c0: write the 16 channels [out_c0, out_c0 + 16) of the wider C8 tensor `out`;
        bn = (a, b): BatchNorm + ReLU x <- max(a x + b, 0) applied to the input on the way in
        (`bn_fold`); stats = (mean, inv_std, eps): the batch statistics of the 16 produced channels (bf16 C8
        output) into entries [out_c0, out_c0 + 16) of the two vectors, out of the conv's epilogue; fold =
        (beta, gamma, a, b, n) (with stats): the reduction that finishes the statistics also forms the
        (a, b) pair of the NEXT consumer's BatchNorm over the first n channels.  Returns `out`."""
        lib = self.lib
        unpool = mask_in is not None
        B = x1.shape[0]
        in_ctot = c8_ctot(x1)            # (x1 may be a channel slice of a wider tensor: the parent's count)
        C1 = x1.shape[1] * 8 if in_c is None else int(in_c)
        if C1 % 16 or C1 > x1.shape[1] * 8 or C1 < self.Cin or C1 - self.Cin >= 16:
            raise RuntimeError('conv of %d input channels on the first %d of %d C8 channels'
                               % (self.Cin, C1, in_ctot))
        if unpool:
            if unpool_hw is None or not is_c8_mask(mask_in) or tuple(mask_in.shape) != tuple(x1.shape) or \
                    (unpool_hw[0] // 2, unpool_hw[1] // 2) != tuple(x1.shape[2:4]) or C1 != in_ctot or \
                    not x1.is_contiguous():
                raise RuntimeError('C8 DePool2D input: up %s, mask %s, unpool_hw %s'
                                   % (tuple(x1.shape), tuple(mask_in.shape), unpool_hw))
            H, W = int(unpool_hw[0]), int(unpool_hw[1])
        else:
            H, W = x1.shape[2], x1.shape[3]
        fullH, fullW = self.out_hw(H, W)
        oy0, ox0, OH, OW = window if window is not None else (0, 0, fullH, fullW)
        fmt = out_format or ('c8' if self.Cout % 8 == 0 else 'nchw')
        if fmt not in ('c8', 'nchw'):
            raise RuntimeError('16-channel C8 layer: bf16 C8 or fp32 NCHW output')
        d = ConvDesc()
        d.B, d.C1, d.C2, d.H, d.W = B, C1, 0, H, W
        d.Cout, d.KH, d.KW, d.pad, d.dil = self.Cout, 3, 3, self.pad, 1
        d.oy0, d.ox0, d.OH, d.OW = oy0, ox0, OH, OW
        d.flags = (CONV_RELU if self.relu else 0) | (CONV_UNPOOL if unpool else 0)
        if place is not None:
            if out is None:
                raise RuntimeError('placement needs a target')
            d.out_H, d.out_W, d.out_y0, d.out_x0 = out.shape[2], out.shape[3], int(place[0]), int(place[1])
        if out is None:
            if out_c0 is not None:
                raise RuntimeError('out_c0 needs a target')
            out = torch.empty((B, self.Cout, OH, OW), dtype=torch.float32, device=x1.device) if fmt == 'nchw' \
                else torch.empty((B, 2, OH, OW, 8), dtype=torch.bfloat16, device=x1.device)
        if fmt == 'nchw':
            ok = out.dim() == 4 and out.dtype == torch.float32 and out.shape[1] == self.Cout and out_c0 is None
        else:
            ok = out.dim() == 5 and out.dtype == torch.bfloat16 and out.shape[1] % 2 == 0
            if ok and (out_c0 is not None or out.shape[1] != 2 or not out.is_contiguous()):
                c0 = int(out_c0 or 0)
                ok = c0 % 16 == 0 and c0 + 16 <= out.shape[1] * 8
                d.out_ctot, d.out_c0 = c8_ctot(out), c0
        if not ok or out.shape[0] != B or (place is None and tuple(out.shape[2:4]) != (OH, OW)):
            raise RuntimeError('bad output target %s for a 16-channel C8 layer' % (tuple(out.shape),))
        if bn is not None:
            if len(bn) != 2 or any(t.dtype != torch.float32 or t.numel() < C1 for t in bn):
                raise RuntimeError('bn = (a, b): float32, at least %d entries' % C1)
        if self._W16c8 is None:     # (packed once per layer, for the REAL channel count, shared with _call_c8:
            #                          a captured graph may point at it -- never replaced after first use)
            dp = ConvDesc()
            dp.B, dp.C1, dp.C2, dp.H, dp.W = 1, self.Cin, 0, 8, 8
            dp.Cout, dp.KH, dp.KW, dp.pad, dp.dil = self.Cout, 3, 3, 1, 1
            dp.OH, dp.OW = 8, 8
            self._W16c8 = torch.empty(lib.iiseg_conv_halo_bf16_weight_bytes(C.byref(dp)) // 2,
                                      dtype=torch.bfloat16, device=self.W.device)
            check(lib.iiseg_conv_halo_bf16_pack(_stream(), C.byref(dp), _ptr(self.W), self.so, self.sc,
                                                _ptr(self._W16c8, torch.bfloat16)),
                  'iiseg_conv_halo_bf16_pack')
        dtp = lambda t: None if t is None else (_c8ptr(t) if is_c8(t) else _ptr(t, t.dtype))
        prof = CONV_PROFILE
        ev0 = _ev() if prof is not None else None
        # split-K launches (small maps, long channel loops) sum their slices through a scratch buffer
        nws = lib.iiseg_conv_c8_m16_workspace_bytes(C.byref(d))
        ws = _m16_workspace(nws, x1.device) if nws else None
        if stats is not None and (fmt != 'c8' or stats[0].dtype != torch.float32 or
                                  stats[1].dtype != torch.float32 or
                                  min(stats[0].numel(), stats[1].numel()) < (out_c0 or 0) + 16):
            raise RuntimeError('stats = (mean, inv_std, eps): float32 vectors covering the produced slice, C8 output')
        if fold is not None and (stats is None or any(t.dtype != torch.float32 or t.numel() < fold[4]
                                                      for t in fold[:4])):
            raise RuntimeError('fold = (beta, gamma, a, b, n) needs stats and float32 vectors of n entries')
        check(lib.iiseg_conv_c8_m16_ws(_stream(), C.byref(d), dtp(x1), in_ctot, dtp(mask_in),
                                       None if bn is None else _ptr(bn[0]), None if bn is None else _ptr(bn[1]),
                                       dtp(self._W16c8), _ptr(self.b), dtp(out), 3 if fmt == 'nchw' else 1,
                                       None if ws is None else C.c_void_p(ws.data_ptr()), int(nws),
                                       None if stats is None else _ptr(stats[0]),
                                       None if stats is None else _ptr(stats[1]),
                                       float(stats[2]) if stats is not None else 0.0,
                                       *((_ptr(fold[0]), _ptr(fold[1]), _ptr(fold[2]), _ptr(fold[3]), int(fold[4]))
                                         if fold is not None else (None, None, None, None, 0))),
              'iiseg_conv_c8_m16_ws')
        if prof is not None:
            prof.append(('conv_c8_m16_kernel', self.flops(B, OH, OW), ev0, _ev()))
            if CONV_PROFILE_INFO is not None:
                CONV_PROFILE_INFO.append(dict(
                    Cin=self.Cin, Cout=self.Cout, C1=C1, C2=0, H=H, W=W, OH=OH, OW=OW, B=B,
                    unpool=unpool, pool=False, add=0, kind=3 if fmt == 'nchw' else 1, flat='m16'))
        return out

    def _call_c8(self, x1, x2, add, add_off, window, out, place, pool_out, mask_in, unpool_hw,
                 mask_out, store_out, out_format, out_c0=None, in_c=None, bn=None, stats=None, fold=None,
                 zins=False):
        """The layer on bf16 C8 activations (include/iiseg.h, iiseg_conv_c8).  x1 / x2 / pool_out:
        C8 tensors (`is_c8`); add: C8 bf16 or C8 fp32 (float32, same 5-D shape); mask_in / mask_out:
        uint8 (B, C/8, h, w, 8).  out_format: 'c8' (default), 'c8f32', or 'nchw' (fp32 NCHW, the
        class-score layer: default when Cout is not a multiple of 8).  Returns the output tensor
        (None with store_out=False)."""
        lib = self.lib
        if not self.c8:
            raise RuntimeError("C8 input needs a 3x3 layer built with mma='bf16c8'")
        if C8_M16 and self.Cout <= 16 and not self.x3 and x2 is None and add is None and pool_out is None and \
                store_out and (out_format or 'c8') in ('c8', 'nchw') and not zins:
            return self._call_c8_m16(x1, window, out, place, mask_in, unpool_hw, out_format, out_c0, in_c, bn,
                                     stats, fold)
        if stats is not None or fold is not None:
            raise RuntimeError('stats: layers with at most 16 output channels')
        if in_c is not None or bn is not None:
            raise RuntimeError('in_c / bn on C8 input: layers with at most 16 output channels')
        unpool = mask_in is not None
        x3 = self.x3
        pair = 2 if x3 else 1                     # chunk planes of a kind-1 tensor per channel chunk
        if x1.shape[1] % pair or (x3 and x2 is not None):
            raise RuntimeError('bf16x3 layer: hi / lo pair input, no channel concat')
        B, CC1 = x1.shape[0], x1.shape[1] // pair
        if unpool:
            if unpool_hw is None or not is_c8_mask(mask_in) or \
                    tuple(mask_in.shape) != (B, CC1) + tuple(x1.shape[2:]) or \
                    (unpool_hw[0] // 2, unpool_hw[1] // 2) != tuple(x1.shape[2:4]) or x2 is not None:
                raise RuntimeError('C8 DePool2D input: up %s, mask %s, unpool_hw %s'
                                   % (tuple(x1.shape), tuple(mask_in.shape), unpool_hw))
            H, W = int(unpool_hw[0]), int(unpool_hw[1])
        elif zins:
            # the logical input is x1 zero-inserted (include/iiseg.h IISEG_CONV_ZINS): TransitionUp
            if x3 or x2 is not None or self.pad != 0:
                raise RuntimeError('zero-inserted input: plain single-source valid layers')
            H, W = 2 * x1.shape[2] + 3, 2 * x1.shape[3] + 3
        else:
            H, W = x1.shape[2], x1.shape[3]
        # x1 as a channel slice of a wider C8 tensor (a dense block's stack): the parent's channel count
        x1_ctot = 0 if x1.is_contiguous() else c8_ctot(x1)
        if x1_ctot and (x3 or unpool):
            raise RuntimeError('channel-slice input: plain bf16 launches only')
        CC2 = 0
        if x2 is not None:
            if not is_c8(x2) or x2.shape[0] != B or tuple(x2.shape[2:4]) != (H, W):
                raise RuntimeError('concat shapes %s vs %s' % (tuple(x1.shape), tuple(x2.shape)))
            CC2 = x2.shape[1]
            if CC1 * 8 + CC2 * 8 != self.Cin:
                raise RuntimeError('C8 concat needs unpadded sources')
        if (CC1 + CC2) * 8 < self.Cin or (CC1 % 2) or (CC2 % 2):
            raise RuntimeError('conv expects %d input channels in whole 16-channel groups, got '
                               '%d + %d chunks' % (self.Cin, CC1, CC2))
        fullH, fullW = self.out_hw(H, W)
        oy0, ox0, OH, OW = window if window is not None else (0, 0, fullH, fullW)
        if not store_out and pool_out is not None:
            # only the pool is kept: the unpaired last row / column of an odd map (Pool2DLayer
            # ignore_border) feeds nothing -- do not compute it
            OH, OW = max(OH & ~1, 2 if OH > 1 else 1), max(OW & ~1, 2 if OW > 1 else 1)
        fmt = out_format or ('c8' if self.Cout % 8 == 0 else 'nchw')
        kind = {'c8': 1, 'c8f32': 2, 'nchw': 3}[fmt]
        d = ConvDesc()
        d.B, d.C1, d.C2, d.H, d.W = B, CC1 * 8, CC2 * 8, H, W
        d.Cout, d.KH, d.KW, d.pad, d.dil = self.Cout, 3, 3, self.pad, 1
        d.oy0, d.ox0, d.OH, d.OW = oy0, ox0, OH, OW
        d.flags = (CONV_RELU if self.relu else 0) | (CONV_UNPOOL if unpool else 0) | \
            (CONV_X3 if x3 else 0) | (CONV_ZINS if zins else 0)
        add_kind = 0
        if add is not None:
            add_kind = 1 if is_c8(add) else 2
            if add.dim() != 5 or add.shape[0] != B or \
                    add.shape[1] != c8_chunks(self.Cout) * (pair if add_kind == 1 else 1) or \
                    add.dtype not in (torch.bfloat16, torch.float32):
                raise RuntimeError('add tensor shape %s' % (tuple(add.shape),))
            d.AH, d.AW, d.ay0, d.ax0 = add.shape[2], add.shape[3], add_off[0], add_off[1]
        oc8 = (self.Cout + 15) // 16 * 2
        if place is not None:
            if out is None:
                raise RuntimeError('placement needs a target')
            d.out_H, d.out_W, d.out_y0, d.out_x0 = out.shape[2], out.shape[3], int(place[0]), int(place[1])
        if not store_out:
            if pool_out is None:
                raise RuntimeError('store_out=False needs pool_out')
            out, kind = None, 0
        elif out is None:
            if fmt == 'nchw':
                out = torch.empty((B, self.Cout, OH, OW), dtype=torch.float32, device=x1.device)
            else:
                out = torch.empty((B, oc8 * (pair if fmt == 'c8' else 1), OH, OW, 8), device=x1.device,
                                  dtype=torch.bfloat16 if fmt == 'c8' else torch.float32)
        if out_c0 is not None:
            # the Cout channels written as chunk planes [out_c0 / 8, ...) of a wider C8 tensor
            if out is None or fmt != 'c8' or x3 or not is_c8(out) or out_c0 % 16 or \
                    out_c0 + oc8 * 8 > out.shape[1] * 8 or out.shape[0] != B or \
                    (place is None and tuple(out.shape[2:4]) != (OH, OW)):
                raise RuntimeError('bad C8 output slice %s @%s' % (None if out is None else tuple(out.shape), out_c0))
            d.out_ctot, d.out_c0 = c8_ctot(out), int(out_c0)
        elif out is not None:
            ok = (out.dim() == 4 and out.dtype == torch.float32 and out.shape[1] == self.Cout) \
                if fmt == 'nchw' else \
                (out.dim() == 5 and out.shape[1] == oc8 * (pair if fmt == 'c8' else 1) and
                 out.dtype == (torch.bfloat16 if fmt == 'c8' else torch.float32))
            if not ok or out.shape[0] != B or (place is None and tuple(out.shape[2:4]) != (OH, OW)):
                raise RuntimeError('bad C8 output target %s' % (tuple(out.shape),))
        if pool_out is not None:
            if not is_c8(pool_out) or \
                    tuple(pool_out.shape) != (B, oc8 * pair, fullH // 2, fullW // 2, 8):
                raise RuntimeError('pool_out shape %s' % (tuple(pool_out.shape),))
            if mask_out is not None and (not is_c8_mask(mask_out) or
                                         tuple(mask_out.shape) != (B, oc8, fullH // 2, fullW // 2, 8)):
                raise RuntimeError('mask_out: uint8 (B, Cout/8, H/2, W/2, 8)')
        elif mask_out is not None:
            raise RuntimeError('mask_out needs pool_out')
        if self._W16c8 is None:
            # packed for the REAL channel count: the rows of padding channels are zero
            dp = ConvDesc()
            dp.B, dp.C1, dp.C2, dp.H, dp.W = 1, self.Cin, 0, 8, 8
            dp.Cout, dp.KH, dp.KW, dp.pad, dp.dil = self.Cout, 3, 3, 1, 1
            dp.OH, dp.OW = 8, 8
            Wsrc, so, sc = self.W, self.so, self.sc
            if x3:
                # two k-groups of whole 16-channel k-tiles [W_hi | W_lo]; the split values are bf16
                # numbers, the pack's rounding leaves them as they are
                Cp = (self.Cin + 15) // 16 * 16
                Wsrc = torch.empty((self.Cout, 2 * Cp, 3, 3), dtype=torch.float32, device=self.W.device)
                check(lib.iiseg_conv_c8_split_weights(_stream(), _ptr(self.W), self.so, self.sc,
                                                      self.Cout, self.Cin, _ptr(Wsrc)),
                      'iiseg_conv_c8_split_weights')
                dp.C1, so, sc = 2 * Cp, 2 * Cp * 9, 9
            self._W16c8 = torch.empty(lib.iiseg_conv_halo_bf16_weight_bytes(C.byref(dp)) // 2,
                                      dtype=torch.bfloat16, device=self.W.device)
            check(lib.iiseg_conv_halo_bf16_pack(_stream(), C.byref(dp), _ptr(Wsrc), so, sc,
                                                _ptr(self._W16c8, torch.bfloat16)),
                  'iiseg_conv_halo_bf16_pack')
            del Wsrc
        dtp = lambda t: None if t is None else (_c8ptr(t) if (is_c8(t) and t is not self._W16c8) else
                                                _ptr(t, t.dtype))
        if out is not None and is_c8(out) and not out.is_contiguous():
            c8_ctot(out)                   # (validates the slice layout)
        # (the pool rides in the conv's epilogue on every pixel tiling: with pool_out the pixels of a
        # tile are ordered by 2x2 pooling windows)
        conv_out, conv_kind, conv_pool, conv_mask = out, kind, pool_out, mask_out
        prof = CONV_PROFILE
        ev0 = _ev() if prof is not None else None
        check(lib.iiseg_conv_c8_slice(_stream(), C.byref(d), dtp(x1), int(x1_ctot), dtp(x2), dtp(mask_in),
                                      dtp(self._W16c8), _ptr(self.b), dtp(add), add_kind, dtp(conv_out),
                                      conv_kind, dtp(conv_pool), dtp(conv_mask)), 'iiseg_conv_c8_slice')
        if prof is not None:
            prof.append(('conv_c8_kernel<x3>' if x3 else 'conv_c8_kernel', self.flops(B, OH, OW), ev0,
                         _ev()))
            if CONV_PROFILE_INFO is not None:
                CONV_PROFILE_INFO.append(dict(
                    Cin=self.Cin, Cout=self.Cout, C1=d.C1, C2=d.C2, H=H, W=W, OH=OH, OW=OW, B=B,
                    unpool=unpool, pool=pool_out is not None, add=add_kind, kind=conv_kind,
                    flat=self.c8_tiling(d, pool_out is not None)))
        return out

    def _call_halo_bf16(self, d, x1, x2, pre, pooled, add, out, pool_out, mask_in, mask_out, prof,
                        B, H, W):
        """Direct 3x3 form on the bf16 matrix pipe (include/iiseg.h, iiseg_conv_halo_bf16)."""
        if pool_out is not None:
            fh, fw = self.out_hw(H, W)
            if tuple(pool_out.shape) != (B, self.Cout, fh // 2, fw // 2):
                raise RuntimeError('pool_out shape %s' % (tuple(pool_out.shape),))
        if self._W16 is None:
            self._W16 = torch.empty(self.lib.iiseg_conv_halo_bf16_weight_bytes(C.byref(d)) // 2,
                                    dtype=torch.bfloat16, device=self.W.device)
            check(self.lib.iiseg_conv_halo_bf16_pack(_stream(), C.byref(d), _ptr(self.W), self.so,
                                                     self.sc, _ptr(self._W16, torch.bfloat16)),
                  'iiseg_conv_halo_bf16_pack')
        ev0 = _ev() if prof is not None else None
        check(self.lib.iiseg_conv_halo_bf16(_stream(), C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre),
                                            _ptr(pooled), _ptr(self._W16, torch.bfloat16),
                                            _ptr(self.b), _ptr(add), _ptr(out), _ptr(pool_out),
                                            _ptr(mask_in, torch.uint8),
                                            _ptr(mask_out, torch.uint8)),
              'iiseg_conv_halo_bf16')
        if prof is not None:
            prof.append(('conv_halo_bf16_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
        return out

    def _call_wino_bf16(self, d, x1, x2, pre, pooled, add, out, prof):
        """bf16-operand Winograd form (include/iiseg.h, iiseg_conv_wino_bf16): input transform ->
        V16, then one kernel for the 16 GEMMs + output transform + epilogue."""
        lib = self.lib
        if self._U16 is None:
            self._U16 = torch.empty(lib.iiseg_conv_wino_bf16_weight_bytes(C.byref(d)) // 2,
                                    dtype=torch.bfloat16, device=self.W.device)
            check(lib.iiseg_conv_wino_bf16_pack(_stream(), C.byref(d), _ptr(self.W), self.so,
                                                self.sc, _ptr(self._U16, torch.bfloat16)),
                  'iiseg_conv_wino_bf16_pack')
        ws = _wino_workspace((lib.iiseg_conv_wino_bf16_workspace_bytes(C.byref(d)) + 3) // 4,
                             x1.device)
        args = (C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre), _ptr(pooled),
                _ptr(self._U16, torch.bfloat16), _ptr(self.b), _ptr(add), _ptr(ws), _ptr(out))
        if prof is None:
            check(lib.iiseg_conv_wino_bf16(_stream(), *args, 3), 'iiseg_conv_wino_bf16')
            return out
        r0, c0 = d.oy0 - ((d.oy0 - d.tile_y0) & 1), d.ox0 - ((d.ox0 - d.tile_x0) & 1)
        T = d.B * ((d.oy0 + d.OH - r0 + 1) // 2) * ((d.ox0 + d.OW - c0 + 1) // 2)
        kc = (self.Cin + 63) // 64 * 64
        gemm_flops = 16 * 2.0 * kc * self.Cout * T            # multiplies actually issued
        ev0 = _ev()
        for name, stage, fl in (('wino_input_bf16_kernel', 1, 0.0),
                                ('wino_fused_bf16_kernel', 2, gemm_flops)):
            check(lib.iiseg_conv_wino_bf16(_stream(), *args, stage), 'iiseg_conv_wino_bf16')
            ev1 = _ev()
            prof.append((name, fl, ev0, ev1))
            ev0 = ev1
        return out

    def _call_wino_f64(self, d, x1, x2, pre, pooled, add, out, prof, B):
        """float64 Winograd form (include/iiseg.h, iiseg_conv_wino_f64)."""
        lib = self.lib
        dt = torch.float64
        if self._U is None:
            self._U = torch.empty(lib.iiseg_conv_wino_f64_weight_elems(C.byref(d)), dtype=dt,
                                  device=self.W.device)
            check(lib.iiseg_conv_wino_pack_f64(_stream(), C.byref(d), _ptr(self.W, dt), self.so, self.sc,
                                               _ptr(self._U, dt)), 'iiseg_conv_wino_pack_f64')
        ws = _wino_workspace64(lib.iiseg_conv_wino_f64_workspace_elems(C.byref(d)), x1.device)
        ev0 = _ev() if prof is not None else None
        check(lib.iiseg_conv_wino_f64(_stream(), C.byref(d), _ptr(x1, dt), _ptr(x2, dt), _ptr(pre, dt),
                                      _ptr(pooled, dt), _ptr(self._U, dt),
                                      _ptr(self.b, dt), _ptr(add, dt), _ptr(ws, dt), _ptr(out, dt)),
              'iiseg_conv_wino_f64')
        if prof is not None:
            prof.append(('wino64_gemm_kernel', self.flops(B, d.OH, d.OW), ev0, _ev()))
        return out

    def _call_wino(self, d, x1, x2, pre, pooled, add, out, prof):
        """Winograd F(2x2,3x3) form of the layer (include/iiseg.h, iiseg_conv_wino_f32)."""
        lib = self.lib
        if self._U is None:
            self._U = torch.empty(lib.iiseg_conv_wino_weight_elems(C.byref(d)), dtype=torch.float32,
                                  device=self.W.device)
            check(lib.iiseg_conv_wino_pack_f32(_stream(), C.byref(d), _ptr(self.W), self.so, self.sc,
                                               _ptr(self._U)), 'iiseg_conv_wino_pack_f32')
        ws = _wino_workspace(lib.iiseg_conv_wino_workspace_elems(C.byref(d)), x1.device)
        args = (C.byref(d), _ptr(x1), _ptr(x2), _ptr(pre), _ptr(pooled), _ptr(self._U),
                _ptr(self.b), _ptr(add), _ptr(ws), _ptr(out))
        # (a skip-add would be read by the fused kernel's epilogue, uncoalesced and with nothing to
        # hide its latency: such layers keep the separate, coalesced output transform)
        fused = 8 if (self.Cin <= WINO_FUSED_MAX_CIN and self.Cin % 32 == 0 and add is None) else 0
        if prof is None:
            check(lib.iiseg_conv_wino_f32(_stream(), *args, 7 | fused), 'iiseg_conv_wino_f32')
            return out
        # profiling: the kernels separately, events around each
        stages = (1 | fused, 2 | fused) if fused else (1, 2, 4)
        names = ('wino_input_kernel', 'wino_fused_kernel' if fused else 'wino_gemm_kernel',
                 'wino_output_kernel')
        r0, c0 = d.oy0 - ((d.oy0 - d.tile_y0) & 1), d.ox0 - ((d.ox0 - d.tile_x0) & 1)
        T = d.B * ((d.oy0 + d.OH - r0 + 1) // 2) * ((d.ox0 + d.OW - c0 + 1) // 2)
        gemm_flops = 16 * 2.0 * self.Cin * self.Cout * T      # multiplies actually issued
        ev0 = _ev()
        for i, stage in enumerate(stages):
            check(lib.iiseg_conv_wino_f32(_stream(), *args, stage), 'iiseg_conv_wino_f32')
            ev1 = _ev()
            prof.append((names[i], gemm_flops if i == 1 else 0.0, ev0, ev1))
            ev0 = ev1
        return out


DECONV_PHASE = True      # K = 2 stride layers on the output-phase kernel (csrc/deconv_phase.hip)


class Deconv:
    """Small-channel transposed convolution (Lasagne Deconv2DLayer W[in,out,k,k], P3)."""

    def __init__(self, W, b, stride, device='cuda', dtype=torch.float32):
        self.dtype = dtype
        self.W = torch.as_tensor(W).to(dtype).contiguous().to(device)
        self.b = None if b is None else torch.as_tensor(b).to(dtype).contiguous().to(device)
        self.Cin, self.Cout, self.K, k2 = self.W.shape
        if k2 != self.K:
            raise RuntimeError('square kernels only')
        self.stride = int(stride)
        self._wp = None             # weights packed for the output-phase kernel (first call)
        self.last_form = None

    def out_hw(self, H, W):
        return (H - 1) * self.stride + self.K, (W - 1) * self.stride + self.K

    def __call__(self, x, add=None, add_off=(0, 0), window=None, out=None):
        dt = self.dtype
        B, Cin, H, W = x.shape
        if Cin != self.Cin:
            raise RuntimeError('deconv expects %d channels, got %d' % (self.Cin, Cin))
        fullH, fullW = self.out_hw(H, W)
        oy0, ox0, OH, OW = window if window is not None else (0, 0, fullH, fullW)
        d = DeconvDesc()
        d.B, d.Cin, d.H, d.W, d.Cout, d.K, d.stride = B, Cin, H, W, self.Cout, self.K, self.stride
        d.oy0, d.ox0, d.OH, d.OW = oy0, ox0, OH, OW
        if add is not None:
            d.AH, d.AW, d.ay0, d.ax0 = add.shape[2], add.shape[3], add_off[0], add_off[1]
            if add.shape[0] != B or add.shape[1] != self.Cout:
                raise RuntimeError('add tensor shape %s' % (tuple(add.shape),))
        if out is None:
            out = torch.empty((B, self.Cout, OH, OW), dtype=dt, device=x.device)
        lib = _lib.load()
        # K = 2 stride (the FCN-8 upsamplers): the output-phase kernel, weights packed once per layer
        if DECONV_PHASE and lib.iiseg_deconv_phase_supported(C.byref(d), 0 if add is None else 1,
                                                             1 if dt == torch.float64 else 0):
            if self._wp is None:
                self._wp = torch.empty(lib.iiseg_deconv_phase_weight_elems(C.byref(d)), dtype=dt, device=x.device)
                check(_fn('deconv_phase_pack', dt)(_stream(), C.byref(d), _ptr(self.W, dt), _ptr(self._wp, dt)),
                      'iiseg_deconv_phase_pack')
            check(_fn('deconv_phase', dt)(_stream(), C.byref(d), _ptr(x, dt), _ptr(self._wp, dt),
                                          _ptr(self.b, dt), _ptr(add, dt), _ptr(out, dt)), 'iiseg_deconv_phase')
            self.last_form = 'phase'
            return out
        check(_fn('deconv', dt)(_stream(), C.byref(d), _ptr(x, dt), _ptr(self.W, dt),
                                _ptr(self.b, dt), _ptr(add, dt), _ptr(out, dt)), 'iiseg_deconv')
        self.last_form = 'gather'
        return out


def is_c8(t):
    """True for a bf16 C8 activation tensor: (B, C/8, H, W, 8) bfloat16 (include/iiseg.h)."""
    return isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16 and t.dim() == 5 and \
        t.shape[-1] == 8


def c8_ctot(t):
    """Channels per image of the C8 TENSOR the operand `t` lies in: `t` may be a channel slice
    `parent[:, c0 // 8:c1 // 8]` of a contiguous C8 tensor (a dense block's stack inside the buffer of its
    resolution) -- same image stride as the parent, data pointer at the slice's first chunk plane.  Raises
    for any other non-contiguous layout."""
    if not is_c8(t):
        raise RuntimeError('expected a bf16 C8 tensor')
    B, C8n, H, W, _ = t.shape
    st = t.stride()
    if tuple(st[1:]) != (H * W * 8, W * 8, 8, 1) or (B > 1 and (st[0] % (H * W * 8) or st[0] < C8n * H * W * 8)):
        raise RuntimeError('C8 operand: a contiguous tensor or a channel slice of one, got strides %s' % (st,))
    return C8n * 8 if B == 1 else st[0] // (H * W * 8) * 8


def _c8ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def is_c8_mask(t):
    return isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.dim() == 5 and t.shape[-1] == 8


def c8_chunks(channels):
    """Chunks of a C8 tensor holding `channels` channels: whole 16-channel k-tiles."""
    return (int(channels) + 15) // 16 * 2


def empty_c8(B, channels, H, W, device, dtype=torch.bfloat16, x3=False):
    """x3: the hi / lo pair of the 'bf16x3' mode, (B, 2 C/8, H, W, 8) (include/iiseg.h,
    IISEG_CONV_X3)."""
    return torch.empty((B, c8_chunks(channels) * (2 if x3 else 1), H, W, 8), dtype=dtype, device=device)


def nchw_to_c8(x, out=None, x3=False):
    """fp32 NCHW -> bf16 C8 (channels padded with zeros to a whole 16-channel group); x3: the
    hi / lo pair."""
    B, Cc, H, W = x.shape
    if out is None:
        out = empty_c8(B, Cc, H, W, x.device, x3=x3)
    if tuple(out.shape) != (B, c8_chunks(Cc) * (2 if x3 else 1), H, W, 8) or not is_c8(out):
        raise RuntimeError('nchw_to_c8 target %s' % (tuple(out.shape),))
    fn = _lib.load().iiseg_nchw_to_c8x3 if x3 else _lib.load().iiseg_nchw_to_c8
    check(fn(_stream(), _ptr(x), C.c_void_p(out.data_ptr()), B, Cc, H, W, c8_chunks(Cc)),
          'iiseg_nchw_to_c8')
    return out


def nchw_to_c8_slice(x, out8, c0):
    """fp32 NCHW (B, C, H, W) -> channels [c0, c0 + C) of the wider C8 tensor `out8` (c0 % 8 == 0; a
    partial last chunk is zero-padded)."""
    B, Cc, H, W = x.shape
    if not is_c8(out8) or c0 % 8 or tuple(out8.shape[2:4]) != (H, W) or out8.shape[0] != B:
        raise RuntimeError('nchw_to_c8_slice target %s @%d' % (tuple(out8.shape), c0))
    check(_lib.load().iiseg_nchw_to_c8_slice(_stream(), _ptr(x), C.c_void_p(out8.data_ptr()), B, Cc, H, W,
                                             c8_ctot(out8) // 8, c0 // 8), 'iiseg_nchw_to_c8_slice')
    return out8


def c8_slice_to_nchw(x8, c0, channels, out=None):
    """Channels [c0, c0 + channels) of the C8 tensor `x8` -> fp32 NCHW."""
    if not is_c8(x8) or c0 % 8:
        raise RuntimeError('c8_slice_to_nchw needs a C8 tensor and c0 % 8 == 0')
    B, _, H, W, _ = x8.shape
    C8n = c8_ctot(x8) // 8
    if out is None:
        out = torch.empty((B, int(channels), H, W), dtype=torch.float32, device=x8.device)
    check(_lib.load().iiseg_c8_slice_to_nchw(_stream(), C.c_void_p(x8.data_ptr()), _ptr(out), B, int(channels),
                                             H, W, C8n, c0 // 8), 'iiseg_c8_slice_to_nchw')
    return out


def c8x3_to_float(t):
    """hi / lo pair (B, 2 C/8, H, W, 8) -> float32 (B, C/8 * 8, H, W) (torch; tests and debugging)."""
    B, C2, H, W, _ = t.shape
    v = t[:, :C2 // 2].to(torch.float32) + t[:, C2 // 2:].to(torch.float32)
    return v.permute(0, 1, 4, 2, 3).reshape(B, C2 // 2 * 8, H, W)


def c8_to_nchw(x8, channels, out=None, x3=False):
    """bf16 C8 -> fp32 NCHW (first `channels` channels); x3: from the hi / lo pair."""
    if not is_c8(x8) or not x8.is_contiguous() or (x3 and x8.shape[1] % 2):
        raise RuntimeError('c8_to_nchw needs a contiguous C8 tensor')
    B, C8n, H, W, _ = x8.shape
    if x3:
        C8n //= 2
    if out is None:
        out = torch.empty((B, int(channels), H, W), dtype=torch.float32, device=x8.device)
    fn = _lib.load().iiseg_c8x3_to_nchw if x3 else _lib.load().iiseg_c8_to_nchw
    check(fn(_stream(), C.c_void_p(x8.data_ptr()), _ptr(out), B, int(channels), H, W, C8n),
          'iiseg_c8_to_nchw')
    return out


def unpool_c8(up, mask, out, window=None):
    """DePool2D materialised on C8 tensors (include/iiseg.h iiseg_unpool_c8): `out` (B, C8, H, W, 8) <- up
    (B, C8, H/2, W/2, 8) under the mask bytes, for the pooled-coordinate `window` (y0, x0, h, w) (default: all);
    the rest of `out` is left as it is."""
    B, C8n, H, W, _ = out.shape
    if not (is_c8(up) and is_c8(out) and is_c8_mask(mask)) or tuple(up.shape) != (B, C8n, H // 2, W // 2, 8) or \
            tuple(mask.shape) != tuple(up.shape) or not (up.is_contiguous() and out.is_contiguous() and
                                                          mask.is_contiguous()):
        raise RuntimeError('unpool_c8: up %s mask %s out %s' % (tuple(up.shape), tuple(mask.shape), tuple(out.shape)))
    y0, x0, wh, ww = window if window is not None else (0, 0, H // 2, W // 2)
    check(_lib.load().iiseg_unpool_c8(_stream(), _c8ptr(up), C.c_void_p(mask.data_ptr()), _c8ptr(out), B * C8n,
                                      H, W, int(y0), int(x0), int(wh), int(ww)), 'iiseg_unpool_c8')
    return out


def pool_mask_c8(pre, pooled, mask, origin, full_hw, window, x3=False):
    """2x2 max-pool (+ DePool2D mask bytes, `mask` may be None) of the pooled-coordinate `window`
    (y0, x0, h, w) from the stored piece `pre` (C8 bf16, or C8 fp32: the unrounded conv results)
    whose corner sits at `origin` of the `full_hw` map, into the full-size `pooled` / `mask` tensors."""
    B, C8n, PH, PW, _ = pre.shape
    y0, x0, wh, ww = window
    if wh <= 0 or ww <= 0:
        return pooled
    if x3:
        check(_lib.load().iiseg_pool_mask_c8x3(
            _stream(), C.c_void_p(pre.data_ptr()), C.c_void_p(pooled.data_ptr()),
            None if mask is None else C.c_void_p(mask.data_ptr()), B, C8n, PH, PW, int(origin[0]),
            int(origin[1]), int(full_hw[0]), int(full_hw[1]), int(y0), int(x0), int(wh), int(ww)),
            'iiseg_pool_mask_c8x3')
        return pooled
    check(_lib.load().iiseg_pool_mask_c8(
        _stream(), C.c_void_p(pre.data_ptr()), 1 if pre.dtype == torch.float32 else 0,
        C.c_void_p(pooled.data_ptr()),
        None if mask is None else C.c_void_p(mask.data_ptr()), B * C8n, PH, PW, int(origin[0]),
        int(origin[1]), int(full_hw[0]), int(full_hw[1]), int(y0), int(x0), int(wh), int(ww)),
        'iiseg_pool_mask_c8')
    return pooled


def maxpool2x2(x, out=None, window=None):
    """`window` = (y0, x0, h, w) in pooled coordinates: only that region of `out` is written."""
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cc, H // 2, W // 2), dtype=x.dtype, device=x.device)
    if window is not None:
        y0, x0, wh, ww = window
        check(_fn('maxpool2x2_window', x.dtype)(_stream(), _ptr(x, x.dtype), _ptr(out, x.dtype),
                                                B * Cc, H, W, y0, x0, wh, ww),
              'iiseg_maxpool2x2_window')
        return out
    check(_fn('maxpool2x2', x.dtype)(_stream(), _ptr(x, x.dtype), _ptr(out, x.dtype), B * Cc, H, W),
          'iiseg_maxpool2x2')
    return out


def unpool_eqmask(up, pre, pooled, out=None, window=None):
    """DePool2D, materialised.  `window` = (y0, x0, h, w): only that region of the (H, W) planes
    is produced (in place in `out`, the rest is left as it is)."""
    B, Cc, H, W = pre.shape
    if tuple(up.shape) != (B, Cc, H // 2, W // 2) or up.shape != pooled.shape:
        raise RuntimeError('unpool shapes: up %s pre %s pooled %s'
                           % (tuple(up.shape), tuple(pre.shape), tuple(pooled.shape)))
    if out is None:
        out = torch.empty_like(pre)
    dt = pre.dtype
    if window is not None:
        y0, x0, wh, ww = window
        check(_fn('unpool_eqmask_window', dt)(_stream(), _ptr(up, dt), _ptr(pre, dt),
                                              _ptr(pooled, dt), _ptr(out, dt), B * Cc, H, W, y0, x0,
                                              wh, ww), 'iiseg_unpool_eqmask_window')
        return out
    check(_fn('unpool_eqmask', dt)(_stream(), _ptr(up, dt), _ptr(pre, dt), _ptr(pooled, dt),
                                   _ptr(out, dt), B * Cc, H, W), 'iiseg_unpool_eqmask')
    return out


def crop_softmax(score, H, W, off=None, out=None, minuend=None):
    """softmax over channels of the (H,W) window of score; `off` defaults to the center crop
    offset (dim - target)//2 (P6).  With `minuend` returns minuend - softmax (de_fn)."""
    B, Cc, SH, SW = score.shape
    sy0, sx0 = off if off is not None else ((SH - H) // 2, (SW - W) // 2)
    dt = score.dtype
    if out is None:
        out = torch.empty((B, Cc, H, W), dtype=dt, device=score.device)
    check(_fn('crop_softmax', dt)(_stream(), _ptr(score, dt), _ptr(minuend, dt), _ptr(out, dt), B,
                                  Cc, SH, SW, sy0, sx0, H, W), 'iiseg_crop_softmax')
    return out


class RefineState:
    """Device-side state of the batched refinement loop: per-image active flag, iteration
    count, last norm and the per-block norm partials (reference loop state of
    iterative_inference.py:258-284 for a whole batch)."""

    def __init__(self, B, H, W, device):
        lib = _lib.load()
        self.B, self.H, self.W = B, H, W
        self.nblk = lib.iiseg_refine_partials(H, W)
        self.active = torch.ones(B, dtype=torch.int32, device=device)
        self.iters = torch.zeros(B, dtype=torch.int32, device=device)
        self.last_norm = torch.zeros(B, dtype=torch.float64, device=device)
        self.partial = torch.zeros(B * self.nblk, dtype=torch.float64, device=device)

    def reset(self):
        self.active.fill_(1)
        self.iters.zero_()
        self.last_norm.zero_()


def refine_update(score, y, state, step, off=None, y8=None):
    """One fused refinement step on y (in place) from the DAE's pre-softmax score map.  `y8`: a
    bf16 C8 tensor (B, chunks, H, W, 8) that also receives the updated map (the DAE's input format
    under mma='bf16c8'; float32 only)."""
    B, Cc, SH, SW = score.shape
    H, W = y.shape[2], y.shape[3]
    sy0, sx0 = off if off is not None else ((SH - H) // 2, (SW - W) // 2)
    dt = y.dtype
    if y8 is not None:
        if dt != torch.float32 or not is_c8(y8) or tuple(y8.shape) != (B, y8.shape[1], H, W, 8) or \
                y8.shape[1] * 8 < Cc:
            raise RuntimeError('y8: bf16 C8 tensor of y\'s geometry')
        check(_lib.load().iiseg_refine_update_c8_f32(
            _stream(), _ptr(score), _ptr(y), _ptr(state.active, torch.int32),
            _ptr(state.partial, torch.float64), C.c_void_p(y8.data_ptr()), y8.shape[1], B, Cc, SH, SW,
            sy0, sx0, H, W, float(step)), 'iiseg_refine_update_c8_f32')
        return
    check(_fn('refine_update', dt)(_stream(), _ptr(score, dt), _ptr(y, dt),
                                   _ptr(state.active, torch.int32),
                                   _ptr(state.partial, torch.float64), B, Cc, SH, SW, sy0, sx0,
                                   H, W, float(step)), 'iiseg_refine_update')


def sqerr_softmax_bwd(score, y, off=None):
    """dE/dscore for E = sum (softmax(score window) - y)^2 (true-gradient mode)."""
    B, Cc, SH, SW = score.shape
    H, W = y.shape[2], y.shape[3]
    sy0, sx0 = off if off is not None else ((SH - H) // 2, (SW - W) // 2)
    dt = y.dtype
    g = torch.empty_like(y)
    check(_fn('sqerr_softmax_bwd', dt)(_stream(), _ptr(score, dt), _ptr(y, dt), _ptr(g, dt), B, Cc,
                                       SH, SW, sy0, sx0, H, W), 'iiseg_sqerr_softmax_bwd')
    return g


def depool_bwd(gout, pre, pooled):
    """Adjoint of DePool2D w.r.t. its input: masked 2x2 sum of gout (B,C,H,W) -> (B,C,H/2,W/2)."""
    dt = gout.dtype
    B, Cc, H, W = pre.shape
    out = torch.empty_like(pooled)
    check(_fn('depool_bwd', dt)(_stream(), _ptr(gout, dt), _ptr(pre, dt), _ptr(pooled, dt),
                                _ptr(out, dt), B * Cc, H, W), 'iiseg_depool_bwd')
    return out


def pool_relu_bwd(gpool, pre, pooled):
    """Adjoint of maxpool2x2(relu(z)) w.r.t. z, given pre = relu(z) and pooled."""
    dt = gpool.dtype
    B, Cc, H, W = pre.shape
    out = torch.empty_like(pre)
    check(_fn('pool_relu_bwd', dt)(_stream(), _ptr(gpool, dt), _ptr(pre, dt), _ptr(pooled, dt),
                                   _ptr(out, dt), B * Cc, H, W), 'iiseg_pool_relu_bwd')
    return out


def grad_update(score, gthrough, y, state, step, off=None):
    """True-gradient step on y (in place): grad = gthrough - 2 (softmax(score) - y)."""
    B, Cc, SH, SW = score.shape
    H, W = y.shape[2], y.shape[3]
    sy0, sx0 = off if off is not None else ((SH - H) // 2, (SW - W) // 2)
    dt = y.dtype
    check(_fn('grad_update', dt)(_stream(), _ptr(score, dt), _ptr(gthrough, dt), _ptr(y, dt),
                                 _ptr(state.active, torch.int32),
                                 _ptr(state.partial, torch.float64), B, Cc, SH, SW, sy0, sx0, H, W,
                                 float(step)), 'iiseg_grad_update')


def refine_finalize(state, eps, nblk=None):
    """`nblk`: partials per image the update kernel wrote (default: those of iiseg_refine_update_*; the context
    module's fused tail writes one per 16 x 64 tile, `ctx_tail`)."""
    lib = _lib.load()
    check(lib.iiseg_refine_finalize(_stream(), _ptr(state.partial, torch.float64),
                                    _ptr(state.active, torch.int32),
                                    _ptr(state.iters, torch.int32),
                                    _ptr(state.last_norm, torch.float64), state.B,
                                    state.nblk if nblk is None else int(nblk),
                                    state.H * state.W, float(eps)), 'iiseg_refine_finalize')


CTX_TAIL = os.environ.get('IISEG_CTX_TAIL', '1') != '0'


def ctx_tail_supported(conv6, conv7, y):
    """Can `ctx_tail` run these two layers on this map?  (fp32, 3x3 'valid' dilation 1 + ReLU, then 1x1
    linear, 12 to 16 padded channels on either side.)"""
    Cc = y.shape[1]
    return (CTX_TAIL and y.dtype == torch.float32 and conv6.dtype == torch.float32 and
            (conv6.KH, conv6.KW, conv6.dil, conv6.pad, conv6.relu) == (3, 3, 1, 0, True) and
            (conv7.KH, conv7.KW, conv7.pad, conv7.relu) == (1, 1, 0, False) and
            conv6.Cin == conv6.Cout == conv7.Cin == conv7.Cout == Cc and 9 <= Cc <= 16 and
            not conv6.transposed and not conv7.transposed)


def ctx_tail(conv6, conv7, x, y, state, step, ycat=None, cat_c0=0, cat_off=(0, 0)):
    """The context module's last two layers + the refinement update as ONE launch (include/iiseg.h,
    iiseg_ctx_tail_f32): x (B, C, H + 2, W + 2) -> y (B, C, H, W) updated in place (and mirrored into channels
    [cat_c0, cat_c0 + C) of `ycat` at `cat_off`).  Returns the number of norm partials per image it wrote
    (pass it to `refine_finalize`)."""
    lib = _lib.load()
    B, Cc, H, W = y.shape
    if tuple(x.shape) != (B, Cc, H + 2, W + 2):
        raise RuntimeError('ctx_tail: x %s for y %s' % (tuple(x.shape), tuple(y.shape)))
    d6, wp6, _ = conv6._plan(B, Cc, 0, H + 2, W + 2, None, None, False)
    d7, wp7, _ = conv7._plan(B, Cc, 0, H, W, None, None, False)
    nblk = lib.iiseg_ctx_tail_partials(H, W)
    if B * nblk > state.partial.numel():
        raise RuntimeError('ctx_tail: the state holds %d partials, %d needed' % (state.partial.numel(), B * nblk))
    cat = (0, 0, 0, 0, 0, 0)
    if ycat is not None:
        if ycat.dim() != 4 or ycat.shape[0] != B or ycat.dtype != torch.float32:
            raise RuntimeError('ctx_tail: ycat %s' % (tuple(ycat.shape),))
        cat = (ycat.shape[1], int(cat_c0), ycat.shape[2], ycat.shape[3], int(cat_off[0]), int(cat_off[1]))
    prof = CONV_PROFILE
    ev0 = _ev() if prof is not None else None
    check(lib.iiseg_ctx_tail_f32(_stream(), _ptr(x), _ptr(wp6), d6.Mpad, _ptr(conv6.b), _ptr(wp7), d7.Mpad,
                                 _ptr(conv7.b), _ptr(y), _ptr(state.active, torch.int32),
                                 _ptr(state.partial, torch.float64), _ptr(ycat), *cat, B, Cc, H, W, float(step)),
          'iiseg_ctx_tail_f32')
    if prof is not None:
        kern = 'ctx_tail_kernel'
        prof.append((kern, conv6.flops(B, H, W) + conv7.flops(B, H, W), ev0, _ev()))
        # algorithmic bytes: x planes read, y read and written, the mirror written
        KERNEL_BYTES[kern] = KERNEL_BYTES.get(kern, 0.0) + \
            4.0 * B * Cc * ((H + 2) * (W + 2) + (3 if ycat is not None else 2) * H * W)
    return nblk


def count_nonfinite(x, counter):
    """counter (int32, 1 element, device) += NaN / Inf elements of x (no synchronisation)."""
    check(_fn('count_nonfinite', x.dtype)(_stream(), _ptr(x, x.dtype), int(x.numel()),
                                          _ptr(counter, torch.int32)), 'iiseg_count_nonfinite')


def confusion_accumulate(y, t, cm, sums, active=None):
    """cm (C*(C+1)) int64 and sums (2) float64 are accumulated in place; with `active` (B int32)
    only images whose flag is non-zero are counted."""
    B, Cc, H, W = y.shape
    if tuple(t.shape) != (B, Cc + 1, H, W):
        raise RuntimeError('target must be one-hot (B,C+1,H,W) with void last, got %s'
                           % (tuple(t.shape),))
    dt = y.dtype
    if active is not None:
        check(_fn('confusion_masked', dt)(_stream(), _ptr(y, dt), _ptr(t, dt),
                                          _ptr(active, torch.int32), _ptr(cm, torch.int64),
                                          _ptr(sums, torch.float64), B, Cc, H * W),
              'iiseg_confusion_masked')
        return
    check(_fn('confusion', dt)(_stream(), _ptr(y, dt), _ptr(t, dt), _ptr(cm, torch.int64),
                               _ptr(sums, torch.float64), B, Cc, H * W), 'iiseg_confusion')




class Conv1x1C8:
    """A 1x1 layer on a bf16 C8 tensor (csrc/conv1x1_c8.hip; include/iiseg.h iiseg_conv1x1_c8): FC-DenseNet's
    TransitionDown (`bn` + `pool=True`: BN -> ReLU -> 1x1 conv -> 2x2 max-pool into a slice of the next
    stack) and its 1x1 class-score layer (`pool=False`: fp32 NCHW).  W (Cout, Cin, 1, 1), b (Cout) or None."""

    def __init__(self, W, b, device='cuda'):
        self.lib = _lib.load()
        self.W = torch.as_tensor(W).to(torch.float32).contiguous().to(device)
        self.b = None if b is None else torch.as_tensor(b).to(torch.float32).contiguous().to(device)
        if self.W.dim() != 4 or tuple(self.W.shape[2:]) != (1, 1):
            raise ValueError('Conv1x1C8 takes (Cout, Cin, 1, 1) filters')
        self.Cout, self.Cin = int(self.W.shape[0]), int(self.W.shape[1])
        self._wp = {}

    def flops(self, B, H, W):
        return 2.0 * self.Cin * self.Cout * H * W * B

    def __call__(self, x8, in_c=None, bn=None, pool=False, out=None, out_c0=0):
        in_ctot = c8_ctot(x8)            # (a contiguous C8 tensor or a channel slice of one)
        B, C8n, H, W, _ = x8.shape
        cin = int(in_c) if in_c is not None else C8n * 8
        if cin % 16 or cin > C8n * 8 or cin < self.Cin or cin - self.Cin >= 16:
            raise RuntimeError('Conv1x1C8: %d input channels of %d (layer: %d)' % (cin, C8n * 8, self.Cin))
        wp = self._wp.get(cin)
        if wp is None:
            n = self.lib.iiseg_conv1x1_c8_weight_bytes(self.Cout, cin)
            wp = self._wp[cin] = torch.empty(int(n) // 2, dtype=torch.bfloat16, device=self.W.device)
            check(self.lib.iiseg_conv1x1_c8_pack(_stream(), _ptr(self.W), self.Cin, 1, self.Cout, self.Cin, cin,
                                                 _ptr(wp, torch.bfloat16)), 'iiseg_conv1x1_c8_pack')
        if pool:
            if out is None:
                out = torch.zeros((B, c8_chunks(self.Cout), H // 2, W // 2, 8), dtype=torch.bfloat16,
                                  device=x8.device)
                out_c0 = 0
            if not is_c8(out) or out.shape[0] != B or tuple(out.shape[2:4]) != (H // 2, W // 2) or \
                    out_c0 + self.Cout > out.shape[1] * 8:
                raise RuntimeError('Conv1x1C8 pool target %s' % (tuple(out.shape),))
            octot = c8_ctot(out)
        else:
            if out is None:
                out = torch.empty((B, self.Cout, H, W), dtype=torch.float32, device=x8.device)
            if tuple(out.shape) != (B, self.Cout, H, W) or out.dtype != torch.float32 or not out.is_contiguous():
                raise RuntimeError('Conv1x1C8 target %s' % (tuple(out.shape),))
            octot = 0
        a, b_ = bn if bn is not None else (None, None)
        if a is not None and (a.numel() < cin or b_.numel() < cin):
            raise RuntimeError('Conv1x1C8: folded BN vectors shorter than the input')
        prof = CONV_PROFILE
        ev0 = _ev() if prof is not None else None
        check(self.lib.iiseg_conv1x1_c8(_stream(), C.c_void_p(x8.data_ptr()), B, cin, in_ctot, H, W, _ptr(a),
                                        _ptr(b_), _ptr(wp, torch.bfloat16), _ptr(self.b), self.Cout,
                                        1 if pool else 0, C.c_void_p(out.data_ptr()), octot, int(out_c0)),
              'iiseg_conv1x1_c8')
        if prof is not None:
            prof.append(('conv1x1_c8_kernel', self.flops(B, H, W), ev0, _ev()))
        return out


def bn_fold(beta, gamma, mean, inv_std, n, a=None, b=None, cap=None):
    """(a, b) with max(a x + b, 0) == relu(BatchNorm(x)) for the first n channels (float32 vectors of
    `cap` >= n entries, zero beyond n): the input-side BN + ReLU of the 16-channel C8 layers."""
    cap = int(cap or n)
    if a is None:
        a = torch.zeros(cap, dtype=torch.float32, device=beta.device)
        b = torch.zeros(cap, dtype=torch.float32, device=beta.device)
    check(_lib.load().iiseg_bn_fold_f32(_stream(), _ptr(beta), _ptr(gamma), _ptr(mean), _ptr(inv_std),
                                        _ptr(a), _ptr(b), int(n)), 'iiseg_bn_fold_f32')
    return a, b


def bn_stats_c8(buf8, c0, n, mean, inv_std, eps=1e-4):
    """Batch statistics (P10) of channels [c0, c0 + n) of the C8 tensor `buf8` into mean[c0:c0+n],
    inv_std[c0:c0+n] (float32 vectors)."""
    lib = _lib.load()
    B, C8n, H, W, _ = buf8.shape
    ctot = c8_ctot(buf8)
    key = _ws_key(buf8.device)
    need = lib.iiseg_bn_stats_c8_workspace_elems(int(n))
    ws = _bn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _bn_ws[key] = torch.empty(int(need), dtype=torch.float64, device=buf8.device)
    check(lib.iiseg_bn_stats_c8(_stream(), _c8ptr(buf8), B, ctot, int(c0), int(n), H, W,
                                float(eps), _ptr(mean), _ptr(inv_std), _ptr(ws, torch.float64)),
          'iiseg_bn_stats_c8')


def bn_stats(buf, c0, n, mean, inv_std, eps=1e-4):
    """Batch statistics (P10) of channels [c0, c0+n) of `buf` (B, Ctot, H, W) into
    mean[c0:c0+n], inv_std[c0:c0+n] (1-D tensors of length >= c0+n)."""
    B, Ctot, H, W = buf.shape
    dt = buf.dtype
    item = buf.element_size()
    _ptr(buf, dt), _ptr(mean, dt), _ptr(inv_std, dt)
    xp = C.c_void_p(buf.data_ptr() + c0 * H * W * item)
    ws = _bn_ws.get(_ws_key(buf.device))
    need = _lib.load().iiseg_bn_stats_workspace_elems(n)
    if ws is None or ws.numel() < need:
        ws = _bn_ws[_ws_key(buf.device)] = torch.empty(int(need), dtype=torch.float64, device=buf.device)
    check(_fn('bn_stats', dt)(_stream(), xp, Ctot * H * W, B, n, H * W, float(eps),
                              C.c_void_p(mean.data_ptr() + c0 * item),
                              C.c_void_p(inv_std.data_ptr() + c0 * item),
                              _ptr(ws, torch.float64)), 'iiseg_bn_stats')


def bn_relu(buf, n, beta, gamma, mean, inv_std, out=None):
    """relu((x - mean) * (gamma * inv_std) + beta) of the first n channels of `buf`
    (B, Ctot, H, W) -> dense (B, n, H, W)."""
    B, Ctot, H, W = buf.shape
    dt = buf.dtype
    if out is None:
        out = torch.empty((B, n, H, W), dtype=dt, device=buf.device)
    check(_fn('bn_relu', dt)(_stream(), _ptr(buf, dt), Ctot * H * W, B, n, H * W, _ptr(beta, dt),
                             _ptr(gamma, dt), _ptr(mean, dt), _ptr(inv_std, dt), _ptr(out, dt)),
          'iiseg_bn_relu')
    return out


def bn_affine(x, bnp, window=None):
    """In place x = (x - mean) * (gamma * inv_std) + beta per channel (stored-average BatchNorm),
    on the window (y0, x0, h, w) of x's planes or everywhere.  bnp = (beta, gamma, mean, inv_std)."""
    B, Cc, H, W = x.shape
    dt = x.dtype
    y0, x0, wh, ww = window if window is not None else (0, 0, H, W)
    if wh <= 0 or ww <= 0:
        return x
    beta, gamma, mean, inv_std = bnp
    check(_fn('bn_affine_window', dt)(_stream(), _ptr(x, dt), B, Cc, H, W, y0, x0, wh, ww,
                                      _ptr(beta, dt), _ptr(gamma, dt), _ptr(mean, dt),
                                      _ptr(inv_std, dt)), 'iiseg_bn_affine_window')
    return x


def add_noise(x, eps, sigma):
    """x + sigma * eps (GaussianNoiseLayer with the caller's standard-normal sample eps)."""
    dt = x.dtype
    out = torch.empty_like(x)
    check(_fn('add_noise', dt)(_stream(), _ptr(x, dt), _ptr(eps, dt), float(sigma), _ptr(out, dt),
                               x.numel()), 'iiseg_add_noise')
    return out


def dropout_apply(x, keep, p):
    """In place x * keep / (1 - p) (DropoutLayer, rescale=True, with the caller's 0/1 mask)."""
    dt = x.dtype
    check(_fn('dropout_apply', dt)(_stream(), _ptr(x, dt), _ptr(keep, dt), float(p), x.numel()),
          'iiseg_dropout_apply')
    return x
