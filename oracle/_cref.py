"""Builds (gcc) and binds oracle/conv_ref.c.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'conv_ref.c')
LIB = os.path.join(HERE, 'libconv_ref.so')
# -ffp-contract=off: separate IEEE multiply and add everywhere (vector body and scalar tails
# round identically); -mavx2 only (no FMA), runs on any host the GPU boxes use.
CFLAGS = ['-O3', '-mavx2', '-ffp-contract=off', '-fopenmp', '-shared', '-fPIC']

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        r = subprocess.run(['gcc'] + CFLAGS + ['-o', LIB, SRC], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('gcc failed for oracle/conv_ref.c:\n' + r.stderr)
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.iio_conv2d_f64.restype = C.c_int
        lib.iio_conv2d_f64.argtypes = [C.c_void_p] * 4 + [C.c_int] * 10
        _lib = lib
    return _lib


def conv2d_f64(x, W, b, pad, dilation, relu):
    x = np.ascontiguousarray(x, dtype=np.float64)
    W = np.ascontiguousarray(W, dtype=np.float64)
    B, Cc, H, Wd = x.shape
    O, Ci, kh, kw = W.shape
    assert Ci == Cc, (Ci, Cc)
    OH = H + 2 * pad - dilation * (kh - 1)
    OW = Wd + 2 * pad - dilation * (kw - 1)
    assert OH > 0 and OW > 0
    out = np.empty((B, O, OH, OW), dtype=np.float64)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float64)
        bp = b.ctypes.data
    st = load().iio_conv2d_f64(x.ctypes.data, W.ctypes.data, bp, out.ctypes.data, B, Cc, H, Wd,
                               O, kh, kw, int(pad), int(dilation), int(bool(relu)))
    if st != 0:
        raise MemoryError('iio_conv2d_f64 failed')
    return out
