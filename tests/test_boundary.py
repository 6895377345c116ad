"""CPU: the drop-in boundary.  The C-ABI library builds for gfx950, loads, and exports exactly
the symbols include/iiseg.h declares; the product never touches the oracle; a missing library
fails loudly.  No compute calls here (no GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'iterative_inference_segm_amd')


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'iiseg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(iiseg_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_entry_points():
    names = header_functions()
    for required in ['iiseg_conv_f32', 'iiseg_conv_pack_f32', 'iiseg_maxpool2x2_f32',
                     'iiseg_unpool_eqmask_f32', 'iiseg_deconv_f32', 'iiseg_crop_softmax_f32',
                     'iiseg_refine_update_f32', 'iiseg_refine_finalize', 'iiseg_confusion_f32',
                     'iiseg_strerror', 'iiseg_abi_version']:
        assert required in names


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in header_functions():
        assert hasattr(lib, name), 'libiiseg_hip.so does not export %s' % name
    from iterative_inference_segm_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()      # binding covers the whole header
    loaded = _lib.load()
    assert loaded.iiseg_abi_version() == _lib.ABI_VERSION
    assert loaded.iiseg_target_arch() == b'gfx950'
    assert b'shape' in loaded.iiseg_strerror(-2)


def test_binding_argument_counts_match_the_header():
    """Every ctypes signature has exactly as many arguments as the header's prototype (a wrong
    count only shows up as a TypeError at the first GPU call otherwise)."""
    from iterative_inference_segm_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'iiseg.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = dict(re.findall(r'\b(iiseg_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', text))
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, args in protos.items():
        args = args.strip()
        n = 0 if args in ('', 'void') else len(args.split(','))
        assert n == len(_lib.SIGNATURES[name][1]), '%s: header %d arguments, binding %d' % (
            name, n, len(_lib.SIGNATURES[name][1]))


def test_library_contains_gfx950_code_object(built_lib):
    blob = open(built_lib, 'rb').read()
    assert b'gfx950' in blob and b'conv_taps_f32_kernel' in blob and b'conv_igemm_f32_kernel' in blob
    for other in (b'gfx90a', b'gfx942', b'sm_80'):
        assert other not in blob


def test_argument_validation_without_a_gpu(built_lib):
    """Status codes for bad descriptors are produced before any launch (host-side checks)."""
    from iterative_inference_segm_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.iiseg_conv_plan(ctypes.byref(d)) == -2                    # IISEG_ERR_SHAPE
    d.B, d.C1, d.H, d.W, d.Cout, d.KH, d.KW, d.pad, d.dil = 1, 3, 8, 8, 11, 3, 3, 1, 1
    d.OH, d.OW = 8, 8
    assert lib.iiseg_conv_plan(ctypes.byref(d)) == 0
    assert (d.Kpad, d.Mpad) == (36, 32)                                  # 2 x 2 channels x 9 taps
    assert lib.iiseg_conv_ktab_entries(ctypes.byref(d)) == 36
    assert lib.iiseg_conv_f32(None, ctypes.byref(d), *([None] * 9)) == -1   # IISEG_ERR_NULL
    d.OH = 9                                                             # window outside output
    assert lib.iiseg_conv_f32(None, ctypes.byref(d), *([None] * 9)) == -2
    d7 = _lib.ConvDesc()
    d7.B, d7.C1, d7.H, d7.W, d7.Cout, d7.KH, d7.KW, d7.dil, d7.OH, d7.OW = 1, 5, 9, 9, 40, 7, 7, 1, 3, 3
    assert lib.iiseg_conv_plan(ctypes.byref(d7)) == 0
    assert (d7.Kpad, d7.Mpad) == (256, 64)                               # 245 -> 16-multiple
    with pytest.raises(RuntimeError, match='shape'):
        _lib.check(-2, 'x')


def test_product_never_imports_the_oracle():
    offenders = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
                    offenders.append(f)
    for f in ('iterative_inference.py',):
        path = os.path.join(ROOT, f)
        if os.path.exists(path) and re.search(r'^\s*(from|import)\s+oracle\b', open(path).read(),
                                              flags=re.M):
            offenders.append(f)
    assert not offenders, 'product files import the oracle: %s' % offenders


def test_missing_library_fails_loudly(tmp_path):
    code = ('import iterative_inference_segm_amd._lib as L\n'
            'L.LIB_PATH = %r\n'
            'try:\n    L.load()\nexcept RuntimeError as e:\n    print("RAISED", e)\n'
            % str(tmp_path / 'nope.so'))
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True)
    assert 'RAISED' in out.stdout and 'no CPU fallback' in out.stdout


def test_ops_refuse_host_tensors(built_lib):
    import torch
    from iterative_inference_segm_amd import ops
    with pytest.raises(RuntimeError, match='device tensors'):
        ops.maxpool2x2(torch.zeros(1, 1, 4, 4))
