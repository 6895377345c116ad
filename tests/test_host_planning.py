"""CPU: the planning rules of the library that decide HOW a layer is summed (split factors, kernel forms) are
functions of the layer geometry alone -- never of the batch -- so an image gets the same result alone and in a batch,
and a loop-invariant border equals its recomputed window.  Host-side C functions only (no GPU, no launches)."""
import ctypes as C

import pytest


@pytest.fixture(scope='module')
def lib(built_lib):
    from iterative_inference_segm_amd import _lib
    return _lib.load()


def desc(B, Cin, H, W, Cout, k=3, pad=1, dil=1, window=None):
    from iterative_inference_segm_amd._lib import ConvDesc
    d = ConvDesc()
    d.B, d.C1, d.C2, d.H, d.W = B, Cin, 0, H, W
    d.Cout, d.KH, d.KW, d.pad, d.dil = Cout, k, k, pad, dil
    fh, fw = H + 2 * pad - dil * (k - 1), W + 2 * pad - dil * (k - 1)
    d.oy0, d.ox0, d.OH, d.OW = window if window is not None else (0, 0, fh, fw)
    return d


def test_m16_split_k_factor_does_not_depend_on_the_batch(lib):
    """iiseg_conv_c8_m16_workspace_bytes = tiles x (256 bytes of statistics + S slabs of 32 KiB): the slab count per
    tile (the split factor, i.e. the association of the channel sum) is the same at every batch size."""
    slab = 8 * 256 * 4 * 4
    for (hw, cin) in ((7, 656), (14, 464), (28, 304), (56, 192), (224, 48)):
        per_tile = []
        for B in (1, 2, 32, 64):
            n = lib.iiseg_conv_c8_m16_workspace_bytes(C.byref(desc(B, cin, hw, hw, 16)))
            tiles = B * (-(-hw * hw // 512) if hw * hw > 512 else 1)
            assert n % B == 0
            per_tile.append(n // B)
        assert len(set(per_tile)) == 1, (hw, cin, per_tile)
        s_factor, stat = divmod(per_tile[0], slab)
        assert stat == 256 * (1 if hw * hw <= 512 else -(-hw * hw // 512)) or hw * hw > 512
        if hw <= 14:
            assert s_factor == min(16, cin // 16 // 4)  # one tile per image: 16 slices, at least four k-tiles each
        if hw >= 112:
            assert per_tile[0] < slab                  # many tiles per image: statistics only, no slabs


def test_gemm_f64_split_factor_does_not_depend_on_the_batch(lib):
    """fc6 / fc7 / score_fr as float64 split-K GEMMs: workspace = Tpad x (Kpad + S Mpad) doubles; S from the layer."""
    for (cin, cout) in ((25088, 4096), (4096, 4096), (4096, 11)):
        S = set()
        for B in (1, 10, 32, 64):
            d = desc(B, cin, 7, 7, cout, k=1, pad=0)
            assert lib.iiseg_conv_plan_f64(C.byref(d)) == 0
            assert lib.iiseg_conv_gemm_f64_supported(C.byref(d)) == 1
            n = lib.iiseg_conv_gemm_f64_workspace_elems(C.byref(d))
            tpad = -(-B * 49 // 128) * 128
            assert n % tpad == 0
            s, r = divmod(n // tpad - d.Kpad, d.Mpad)
            assert r == 0 and s >= 1
            S.add(s)
        assert len(S) == 1, (cin, cout, S)


def test_kernel_form_queries_follow_the_layer_not_the_window(lib):
    """Which kernel a float64 / small-channel layer runs on is the same for a full-map launch and a window of it."""
    for window in (None, (4, 6, 20, 30)):
        d = desc(2, 64, 60, 70, 128, window=window)
        assert lib.iiseg_conv_plan_f64(C.byref(d)) == 0
        assert lib.iiseg_conv_halo_f64_supported(C.byref(d)) == 1
        d = desc(2, 11, 60, 70, 11, pad=0, dil=4, window=window)
        assert lib.iiseg_conv_plan(C.byref(d)) == 0
        assert lib.iiseg_conv_small_supported(C.byref(d)) == 1
    d = desc(2, 11, 60, 70, 11, pad=1)
    assert lib.iiseg_conv_plan(C.byref(d)) == 0
    assert lib.iiseg_conv_small_supported(C.byref(d)) == 0     # zero-padded: the halo kernel
    d = desc(2, 64, 60, 70, 128, dil=2)
    assert lib.iiseg_conv_plan_f64(C.byref(d)) == 0
    assert lib.iiseg_conv_halo_f64_supported(C.byref(d)) == 0   # dilated float64: the static-tap kernel


def test_deconv_output_phase_form_is_a_function_of_the_layer(lib):
    """iiseg_deconv_phase_supported: K = 2 stride at stride 2 / 8 with at most 16 channels on either side (the three
    FCN-8 upsamplers, models/fcn8.py:90,100,109) -- whatever the batch, the map or the window, so a window and the
    full map of a layer come from the same kernel; a skip tensor only at stride 2; packed-weight size
    [stride][2][2][Cin][stride][Cout padded to 12 / 16]."""
    from iterative_inference_segm_amd._lib import DeconvDesc

    def dd(B, Cin, H, W, Cout, K, s, window=None):
        d = DeconvDesc()
        d.B, d.Cin, d.H, d.W, d.Cout, d.K, d.stride = B, Cin, H, W, Cout, K, s
        fh, fw = (H - 1) * s + K, (W - 1) * s + K
        d.oy0, d.ox0, d.OH, d.OW = window if window is not None else (0, 0, fh, fw)
        return d

    for B in (1, 10, 64):
        for hw in (3, 28, 60):
            for win in (None, (1, 2, 5, 7)):
                assert lib.iiseg_deconv_phase_supported(C.byref(dd(B, 11, hw, hw, 11, 16, 8, win)), 0, 0) == 1
                assert lib.iiseg_deconv_phase_supported(C.byref(dd(B, 11, hw, hw, 11, 16, 8, win)), 0, 1) == 1
                assert lib.iiseg_deconv_phase_supported(C.byref(dd(B, 11, hw, hw, 11, 16, 8, win)), 1, 0) == 0
                assert lib.iiseg_deconv_phase_supported(C.byref(dd(B, 11, hw, hw, 11, 4, 2, win)), 1, 0) == 1
    assert lib.iiseg_deconv_phase_supported(C.byref(dd(2, 11, 8, 8, 11, 3, 2)), 0, 0) == 0      # K != 2 stride
    assert lib.iiseg_deconv_phase_supported(C.byref(dd(2, 11, 8, 8, 11, 8, 4)), 0, 0) == 0      # stride 4: no form
    assert lib.iiseg_deconv_phase_supported(C.byref(dd(2, 21, 8, 8, 21, 4, 2)), 0, 0) == 0      # 21 classes: gather
    assert lib.iiseg_deconv_phase_weight_elems(C.byref(dd(2, 11, 8, 8, 11, 16, 8))) == 8 * 4 * 11 * 8 * 12
    assert lib.iiseg_deconv_phase_weight_elems(C.byref(dd(2, 5, 8, 8, 16, 4, 2))) == 2 * 4 * 5 * 2 * 16
    assert lib.iiseg_deconv_phase_weight_elems(C.byref(dd(2, 11, 8, 8, 11, 3, 2))) == 0


def test_relaxed_nan_flag_stays_off_the_sources_that_use_nan():
    """build.EXTRA_FLAGS: -fno-honor-nans only on sources whose kernels never use NaN as a value -- the float64
    Winograd transforms mark 'outside the pooled map' with a NaN that must never compare equal."""
    import os
    from iterative_inference_segm_amd import build
    for src, flags in build.EXTRA_FLAGS.items():
        assert src in build.SOURCES
        if '-fno-honor-nans' in flags:
            text = open(os.path.join(build.CSRC, src)).read()
            assert '__builtin_nan' not in text and 'isnan' not in text and 'NAN' not in text, src
    assert 'conv_wino_f64.hip' not in build.EXTRA_FLAGS
