"""ctypes binding of libiiseg_hip.so (the C ABI of include/iiseg.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing a
compute path raises.  The product never routes through the CPU oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libiiseg_hip.so')

ABI_VERSION = 31

CONV_RELU = 1
CONV_UNPOOL = 2
CONV_X3 = 8
CONV_TRANSPOSED2 = 4
CONV_ZINS = 16


class ConvDesc(C.Structure):
    """struct iiseg_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in
                ('B', 'C1', 'C2', 'H', 'W', 'Cout', 'KH', 'KW', 'pad', 'dil',
                 'oy0', 'ox0', 'OH', 'OW', 'AH', 'AW', 'ay0', 'ax0')] + \
               [('flags', C.c_uint32), ('Kpad', C.c_int32), ('Mpad', C.c_int32),
                ('out_ctot', C.c_int32), ('out_c0', C.c_int32),
                ('out_H', C.c_int32), ('out_W', C.c_int32), ('out_y0', C.c_int32),
                ('out_x0', C.c_int32), ('tile_y0', C.c_int32), ('tile_x0', C.c_int32)]


class DeconvDesc(C.Structure):
    """struct iiseg_deconv_desc"""
    _fields_ = [(n, C.c_int32) for n in
                ('B', 'Cin', 'H', 'W', 'Cout', 'K', 'stride', 'oy0', 'ox0', 'OH', 'OW',
                 'AH', 'AW', 'ay0', 'ax0')]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/iiseg.h one to one
SIGNATURES = {
    'iiseg_strerror': (C.c_char_p, [C.c_int]),
    'iiseg_last_hip_error': (C.c_char_p, []),
    'iiseg_abi_version': (C.c_int, []),
    'iiseg_profile_begin': (C.c_int, [C.c_int]),
    'iiseg_profile_count': (C.c_int, []),
    'iiseg_profile_end': (C.c_int, [C.POINTER(C.c_float), C.c_int]),
    'iiseg_target_arch': (C.c_char_p, []),
    'iiseg_conv_ktab_entries': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_plan': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pack_f32': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp, _vp]),
    'iiseg_conv_f32': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 9),
    'iiseg_conv_small_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pool_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pool_f32': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 10),
    'iiseg_conv_mask_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_mask_f32': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 12),
    'iiseg_conv_bnrelu_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_bnrelu_f32': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64] + [_vp] * 8),
    'iiseg_conv_wino_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_weight_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_workspace_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_pack_f32': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_wino_f32': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 9 + [C.c_uint32]),
    'iiseg_conv_wino_bf16_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_bf16_weight_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_bf16_workspace_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_bf16_pack': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_wino_bf16': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 9 + [C.c_uint32]),
    'iiseg_conv_gemm_bf16_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_bf16_weight_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_bf16_workspace_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_bf16_pack': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_gemm_bf16': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 5),
    'iiseg_conv_halo_bf16_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_halo_bf16_weight_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_halo_bf16_pack': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_halo_bf16': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 11),
    'iiseg_conv_wino_f64_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_f64_weight_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_f64_workspace_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_wino_pack_f64': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_wino_f64': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 9),
    'iiseg_conv_c8_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_c8_is_flat': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_c8_force_tiling': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'iiseg_conv_c8_tiling': (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(C.c_int32)]),
    'iiseg_conv_c8': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 6 + [_i32, _vp, _i32, _vp, _vp]),
    'iiseg_conv_c8_slice': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i32] + [_vp] * 5 + [_i32, _vp, _i32, _vp, _vp]),
    'iiseg_unpool_c8': (C.c_int, [_vp] * 4 + [C.c_int] * 7),
    'iiseg_conv_c8_m16_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_c8_m16': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i32] + [_vp] * 6 + [_i32]),
    'iiseg_conv_c8_m16_workspace_bytes': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_c8_m16_ws': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i32] + [_vp] * 6 + [_i32, _vp, _i64, _vp, _vp, _f64] + [_vp] * 4 + [_i32]),
    'iiseg_conv1x1_c8_weight_bytes': (_i64, [C.c_int, C.c_int]),
    'iiseg_conv1x1_c8_pack': (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, C.c_int, C.c_int, _vp]),
    'iiseg_conv1x1_c8': (C.c_int, [_vp, _vp] + [C.c_int] * 5 + [_vp] * 4 + [C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    'iiseg_bn_fold_f32': (C.c_int, [_vp] * 7 + [_i32]),
    'iiseg_bn_stats_c8_workspace_elems': (C.c_int64, [_i32]),
    'iiseg_bn_stats_c8': (C.c_int, [_vp, _vp] + [_i32] * 6 + [C.c_double, _vp, _vp, _vp]),
    'iiseg_nchw_to_c8_slice': (C.c_int, [_vp, _vp, _vp] + [_i32] * 6),
    'iiseg_c8_slice_to_nchw': (C.c_int, [_vp, _vp, _vp] + [_i32] * 6),
    'iiseg_nchw_to_c8': (C.c_int, [_vp, _vp, _vp] + [_i32] * 5),
    'iiseg_c8_to_nchw': (C.c_int, [_vp, _vp, _vp] + [_i32] * 5),
    'iiseg_pool_mask_c8': (C.c_int, [_vp, _vp, _i32, _vp, _vp] + [_i32] * 11),
    'iiseg_conv_c8_split_weights': (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    'iiseg_nchw_to_c8x3': (C.c_int, [_vp, _vp, _vp] + [_i32] * 5),
    'iiseg_c8x3_to_nchw': (C.c_int, [_vp, _vp, _vp] + [_i32] * 5),
    'iiseg_pool_mask_c8x3': (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 12),
    'iiseg_conv_gemm_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_workspace_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_f32': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 5 + [C.c_uint32]),
    'iiseg_maxpool2x2_f32': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32]),
    'iiseg_unpool_eqmask_f32': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    'iiseg_maxpool2x2_window_f32': (C.c_int, [_vp] * 3 + [_i32] * 7),
    'iiseg_maxpool2x2_window_f64': (C.c_int, [_vp] * 3 + [_i32] * 7),
    'iiseg_unpool_eqmask_window_f32': (C.c_int, [_vp] * 5 + [_i32] * 7),
    'iiseg_unpool_eqmask_window_f64': (C.c_int, [_vp] * 5 + [_i32] * 7),
    'iiseg_deconv_f32': (C.c_int, [_vp, C.POINTER(DeconvDesc)] + [_vp] * 5),
    'iiseg_deconv_phase_supported': (C.c_int, [C.POINTER(DeconvDesc), C.c_int, C.c_int]),
    'iiseg_deconv_phase_weight_elems': (C.c_int64, [C.POINTER(DeconvDesc)]),
    'iiseg_deconv_phase_pack_f32': (C.c_int, [_vp, C.POINTER(DeconvDesc), _vp, _vp]),
    'iiseg_deconv_phase_pack_f64': (C.c_int, [_vp, C.POINTER(DeconvDesc), _vp, _vp]),
    'iiseg_deconv_phase_f32': (C.c_int, [_vp, C.POINTER(DeconvDesc)] + [_vp] * 5),
    'iiseg_deconv_phase_f64': (C.c_int, [_vp, C.POINTER(DeconvDesc)] + [_vp] * 5),
    'iiseg_crop_softmax_f32': (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 8),
    'iiseg_refine_partials': (C.c_int, [_i32, _i32]),
    'iiseg_refine_update_f32': (C.c_int, [_vp] * 5 + [_i32] * 8 + [_f32]),
    'iiseg_refine_update_c8_f32': (C.c_int, [_vp] * 6 + [_i32] * 9 + [_f32]),
    'iiseg_refine_finalize': (C.c_int, [_vp] * 5 + [_i32] * 3 + [_f64]),
    'iiseg_ctx_tail_partials': (C.c_int, [_i32, _i32]),
    'iiseg_ctx_tail_f32': (C.c_int, [_vp] * 3 + [_i32] + [_vp] * 2 + [_i32] + [_vp] * 5 + [_i32] * 10 + [_f32]),
    'iiseg_confusion_f32': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_confusion_masked_f32': (C.c_int, [_vp] * 6 + [_i32] * 3),
    'iiseg_confusion_masked_f64': (C.c_int, [_vp] * 6 + [_i32] * 3),
    'iiseg_bn_stats_workspace_elems': (_i64, [_i32]),
    'iiseg_bn_stats_f32': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    'iiseg_bn_relu_f32': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32] + [_vp] * 5),
    'iiseg_bn_stats_f64': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f64, _vp, _vp, _vp]),
    'iiseg_bn_relu_f64': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32] + [_vp] * 5),
    'iiseg_sqerr_softmax_bwd_f32': (C.c_int, [_vp] * 4 + [_i32] * 8),
    'iiseg_sqerr_softmax_bwd_f64': (C.c_int, [_vp] * 4 + [_i32] * 8),
    'iiseg_depool_bwd_f32': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_depool_bwd_f64': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_pool_relu_bwd_f32': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_pool_relu_bwd_f64': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_grad_update_f32': (C.c_int, [_vp] * 6 + [_i32] * 8 + [C.c_float]),
    'iiseg_grad_update_f64': (C.c_int, [_vp] * 6 + [_i32] * 8 + [C.c_double]),
    'iiseg_add_noise_f32': (C.c_int, [_vp, _vp, _vp, C.c_float, _vp, _i64]),
    'iiseg_add_noise_f64': (C.c_int, [_vp, _vp, _vp, C.c_double, _vp, _i64]),
    'iiseg_dropout_apply_f32': (C.c_int, [_vp, _vp, _vp, C.c_float, _i64]),
    'iiseg_dropout_apply_f64': (C.c_int, [_vp, _vp, _vp, C.c_double, _i64]),
    'iiseg_bn_affine_window_f32': (C.c_int, [_vp, _vp] + [_i32] * 8 + [_vp] * 4),
    'iiseg_bn_affine_window_f64': (C.c_int, [_vp, _vp] + [_i32] * 8 + [_vp] * 4),
    # float64 (strict-parity) variants
    'iiseg_conv_plan_f64': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pack_f64': (C.c_int, [_vp, C.POINTER(ConvDesc), _vp, _i64, _i64, _vp]),
    'iiseg_conv_f64': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 8),
    'iiseg_conv_halo_f64_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pool_f64_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_pool_f64': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 9),
    'iiseg_conv_mask_f64_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_mask_f64': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 11),
    'iiseg_conv_gemm_f64_supported': (C.c_int, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_f64_workspace_elems': (_i64, [C.POINTER(ConvDesc)]),
    'iiseg_conv_gemm_f64': (C.c_int, [_vp, C.POINTER(ConvDesc)] + [_vp] * 5),
    'iiseg_im2col_f64': (C.c_int, [_vp, _vp, _vp] + [_i32] * 6),
    'iiseg_maxpool2x2_f64': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32]),
    'iiseg_unpool_eqmask_f64': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    'iiseg_deconv_f64': (C.c_int, [_vp, C.POINTER(DeconvDesc)] + [_vp] * 5),
    'iiseg_crop_softmax_f64': (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 8),
    'iiseg_refine_update_f64': (C.c_int, [_vp] * 5 + [_i32] * 8 + [_f64]),
    'iiseg_confusion_f64': (C.c_int, [_vp] * 5 + [_i32] * 3),
    'iiseg_count_nonfinite_f32': (C.c_int, [_vp, _vp, _i64, _vp]),
    'iiseg_count_nonfinite_f64': (C.c_int, [_vp, _vp, _i64, _vp]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libiiseg_hip.so is missing (%s). Build it with '
            '`python -m iterative_inference_segm_amd.build`; there is no CPU fallback.' % LIB_PATH)
    # torch first: the library shares torch's HIP runtime (streams and device pointers cross the
    # boundary).  Loaded before torch it brought up a runtime of its own, whose first launch on a GPU
    # box said "no ROCm-capable device is detected" (__graft_entry__.build() followed by smoke()).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.iiseg_abi_version() != ABI_VERSION:
        raise RuntimeError('libiiseg_hip.so ABI %d != binding ABI %d; rebuild'
                           % (lib.iiseg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().iiseg_strerror(status).decode()
        if status == -4:                       # IISEG_ERR_LAUNCH: say what the HIP runtime said
            msg += ': ' + load().iiseg_last_hip_error().decode()
        raise RuntimeError('%s failed: %s (iiseg_status %d)' % (what, msg, status))
