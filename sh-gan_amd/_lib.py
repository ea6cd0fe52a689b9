"""ctypes loader for libshgan_hip.so (the C-ABI of include/shgan_hip.h).

Replaces the reference's JIT plugin loader (lib/model_zoo/stylegan_utils/custom_ops.py:46-124):
the library is prebuilt in-tree by ``sh-gan_amd/build.py``; there is NO fallback -- a missing or
stale library raises, it never silently degrades to a PyTorch path (contrast upfirdn2d.py:18-27)."""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: the .so binds to torch's already-loaded HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libshgan_hip.so')
ABI_VERSION = 36

c_fp = ctypes.c_void_p      # device pointers travel as void*
c_i = ctypes.c_int
c_f = ctypes.c_float
c_l = ctypes.c_long
c_pp = ctypes.POINTER(ctypes.c_void_p)



class DenseGroup(ctypes.Structure):          # shg_dense_group
    _fields_ = [('x1', c_fp), ('x2', c_fp), ('w', c_fp), ('b', c_fp), ('y', c_fp),
                ('ld1', c_i), ('ld2', c_i), ('K1', c_i), ('K2', c_i), ('O', c_i), ('ldy', c_i),
                ('wgain', c_f), ('bgain', c_f)]


class StyleGroup(ctypes.Structure):          # shg_style_group
    _fields_ = [('styles', c_fp), ('wsq', c_fp), ('s_out', c_fp), ('dcoef', c_fp),
                ('ld', c_i), ('I', c_i), ('O', c_i), ('OP', c_i), ('demod', c_i), ('pre_gain', c_f)]


# name -> argtypes (restype is int unless noted); mirrors include/shgan_hip.h one to one
_SIGS = {
    'shg_abi_version': [],
    'shg_device_info': [c_i, ctypes.c_char_p, c_i],
    'shg_upfirdn2d_out_size': [c_i] * 12 + [ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    'shg_upfirdn2d_f32': [c_fp, c_fp, c_fp] + [c_i] * 15 + [c_f, c_fp],
    'shg_upfirdn2d_strided': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(c_l), ctypes.POINTER(c_l), c_i, c_i, c_l, c_l] + [c_i] * 9
                             + [c_f, c_fp],
    'shg_upfirdn2d_epilogue_f32': [c_fp, c_fp, c_fp] + [c_i] * 15 + [c_f, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_bias_act_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_fp],
    'shg_fma_f32': [c_fp, c_fp, c_fp, c_fp, c_l, c_fp],
    'shg_fma_bcast': [c_fp, c_fp, c_fp, c_fp, c_i] + [ctypes.POINTER(c_l)] * 4 + [c_i, c_fp],
    'shg_mul_reduce': [c_fp, c_fp, c_fp, c_i, c_i] + [ctypes.POINTER(c_l)] * 3 + [c_i, c_i, c_fp],
    'shg_scale_channels_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_fp],
    'shg_sum_partials_f32': [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_scale_cast_f32_f16': [c_fp, c_fp, c_l, c_f, c_i, c_fp],
    'shg_planes_to_image_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv_weight_prep_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_fp],
    'shg_conv2d_f32': [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_l, c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_i,
                       c_fp, ctypes.c_size_t, c_fp],
    'shg_conv2d_workspace_bytes': [c_i] * 10,
    'shg_conv_wino_chunk': [],
    'shg_conv_weight_prep_wino_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_wino_f32': [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_conv2d_wino_ws_f32': [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp, ctypes.c_size_t, c_fp],
    'shg_conv2d_wino_workspace_bytes': [c_i] * 6,
    'shg_conv_wino4_weight_elems': [c_i, c_i],
    'shg_conv_weight_prep_wino4_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_wino4_supported': [c_i] * 5,
    'shg_conv2d_wino4_f32': [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_conv2d_wino4_ws_f32': [c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp, ctypes.c_size_t, c_fp],
    'shg_conv2d_wino4_workspace_bytes': [c_i] * 6,
    'shg_upfir_planar_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_upfir_planar_sep_supported': [c_i, c_i],
    'shg_upfir_planar_sep_f32': [c_fp, ctypes.POINTER(c_f), c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_i, c_f, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_conv1x1_thin_in_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_fp],
    'shg_torgb_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    'shg_dense_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_fp],
    'shg_demod_weight_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_demod_weight_backward_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_style_factors_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_style_factors_backward_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_matmul_nn_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'shg_matmul_tn_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_fp],
    'shg_normalize_2nd_moment_f32': [c_fp, c_fp, c_i, c_i, c_f, c_fp],
    'shg_modconv_style_prep_f32': [c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'shg_dense_grouped_f32': [ctypes.POINTER(DenseGroup), c_i, c_i, c_fp],
    'shg_modconv_style_prep_grouped_f32': [ctypes.POINTER(StyleGroup), c_i, c_i, c_fp],
    'shg_shu_rfft2_shift_f32': [c_fp, c_l, c_fp, c_i, c_i, c_fp],
    'shg_bias_act_backward_f32': [c_fp, c_fp, c_fp, c_l, c_i, c_f, c_f, c_f, c_fp],
    'shg_conv2d_wgrad_workspace_bytes': [c_i] * 7,
    'shg_conv2d_wgrad_f32': [c_fp, c_fp, c_fp] + [c_i] * 11 + [c_fp, ctypes.c_size_t, c_fp],
    'shg_conv2d_wgrad_wino_supported': [c_i] * 8,
    'shg_conv2d_wgrad_wino_workspace_bytes': [c_i] * 5,
    'shg_conv2d_wgrad_wino_f32': [c_fp, c_fp, c_fp] + [c_i] * 5 + [c_fp, ctypes.c_size_t, c_fp],
    'shg_shu_spectral_f32': [c_fp] * 6 + [c_i] * 4 + [c_fp],
    'shg_shu_split_irfft2_f32': [c_fp, c_fp, c_pp, c_pp, ctypes.POINTER(c_l), c_i, c_i, c_i, c_i, c_fp],
    'shg_shu_split_adjoint_f32': [c_pp, ctypes.POINTER(c_l), c_pp, c_fp, c_i, c_i, c_fp],
    'shg_composite_u8': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_assemble_input_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_assemble_input_u8': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_conv_weight_prep_up_poly_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_up_poly_supported': [c_i, c_i, c_i, c_i, c_i],
    'shg_conv2d_up_poly_f32': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp],
    'shg_conv2d_up_poly_ws_f32': [c_fp, c_fp, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp, c_fp, ctypes.c_size_t, c_fp],
    'shg_conv2d_up_poly_workspace_bytes': [c_i] * 6,
    'shg_fir_down_planar_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'shg_fir_pad2_sep_supported': [c_i, c_i, c_i],
    'shg_fir_resample2_sep_supported': [c_i, c_i, c_i],
    'shg_fir_resample2_sep_f32': [c_fp, ctypes.POINTER(c_f), c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'shg_fir_pad2_sep_f32': [c_fp, ctypes.POINTER(c_f), c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'shg_conv_weight_prep_down_poly_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_down_poly_supported': [c_i, c_i, c_i, c_i, c_i],
    'shg_conv2d_down_poly_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_mask_raster_f32': [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_fp],
    'shg_fid_accumulate_f64': [c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_minibatch_std_f32': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_f16': [c_fp, c_fp, c_fp, c_fp] + [c_i] * 12 + [c_fp],
    'shg_conv2d_f16_needs_clear': [c_i] * 5,
    'shg_conv2d_f16_fused': [c_fp, c_fp, c_fp] + [c_i] * 12 + [c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_i, c_f, c_f, c_f, c_fp, c_fp],
    'shg_conv2d_f16_packed_weight_elems': [c_i] * 3,
    'shg_conv2d_f16_pack_weight': [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'shg_conv2d_f16_pack_weight_oihw': [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    'shg_conv2d_f16_set_routes': [c_i],
    'shg_conv2d_wgrad_f16_workspace_bytes': [c_i] * 6,
    'shg_conv2d_wgrad_f16': [c_fp, c_fp, c_fp] + [c_i] * 10 + [c_fp, ctypes.c_size_t, c_fp],
    'shg_upfirdn2d_f16': [c_fp, c_fp, c_fp] + [c_i] * 15 + [c_f, c_fp],
    'shg_relayout_f32_f16': [c_fp, c_fp, c_i, c_i, c_l, c_i, c_fp],
    'shg_bias_act_f16': [c_fp, c_fp, c_fp, c_l, c_i, c_i, c_f, c_f, c_f, c_fp],
    'shg_bias_act_backward_f16': [c_fp, c_fp, c_fp, c_l, c_i, c_f, c_f, c_f, c_fp],
    'shg_modtail_backward_f32_blocks': [c_l],
    'shg_modtail_backward_f32_cslices': [c_i, c_i, c_l],
    'shg_modtail_backward_f32': [c_fp] * 9 + [c_i, c_i, c_l, c_i, c_f, c_f, c_f, c_fp],
    'shg_modtail_f16': [c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_fp],
    'shg_modtail_backward_f16_blocks': [c_l, c_i],
    'shg_modtail_backward_f16': [c_fp] * 9 + [c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_fp],
}

_lib = None


def use_library(path):
    """Point the loader at another build of the same ABI (tools/ use the -DSHG_ABLATE timing-study library)."""
    global _lib, LIB_PATH
    _lib, LIB_PATH = None, path


class ShgError(RuntimeError):
    """Raised when a C-ABI call returns a negative status (mirrors TORCH_CHECK -> RuntimeError)."""


def get_lib():
    """Load (once) and return the ctypes handle.  Raises if the library is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run `python sh-gan_amd/build.py` '
            '(or __graft_entry__.build()). There is deliberately no CPU / PyTorch fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if an exported symbol is missing
        fn.argtypes = args
        fn.restype = c_i
    lib.shg_last_error.argtypes = []
    lib.shg_last_error.restype = ctypes.c_char_p
    lib.shg_conv2d_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_wino_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_wino4_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_up_poly_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_wgrad_wino_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_wgrad_f16_workspace_bytes.restype = ctypes.c_size_t
    lib.shg_conv2d_f16_packed_weight_elems.restype = c_l
    lib.shg_conv_wino4_weight_elems.restype = c_l
    ver = lib.shg_abi_version()
    if ver != ABI_VERSION:
        raise RuntimeError(f'libshgan_hip.so ABI {ver} != expected {ABI_VERSION}: rebuild with sh-gan_amd/build.py')
    _lib = lib
    return lib


def exported_symbols():
    return sorted(list(_SIGS.keys()) + ['shg_last_error'])


def check(rc, what=''):
    if rc != 0:
        msg = get_lib().shg_last_error().decode(errors='replace')
        raise ShgError(f'{what}: {msg} (code {rc})')
