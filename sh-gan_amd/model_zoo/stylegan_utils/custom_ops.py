"""Plugin loader (interface of lib/model_zoo/stylegan_utils/custom_ops.py:46 ``get_plugin``).

The reference JIT-compiles its CUDA plugin with nvcc/ninja and caches it by md5 digest; here the
HIP library is prebuilt (sh-gan_amd/build.py) and ``get_plugin`` just returns a handle exposing the
same single entry point the pybind module had (upfirdn2d.cpp:98-101)."""
import torch

from ... import _lib, kernels, kernels_f16

_plugins = {}


class _Upfirdn2dPlugin:
    """``plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)``"""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        # upfirdn2d.cpp:38-59: any strides, AT_DISPATCH_FLOATING_TYPES_AND_HALF.  The two dense layouts of the networks (float32 NCHW, float16
        # channels_last) take the streaming kernels; float64, and a dense tensor in the OTHER layout of its dtype (float32 channels_last,
        # float16 NCHW), are served in place by the strided kernel with y in x's memory format (:37), not converted.
        if isinstance(x, torch.Tensor) and x.is_cuda and x.ndim == 4 and isinstance(f, torch.Tensor) and f.ndim == 2:
            cl = x.is_contiguous(memory_format=torch.channels_last)
            if (x.dtype == torch.float64 or (x.dtype == torch.float32 and cl and not x.is_contiguous())
                    or (x.dtype == torch.float16 and x.is_contiguous() and not cl)):
                return kernels.upfirdn2d_strided(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
        if x.dtype == torch.float16:         # halves, NHWC (torch.channels_last)
            return kernels_f16.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
        return kernels.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)


def get_plugin(module_name, sources=None, **build_kwargs):
    """Return the named plugin; raises if libshgan_hip.so is not built (no silent fallback)."""
    if module_name not in _plugins:
        _lib.get_lib()
        if module_name != 'upfirdn2d_plugin':
            raise RuntimeError(f'unknown plugin {module_name!r}: libshgan_hip provides upfirdn2d_plugin')
        _plugins[module_name] = _Upfirdn2dPlugin()
    return _plugins[module_name]
