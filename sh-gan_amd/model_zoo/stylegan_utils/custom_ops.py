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
        if x.dtype == torch.float16:         # upfirdn2d.cpp:59 AT_DISPATCH_FLOATING_TYPES_AND_HALF: halves, NHWC (torch.channels_last)
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
