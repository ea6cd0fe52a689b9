"""2-D convolution with optional FIR up/down-sampling on the HIP kernels -- same call signature and
padding arithmetic as lib/model_zoo/stylegan_utils/conv2d_resample.py:57-154."""
import torch

from ... import kernels
from . import conv2d_gradfix, upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _get_weight_shape(w):
    return [int(s) for s in w.shape]


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """One convolution on the MFMA kernel.  ``flip_weight=True`` = correlation (F.conv2d),
    False = true convolution (conv2d_resample.py:26-51).  ``w`` is [O, I/groups, kh, kw]; for
    ``transpose`` it is the *forward-layout* weight (the caller does not pre-transpose it)."""
    n, c, h, wd = x.shape
    oc, icg, kh, kw = _get_weight_shape(w)
    if isinstance(padding, (list, tuple)):
        if padding[0] != padding[1]:
            raise NotImplementedError('asymmetric conv padding')
        padding = padding[0]
    if x.dtype == torch.float16 or (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)):
        # training path and every float16 tensor (conv2d_resample.py:26-51 verbatim in structure): flip for a true convolution, the transposed form takes
        # the weight as [Cin, Cout, kh, kw]
        if groups != 1:
            raise NotImplementedError('grouped convolutions are forward only')
        wg = w if flip_weight else w.flip([2, 3])
        if transpose:
            return conv2d_gradfix.conv_transpose2d(x, wg.transpose(0, 1), stride=stride, padding=padding)
        return conv2d_gradfix.conv2d(x, wg, stride=stride, padding=padding)
    xs = x.reshape(n * groups, c // groups, h, wd)
    if transpose:
        if stride != 2:
            raise NotImplementedError('transposed convolution is implemented for stride 2')
        pw = kernels.conv_weight_prep(w, flip=not flip_weight, groups=groups)
        y = kernels.conv2d(xs, pw, mode=kernels.MODE_UP2T)
        if padding:
            y = y[:, :, padding:y.shape[2] - padding, padding:y.shape[3] - padding].contiguous()
    else:
        if stride not in (1, 2):
            raise NotImplementedError('strided convolution is implemented for stride 1 and 2')
        pw = kernels.conv_weight_prep(w, flip=not flip_weight, groups=groups)
        y = kernels.conv2d(xs, pw, mode=kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2, pad=padding)
    return y.reshape(n, -1, *y.shape[2:])


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """x [N,C,H,W], w [O, C/groups, kh, kw], f from ``upfirdn2d.setup_filter`` (None = identity).
    Padding is given with respect to the upsampled image and applied once, up front."""
    if not (isinstance(x, torch.Tensor) and x.ndim == 4):
        raise AssertionError('x must be a rank-4 tensor')
    if not (isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype):
        raise AssertionError('w must be a rank-4 tensor of the same dtype as x')
    if not (f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)):
        raise AssertionError('f must be None or a float32 filter')
    if not (isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1 and isinstance(groups, int) and groups >= 1):
        raise AssertionError('up, down and groups must be positive ints')
    oc, icg, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # resampling widens the padding by the filter footprint
    if up > 1:
        px0, px1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2
        py0, py1 = py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2
        py0, py1 = py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2
    pads = [px0, px1, py0, py1]
    pointwise = kh == 1 and kw == 1

    if pointwise and down > 1 and up == 1:       # decimate first, then the cheap 1x1
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if pointwise and up > 1 and down == 1:       # 1x1 at low resolution, then interpolate
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=pads, gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                     # low-pass, then strided convolution
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                   # transposed strided convolution, then low-pass
        px0, px1, py0, py1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=w, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                            flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                                flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 == py0 == py1 and px0 >= 0:    # plain convolution, padding done by the kernel
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    # generic: explicit pad/crop through the FIR op with an identity filter, then an unpadded conv
    x = upfirdn2d.upfirdn2d(x=x, f=None, padding=pads, flip_filter=flip_filter)
    return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
