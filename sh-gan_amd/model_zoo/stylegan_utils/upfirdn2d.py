"""FIR resampling ops on the HIP kernels -- same Python surface as the reference's
lib/model_zoo/stylegan_utils/upfirdn2d.py (``setup_filter`` :66, ``upfirdn2d`` :198,
``filter2d`` :245, ``upsample2d`` :279, ``downsample2d`` :316).

Differences by design: the native library is prebuilt and mandatory (no `_upfirdn2d_ref`
fallback, upfirdn2d.py:237-239).  Differentiable in ``x`` to any order: the backward is the same operator with up and
down exchanged (upfirdn2d.py:174-192)."""
import numpy as np
import torch

from . import custom_ops, misc

_plugin = None


def _init():
    """Load the native plugin (upfirdn2d.py:18-27).  Unlike the reference, failure raises."""
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin('upfirdn2d_plugin')
    return True


def _as_pair(v, what):
    if isinstance(v, int):
        v = [v, v]
    if not (isinstance(v, (list, tuple)) and len(v) == 2 and all(isinstance(e, int) for e in v)):
        raise AssertionError(f'{what} must be an int or a pair of ints')
    vx, vy = v
    if vx < 1 or vy < 1:
        raise AssertionError(f'{what} must be >= 1')
    return vx, vy


def _parse_scaling(scaling):
    return _as_pair(scaling, 'scaling factor')


def _parse_padding(padding):
    """int | [x, y] | [x0, x1, y0, y1]  ->  (padx0, padx1, pady0, pady1)."""
    if isinstance(padding, int):
        padding = [padding] * 4
    if not (isinstance(padding, (list, tuple)) and all(isinstance(e, int) for e in padding)):
        raise AssertionError('padding must be an int or a list of ints')
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    if len(padding) != 4:
        raise AssertionError('padding must have 1, 2 or 4 entries')
    return tuple(padding)


def _get_filter_size(f):
    """-> (fw, fh); None is the 1x1 identity."""
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2)):
        raise AssertionError('f must be a 1-D or 2-D tensor')
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    misc.assert_shape(f, [fh, fw][:f.ndim])
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Taps -> FIR tensor: a 1-D list shorter than 8 becomes its outer product; optionally
    normalised to unit DC gain, flipped, scaled by ``gain`` (2-D) or sqrt(gain) (1-D)."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32).clone()
    if f.ndim not in (0, 1, 2) or f.numel() == 0:
        raise AssertionError('filter must be a non-empty scalar, vector or matrix')
    f = f.reshape(1) if f.ndim == 0 else f
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if f.ndim != (1 if separable else 2):
        raise AssertionError('separable filters must be 1-D')
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return (f * (gain ** (f.ndim / 2))).to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, zero-insert upsample, FIR filter (true convolution unless ``flip_filter``), decimate.

    x: [N,C,H,W] float32 on the HIP device; f: [fh,fw], [taps] (separable) or None (identity).
    ``impl`` is accepted for signature compatibility ('ref' and 'cuda' both run the HIP kernel)."""
    if not isinstance(x, torch.Tensor) or x.ndim != 4:
        raise AssertionError('x must be a rank-4 tensor')
    if impl not in ('ref', 'cuda'):
        raise AssertionError("impl must be 'ref' or 'cuda'")
    _init()
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32):
        raise AssertionError('f must be a float32 tensor of rank 1 or 2')
    f = f.to(x.device)
    if torch.is_grad_enabled() and x.requires_grad:
        return _UpfirdnFn.apply(x, f, (upx, upy), (downx, downy), (px0, px1, py0, py1), bool(flip_filter), float(gain))
    if f.ndim == 2:
        return _plugin.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    # separable: one horizontal and one vertical pass, sqrt(gain) each (upfirdn2d.py:164-168)
    g = float(np.sqrt(gain))
    y = _plugin.upfirdn2d(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, g)
    return _plugin.upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, g)


class _UpfirdnFn(torch.autograd.Function):
    """upfirdn2d under autograd (the reference's Upfirdn2dCuda, upfirdn2d.py:141-192)."""

    @staticmethod
    def forward(ctx, x, f, up, down, padding, flip_filter, gain):
        ctx.save_for_backward(f)
        ctx.cfg = (tuple(x.shape), up, down, padding, flip_filter, gain)
        with torch.no_grad():
            return upfirdn2d(x.detach(), f, up=list(up), down=list(down), padding=list(padding), flip_filter=flip_filter, gain=gain)

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        x_shape, up, down, padding, flip_filter, gain = ctx.cfg
        # (the public operator again: differentiable when dy carries a graph -- second derivatives)
        # (dense in the layout of its dtype: NCHW for float32, channels_last for float16 -- a plain .contiguous() would transpose a half
        # tensor to NCHW and the kernel wrapper would transpose it back: two 1.3 ms copies per 512^2 layer)
        dy = dy.contiguous(memory_format=torch.channels_last) if dy.dtype == torch.float16 else dy.contiguous()
        dx = upfirdn2d_backward(dy, f, x_shape, up=list(up), down=list(down), padding=list(padding),
                                flip_filter=flip_filter, gain=gain)
        return dx, None, None, None, None, None, None


def upfirdn2d_backward(dy, f, x_shape, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Gradient of ``upfirdn2d`` with respect to its input (training row N3): the same operator with up and down swapped,
    the filter flipped and the padding of upfirdn2d.py:174-192 -- dy [N,C,OH,OW] -> dx of shape ``x_shape``."""
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, _, py0, _ = _parse_padding(padding)
    _, _, ih, iw = x_shape
    _, _, oh, ow = dy.shape
    fw, fh = _get_filter_size(f)
    p = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
    return upfirdn2d(dy, f, up=[downx, downy], down=[upx, upy], padding=p, flip_filter=(not flip_filter), gain=gain)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR-filter keeping the spatial size (zero boundary); extra ``padding`` grows/crops it."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pad = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=pad, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by ``up``: output is exactly ``up`` times the input; gain is scaled by upx*upy."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pad = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=pad, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by ``down``: output is the input size divided by ``down``."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    pad = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=pad, flip_filter=flip_filter, gain=gain, impl=impl)
