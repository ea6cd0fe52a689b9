"""``conv2d`` / ``conv_transpose2d`` with the call signature of
lib/model_zoo/stylegan_utils/conv2d_gradfix.py:35-43, forward AND first-order backward on the HIP kernels:

* forward: the fp32-MFMA convolution kernels (``kernels.conv2d``);
* input gradient: the opposite operator with the same weights, as in the reference (conv2d_gradfix.py:118-128): a stride-1
  convolution's is a stride-1 convolution with the flipped, channel-transposed weights, a stride-2 convolution's is the
  stride-2 transposed convolution (cropped / zero-extended by the ``output_padding`` rule of :96-105) and vice versa;
* weight gradient: ``shg_conv2d_wgrad_f32`` (replaces the cuDNN backward-weight call of :140-146), bias gradient: a sum.

Supported geometry is what the generator / discriminator need: 3x3 / 1x1 kernels, stride 1 or 2, dilation 1, symmetric
padding, groups = 1 under autograd (grouped forward only), and the stride-2 3x3 transposed form.  Second-order gradients
(``Conv2dGradWeight.backward``, needed by the R1 / path-length regularisers) are not built."""
import torch

from ... import kernels

enabled = True                       # (conv2d_gradfix.py:22) -- the HIP path is the only path; kept for interface compatibility
weight_gradients_disabled = False    # (conv2d_gradfix.py:23)


class no_weight_gradients:
    """Context manager of conv2d_gradfix.py:25-31."""

    def __enter__(self):
        global weight_gradients_disabled
        self.old = weight_gradients_disabled
        weight_gradients_disabled = True

    def __exit__(self, *exc):
        global weight_gradients_disabled
        weight_gradients_disabled = self.old


def _one(v):
    if isinstance(v, (list, tuple)):
        if len(set(v)) != 1:
            raise NotImplementedError(f'anisotropic stride/padding/dilation {v} is not supported')
        return int(v[0])
    return int(v)


def _fit(y, h, w, lo):
    """Rows/cols [lo, lo+h) x [lo, lo+w) of y, zero-extended where y ends earlier (output_padding)."""
    y = y[:, :, lo:lo + h, lo:lo + w]
    dh, dw = h - y.shape[2], w - y.shape[3]
    if dh or dw:
        y = torch.nn.functional.pad(y, (0, dw, 0, dh))
    return y.contiguous()


def _conv_fwd(x, weight, bias, stride, padding, groups):
    n, c, h, w = x.shape
    pw = kernels.conv_weight_prep(weight, groups=groups)
    y = kernels.conv2d(x.reshape(n * groups, c // groups, h, w), pw, mode=kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2,
                       pad=padding, bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y


def _convt_fwd(x, weight, bias, padding, groups):
    n, c, h, w = x.shape
    ci_g, co_g = weight.shape[0] // groups, weight.shape[1]
    # torch layout [Cin, Cout/g, kh, kw] -> per group [Cout/g, Cin/g, kh, kw]
    wg = weight.reshape(groups, ci_g, co_g, 3, 3).transpose(1, 2).reshape(groups * co_g, ci_g, 3, 3).contiguous()
    pw = kernels.conv_weight_prep(wg, groups=groups)
    y = kernels.conv2d(x.reshape(n * groups, ci_g, h, w), pw, mode=kernels.MODE_UP2T, bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if padding:
        y = y[:, :, padding:y.shape[2] - padding, padding:y.shape[3] - padding].contiguous()
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y


class _Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, w, b, stride, padding), groups = 1."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, padding, bias is not None)
        return _conv_fwd(x.detach(), weight.detach(), None if bias is None else bias.detach(), stride, padding, 1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, has_bias = ctx.geom
        k = weight.shape[2]
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if stride == 1:
                wt = weight.detach().transpose(0, 1).flip(2, 3).contiguous()
                gx = _conv_fwd(g, wt, None, 1, k - 1 - padding, 1)
            elif k == 3:
                full = _convt_fwd(g, weight.detach(), None, 0, 1)                 # [.., 2*OH+1, 2*OW+1]
                gx = _fit(full, x.shape[2], x.shape[3], padding)
            else:
                raise NotImplementedError('conv2d backward: 1x1 stride-2 convolutions (the forward decimates with upfirdn2d first)')
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            gw = kernels.conv2d_wgrad(x.detach(), g, k, k, stride, padding)
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum([0, 2, 3])
        return gx, gw, gb, None, None


class _ConvTranspose2dFn(torch.autograd.Function):
    """y = conv_transpose2d(x, w [Cin,Cout,3,3], b, stride 2, padding), groups = 1."""

    @staticmethod
    def forward(ctx, x, weight, bias, padding):
        ctx.save_for_backward(x, weight)
        ctx.geom = (padding, bias is not None)
        return _convt_fwd(x.detach(), weight.detach(), None if bias is None else bias.detach(), padding, 1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        padding, has_bias = ctx.geom
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _conv_fwd(g, weight.detach(), None, 2, padding, 1)              # conv2d_gradfix.py:124-127, transpose flipped
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            # dw[ci,co,ky,kx] = sum x[n,ci,y,x] * g[n,co,2y-p+ky,2x-p+kx]: the conv weight gradient with the tensors exchanged
            gw = kernels.conv2d_wgrad(g, x.detach(), 3, 3, 2, padding)
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum([0, 2, 3])
        return gx, gw, gb, None


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    stride, padding, dilation = _one(stride), _one(padding), _one(dilation)
    if dilation != 1 or stride not in (1, 2):
        raise NotImplementedError('conv2d: only dilation 1 and stride 1/2 are implemented in HIP')
    if _wants_grad(input, weight, bias):
        if groups != 1 or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (1, 3):
            raise NotImplementedError('conv2d backward: groups = 1 and 1x1 / 3x3 kernels only')
        return _Conv2dFn.apply(input, weight, bias, stride, padding)
    return _conv_fwd(input, weight, bias, stride, padding, groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    stride, padding, dilation = _one(stride), _one(padding), _one(dilation)
    if stride != 2 or dilation != 1 or _one(output_padding) != 0 or tuple(weight.shape[2:]) != (3, 3):
        raise NotImplementedError('conv_transpose2d: only the stride-2 3x3 form is implemented in HIP')
    if _wants_grad(input, weight, bias):
        if groups != 1:
            raise NotImplementedError('conv_transpose2d backward: groups = 1 only')
        return _ConvTranspose2dFn.apply(input, weight, bias, padding)
    return _convt_fwd(input, weight, bias, padding, groups)
