"""Autograd forms of the fused pointwise / FIR operators for the training rows.  The inference path
never comes here: the modules switch to these only when gradients are requested (``wants_grad``).

* ``bias_act``: bias + lrelu_agc (common/utils.py:135-143 under autograd) -- forward ``shg_bias_act_f32``, backward
  ``shg_bias_act_backward_f32`` from the saved OUTPUT (sign and clamp state are readable from it), bias gradient = channel sum;
* FIR resampling has its backward in ``upfirdn2d.py`` (``upfirdn2d`` is its own gradient with up / down exchanged,
  upfirdn2d.py:174-192); convolutions in ``conv2d_gradfix.py``.
Every backward is written with differentiable operators again, so second derivatives (the R1 / path-length regularisers,
stylegan_default_loss.py:76-91, 118-124) work."""
import torch

from ... import kernels, kernels_f16


def wants_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def generic_route(x, *ts):
    """True when a module must compose its forward from the differentiable operators instead of the fused fp32 inference kernels:
    gradients are requested, or ``x`` is a float16 activation (the reference's ``use_fp16`` blocks: the fused kernels are fp32 NCHW,
    the fp16 route is the non-fused algebra of stylegan.py:172-181 on the NHWC fp16-MFMA kernels -- with or without autograd)."""
    return (isinstance(x, torch.Tensor) and x.dtype == torch.float16) or wants_grad(x, *ts)


CL = torch.channels_last


def to_block_dtype(x, use_fp16):
    """``x.to(dtype)`` at a block boundary (stylegan.py:486-495,660-663; comodgan.py:40-43,305-312): float16 activations are kept
    channels-last (NHWC) -- the layout the fp16 kernels take --, float32 ones NCHW-contiguous."""
    if x is None:
        return None
    if use_fp16:
        return x.to(dtype=torch.float16, memory_format=CL)
    return x.to(dtype=torch.float32, memory_format=torch.contiguous_format)


class _BiasActBwdFn(torch.autograd.Function):
    """dx = g * slope(y): linear in g, piecewise constant in y -- its derivative with respect to g is the same operator."""

    @staticmethod
    def forward(ctx, g, y, cfg):
        ctx.save_for_backward(y)
        ctx.cfg = cfg
        act, gain, alpha, act_gain, clamp = cfg
        if y.dtype == torch.float16:
            return kernels_f16.bias_act_backward(g.detach().to(torch.float16), y, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
        return kernels.bias_act_backward(g.detach().contiguous(), y, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return _BiasActBwdFn.apply(gg, y, ctx.cfg), None, None


class _BiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, act, gain, alpha, act_gain, clamp):
        if x.dtype == torch.float16:
            y = kernels_f16.bias_act(x.detach(), bias=None if bias is None else bias.detach(), act=act, gain=gain, alpha=alpha,
                                     act_gain=act_gain, clamp=clamp)
        else:
            y = kernels.bias_act(x.detach(), bias=None if bias is None else bias.detach(), act=act, gain=gain, alpha=alpha,
                                 act_gain=act_gain, clamp=clamp)
        ctx.save_for_backward(y)
        ctx.cfg = (act, gain, alpha, act_gain, clamp, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        act, gain, alpha, act_gain, clamp, has_bias = ctx.cfg
        dx = _BiasActBwdFn.apply(g, y, (act, gain, alpha, act_gain, clamp))
        db = None
        if has_bias and ctx.needs_input_grad[1]:
            db = dx.sum([0] + list(range(2, dx.ndim)), dtype=torch.float32)          # (fp16 layers: the channel sum in fp32; the bias is an fp32 parameter)
        return dx, db, None, None, None, None, None


def bias_act(x, bias=None, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    """y = lrelu_agc(x + bias[c]) (or (x + bias) * gain without activation), differentiable in x and bias.  x: [N,C,...]."""
    shape = x.shape
    if x.dtype == torch.float16:
        if x.ndim != 4:
            raise NotImplementedError('bias_act: fp16 tensors are rank-4 activations (dense layers stay fp32 in the reference too)')
        return _BiasActFn.apply(x, bias, bool(act), float(gain), float(alpha), float(act_gain), clamp)
    x4 = x.reshape(shape[0], shape[1], -1, 1) if x.ndim != 4 else x
    return _BiasActFn.apply(x4.contiguous(), bias, bool(act), float(gain), float(alpha), float(act_gain), clamp).reshape(shape)
