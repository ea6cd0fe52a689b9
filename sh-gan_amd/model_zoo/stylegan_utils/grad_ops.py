"""Autograd forms of the fused pointwise / FIR operators for the training rows.  The inference path
never comes here: the modules switch to these only when gradients are requested (``wants_grad``).

* ``bias_act``: bias + lrelu_agc (common/utils.py:135-143 under autograd) -- forward ``shg_bias_act_f32``, backward
  ``shg_bias_act_backward_f32`` from the saved OUTPUT (sign and clamp state are readable from it), bias gradient = channel sum;
* FIR resampling has its backward in ``upfirdn2d.py`` (``upfirdn2d`` is its own gradient with up / down exchanged,
  upfirdn2d.py:174-192); convolutions in ``conv2d_gradfix.py``.
Every backward is written with differentiable operators again, so second derivatives (the R1 / path-length regularisers,
stylegan_default_loss.py:76-91, 118-124) work."""
import torch

from ... import kernels


def wants_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


class _BiasActBwdFn(torch.autograd.Function):
    """dx = g * slope(y): linear in g, piecewise constant in y -- its derivative with respect to g is the same operator."""

    @staticmethod
    def forward(ctx, g, y, cfg):
        ctx.save_for_backward(y)
        ctx.cfg = cfg
        act, gain, alpha, act_gain, clamp = cfg
        return kernels.bias_act_backward(g.detach().contiguous(), y, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return _BiasActBwdFn.apply(gg, y, ctx.cfg), None, None


class _BiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, act, gain, alpha, act_gain, clamp):
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
        db = dx.sum([0] + list(range(2, dx.ndim))) if has_bias and ctx.needs_input_grad[1] else None
        return dx, db, None, None, None, None, None


def bias_act(x, bias=None, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    """y = lrelu_agc(x + bias[c]) (or (x + bias) * gain without activation), differentiable in x and bias.  x: [N,C,...]."""
    shape = x.shape
    x4 = x.reshape(shape[0], shape[1], -1, 1) if x.ndim != 4 else x
    return _BiasActFn.apply(x4.contiguous(), bias, bool(act), float(gain), float(alpha), float(act_gain), clamp).reshape(shape)
