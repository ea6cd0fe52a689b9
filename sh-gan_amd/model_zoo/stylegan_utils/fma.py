"""a*b+c with NumPy broadcasting, differentiable any number of times (interface and gradients of
lib/model_zoo/stylegan_utils/fma.py:15-58) on the HIP kernels: the forward reads broadcast operands through their strides
(``shg_fma_bcast``), the gradient of an operand is ONE product + reduction pass over the broadcast dimensions (``shg_mul_reduce`` =
``_unbroadcast(dout * b, a.shape)``, fma.py:40-58).  float32 or float64 HIP tensors; there is no CPU path."""
import torch

from ... import kernels


def fma(a, b, c):  # => a * b + c
    return _FusedMultiplyAdd.apply(a, b, c)


class _FusedMultiplyAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return kernels.fma(a, b, c)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _MulUnbroadcast.apply(dout, b, a.shape) if ctx.needs_input_grad[0] else None
        db = _MulUnbroadcast.apply(dout, a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _Unbroadcast.apply(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc


class _MulUnbroadcast(torch.autograd.Function):
    """y = _unbroadcast(g * b, shape).  Its own gradients are the same two operators again: dg = broadcast(gy) * b,
    db = _unbroadcast(broadcast(gy) * g, b.shape) -- so R1 / path-length style double backward stays on the kernels."""
    @staticmethod
    def forward(ctx, g, b, shape):
        ctx.save_for_backward(g, b)
        return kernels.mul_reduce(g, b, shape)

    @staticmethod
    def backward(ctx, gy):
        g, b = ctx.saved_tensors
        gy = _lead(gy, g.ndim)
        dg = _FusedMultiplyAdd.apply(gy, b, _zero(g)).expand(g.shape) if ctx.needs_input_grad[0] else None
        db = _MulUnbroadcast.apply(g, gy, b.shape) if ctx.needs_input_grad[1] else None
        return dg, db, None


class _Unbroadcast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, shape):
        ctx.g_shape = g.shape
        return kernels.mul_reduce(g, None, shape)

    @staticmethod
    def backward(ctx, gy):
        return _lead(gy, len(ctx.g_shape)).expand(ctx.g_shape), None


def _lead(t, ndim):
    return t.reshape((1,) * (ndim - t.ndim) + tuple(t.shape))


def _zero(t):
    return torch.zeros((), dtype=t.dtype, device=t.device)
