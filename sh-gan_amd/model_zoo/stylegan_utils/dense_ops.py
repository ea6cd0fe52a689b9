"""The three products of a fully-connected layer as one closed family of differentiable operators on HIP kernels (training rows of
``dense``, lib/model_zoo/stylegan.py:87-98; the reference reaches them through ``torch.addmm`` and its autograd formulas, i.e. the
BLAS library):

    nt(a[N,K], b[M,K], s)  = s * a @ b^T    (+ bias[M] * bs)     forward of a layer          csrc/dense.hip: dense_kernel
    nn(a[N,M], b[M,K], s)  = s * a @ b                            its input gradient          matmul_nn_kernel
    tn(a[N,M], b[N,K], s)  = s * a^T @ b    (and s2 * colsum(a))  its weight / bias gradient  matmul_tn_kernel

Each one's gradients are the other two, so the family is differentiable to any order (the path-length regulariser differentiates
the style affines twice, stylegan_default_loss.py:76-91) without a library GEMM.  The learning-rate gains of the layer
(``weight_gain``, ``bias_gain``) travel as the scalars ``s`` / ``bs``: no separate scaling kernels in either direction."""
import torch

from ... import kernels


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _NT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias, s, bs):
        ctx.save_for_backward(a, b)
        ctx.cfg = (s, bs, bias is not None)
        return kernels.dense(_c(a.detach()), _c(b.detach()), None if bias is None else bias.detach(), wgain=s, bgain=bs)

    @staticmethod
    def backward(ctx, gy):
        a, b = ctx.saved_tensors
        s, bs, has_bias = ctx.cfg
        ga = gb = gbias = None
        if ctx.needs_input_grad[0]:
            ga = nn(gy, b, s)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gb, gbias = tn(gy, a, s, colsum_scale=bs if has_bias and ctx.needs_input_grad[2] else None)
            if not ctx.needs_input_grad[1]:
                gb = None
        return ga, gb, gbias, None, None


class _NN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, s):
        ctx.save_for_backward(a, b)
        ctx.s = s
        return kernels.matmul_nn(_c(a.detach()), _c(b.detach()), s)

    @staticmethod
    def backward(ctx, gy):
        a, b = ctx.saved_tensors
        ga = nt(gy, b, ctx.s) if ctx.needs_input_grad[0] else None
        gb = tn(a, gy, ctx.s)[0] if ctx.needs_input_grad[1] else None
        return ga, gb, None


class _TN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, s, cs):
        ctx.save_for_backward(a, b)
        ctx.cfg = (s, cs)
        out, col = kernels.matmul_tn(_c(a.detach()), _c(b.detach()), s, cs)
        if col is None:
            col = out.new_empty(())          # placeholder output (never read, never differentiated: no fill kernel for it)
            ctx.mark_non_differentiable(col)
        return out, col

    @staticmethod
    def backward(ctx, gout, gcol):
        a, b = ctx.saved_tensors
        s, cs = ctx.cfg
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = nt(b, gout, s)                                   # b @ gout^T
            if cs is not None and gcol is not None:
                ga = ga + (gcol * cs).reshape(1, -1)              # (second-order term of the bias gradient: zero in every loss here)
        if ctx.needs_input_grad[1]:
            gb = nn(a, gout, s)
        return ga, gb, None, None


def nt(a, b, s=1.0, bias=None, bias_scale=1.0):
    return _NT.apply(a, b, bias, float(s), float(bias_scale))


def nn(a, b, s=1.0):
    return _NN.apply(a, b, float(s))


def tn(a, b, s=1.0, colsum_scale=None):
    """-> (s * a^T @ b, colsum_scale * a.sum(0) or None)"""
    out, col = _TN.apply(a, b, float(s), None if colsum_scale is None else float(colsum_scale))
    return out, (col if colsum_scale is not None else None)


def linear(x, weight, bias=None, weight_gain=1.0, bias_gain=1.0):
    """x [N,K] @ (weight [M,K] * weight_gain)^T + bias * bias_gain, differentiable to any order in all three."""
    if x.ndim != 2 or x.dtype != torch.float32:
        raise NotImplementedError('dense_ops.linear: float32 [N, K] rows (the reference keeps its dense layers float32 too)')
    return nt(x, weight, weight_gain, bias, bias_gain)
