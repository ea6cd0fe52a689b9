"""Autograd forms of the fused pointwise / FIR operators for the training rows.  The inference path
never comes here: the modules switch to these only when gradients are requested (``wants_grad``).

* ``bias_act``: bias + lrelu_agc (common/utils.py:135-143 under autograd) -- forward ``shg_bias_act_f32``, backward
  ``shg_bias_act_backward_f32`` from the saved OUTPUT (sign and clamp state are readable from it), bias gradient = channel sum;
* FIR resampling has its backward in ``upfirdn2d.py`` (``upfirdn2d`` is its own gradient with up / down exchanged,
  upfirdn2d.py:174-192); convolutions in ``conv2d_gradfix.py``.
Every backward is written with differentiable operators again, so second derivatives (the R1 / path-length regularisers,
stylegan_default_loss.py:76-91, 118-124) work."""
import os
import threading

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
    want = torch.float16 if use_fp16 else torch.float32
    if RELAYOUT_KERNEL and x.dtype != want and x.dtype in (torch.float16, torch.float32) and kernels_f16.relayout_supported(x):
        return _RelayoutFn.apply(x)
    if use_fp16:
        return x.to(dtype=torch.float16, memory_format=CL)
    return x.to(dtype=torch.float32, memory_format=torch.contiguous_format)


RELAYOUT_KERNEL = True       # (A/B switch: False = torch's .to(dtype, memory_format))


class _RelayoutFn(torch.autograd.Function):
    """The cast between the two activation layouts as one transposing kernel; its gradient is the opposite cast."""
    @staticmethod
    def forward(ctx, x):
        return kernels_f16.relayout(x.detach())

    @staticmethod
    def backward(ctx, g):
        if g.dtype == torch.float32:
            g = g.contiguous()
        else:
            g = g.contiguous(memory_format=CL)
        return _RelayoutFn.apply(g)


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


def _backward_pass_id():
    f = getattr(torch._C, '_current_graph_task_id', None)
    return f() if f is not None else 0


class InputGradJoin:
    """A tensor with two consumers -- a residual block's input feeds its skip branch and its first convolution -- gets the sum of their
    input gradients; autograd forms it with an extra pass over the tensor (0.58 ms for the critic's [16, 64, 512, 512] input).  The
    convolution kernels can add a tensor in their store pass instead: the branch that finishes FIRST in backward leaves its gradient here
    (``stash``), the node registered as consumer (conv2d_gradfix._Conv2dFn / _ConvBiasActFn, conv2d_resample._FirDownConvFn, via ``pending`` /
    ``adopt``) adds it as the ``residual`` of its input-gradient kernel.  Whatever the engine's order, the result is the same sum: a gradient that arrives after the consumer has run
    takes autograd's ordinary path.  Under ``create_graph`` both sides stay on the ordinary path.  A stash is tagged with the id of the
    backward pass that left it (``torch._C._current_graph_task_id``): a gradient left behind by a pass that died before its consumer ran is
    never added to a later pass.  ``pending`` is per thread (threaded data-parallel forwards do not see each other's joins)."""
    _tls = threading.local()       # .pending: the join the next _Conv2dFn.forward adopts (set by ``consumer``, cleared when adopted)

    def __init__(self, x):
        self.grad, self.armed, self.consumer_done, self.task = None, False, False, None
        self.key = (x.data_ptr(), tuple(x.shape), x.dtype)           # the consumer must be a node whose INPUT is this tensor

    @classmethod
    def adopt(cls, x, needs_input_grad):
        """Called by a node's forward with its input: the pending join if it was opened for this very tensor, else None."""
        j = getattr(cls._tls, 'pending', None)
        if j is None or j.key != (x.data_ptr(), tuple(x.shape), x.dtype):
            return None
        cls._tls.pending = None
        j.armed = bool(needs_input_grad)
        return j

    def take(self):
        """Called by the consumer's backward: the stashed gradient (first-order passes only) or None; later arrivals go autograd's way."""
        g = None
        if self.grad is not None and not torch.is_grad_enabled() and self.task == _backward_pass_id():
            g = self.grad
        self.grad = None                                            # (a stash of another pass is dropped, not added)
        self.consumer_done = True
        return g

    class consumer:
        def __init__(self, join):
            self.join = join

        def __enter__(self):
            InputGradJoin._tls.pending = self.join

        def __exit__(self, *exc):
            InputGradJoin._tls.pending = None


class _StashGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, join):
        ctx.join = join
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        j = ctx.join
        if torch.is_grad_enabled() or not j.armed or j.consumer_done or tuple(g.shape) != j.key[1]:
            return g, None
        if j.grad is not None and j.task == _backward_pass_id():   # a second stash in the same pass: autograd's way
            return g, None
        j.grad, j.task = g, _backward_pass_id()
        return None, None


def stash_input_grad(x, join):
    """Identity whose gradient goes to ``join`` (see InputGradJoin) instead of autograd's accumulation, when a consumer is waiting."""
    return _StashGradFn.apply(x, join) if wants_grad(x) else x


class _ScaleCastFn(torch.autograd.Function):
    """``(w * gain).to(float16)`` (forward) / ``g.float() * gain`` (its gradient) as one kernel each; linear, so the backward is the same
    Function with the dtypes exchanged (differentiable to any order)."""
    @staticmethod
    def forward(ctx, x, gain, to_half):
        ctx.gain, ctx.to_half = gain, to_half
        return kernels.scale_cast(x.detach(), gain, to_half)

    @staticmethod
    def backward(ctx, g):
        return _ScaleCastFn.apply(g, ctx.gain, not ctx.to_half), None, None


SCALE_CAST_KERNEL = os.environ.get('SHG_SCALE_CAST', '1') == '1'     # (A/B switch; SHG_SCALE_CAST=0: the tensor operators)


def scaled_weight(weight, gain, dtype):
    """``(weight * gain).to(dtype)`` of a layer's float32 master weight (stylegan.py:228,236-238); for float16 layers on the HIP device one
    launch forward and one backward instead of a product and a cast each way."""
    if SCALE_CAST_KERNEL and dtype == torch.float16 and weight.dtype == torch.float32 and weight.is_cuda:
        return _ScaleCastFn.apply(weight, float(gain), True)
    return (weight * gain).to(dtype)


def channel_sum(t):
    """Sum over every axis but the channel axis, accumulated in float32 (float64 inputs: float64; differentiable).  A reduction to C values runs on C workgroups:
    the RGB branch's [8, 3, 512, 512] bias gradient took 670 us as one reduction; rows first, then the rest, is two launches of ~6 us."""
    acc = t.dtype if t.dtype == torch.float64 else torch.float32
    if t.ndim == 4 and t.shape[1] <= 32 and t.shape[2] * t.shape[3] >= 16384:
        return t.sum(3, dtype=acc).sum([0, 2])
    return t.sum([0] + list(range(2, t.ndim)), dtype=acc)


class _ChannelBiasFn(torch.autograd.Function):
    """y + bias[c] for the layers the fused bias/activation kernels do not take (3-channel float16 RGB outputs).  Written as a node so that
    the bias gradient is ``channel_sum`` -- autograd's own reduction of the broadcast ([8, 3, 512, 512] -> 3 values) runs on 3 workgroups:
    655 us per step."""
    @staticmethod
    def forward(ctx, y, bias):
        ctx.bias_dtype = bias.dtype
        return y + bias.detach().view(1, -1, 1, 1).to(y.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, (channel_sum(g).to(ctx.bias_dtype) if ctx.needs_input_grad[1] else None)


def add_channel_bias(y, bias):
    return _ChannelBiasFn.apply(y, bias)


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
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        act, gain, alpha, act_gain, clamp, has_bias = ctx.cfg
        if has_bias and ctx.needs_input_grad[1] and not torch.is_grad_enabled() and modtail_supported(y):
            # first order: dx and the per-(n, c) sums of the SAME pass (the tail kernel without scale / noise) instead of a second trip over
            # dx for the bias gradient -- 25 half / 100 float reductions of 70 us per training step
            if y.dtype == torch.float16:
                dx, _, s0, _ = kernels_f16.modtail_backward(g.detach().to(torch.float16), y, None, None, want_sums=True, want_noise=False, act=act,
                                                            gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
            else:
                dx, _, s0, _ = kernels.modtail_backward(g.detach().contiguous(), y, None, None, want_sums=True, want_noise=False, act=act, gain=gain,
                                                        alpha=alpha, act_gain=act_gain, clamp=clamp)
            return dx, s0.sum(0).to(ctx.bias_dtype), None, None, None, None, None
        # (create_graph -- R1: the plain slope kernel + a channel sum; the fused sums kernel as a node measured 2 ms slower per Dreg pass)
        dx = _BiasActBwdFn.apply(g, y, (act, gain, alpha, act_gain, clamp))
        db = None
        if has_bias and ctx.needs_input_grad[1]:
            db = channel_sum(dx)                                                     # (fp16 layers: the channel sum in fp32; the bias is an fp32 parameter)
        return dx, db, None, None, None, None, None


def bias_act_grads(g, y, cfg, need_bias, bias_dtype):
    """(dL/dz, dL/dbias | None) of y = A(z + bias) from g = dL/dy and the saved OUTPUT y: what ``_BiasActFn.backward`` computes, for nodes
    that fuse the activation into their producer (conv2d_gradfix._ConvBiasActFn).  First-order passes take dz and the per-(n, c) sums from
    one kernel; under ``create_graph`` dz is the differentiable ``_BiasActBwdFn`` and the bias gradient a tensor reduction of it."""
    act, gain, alpha, act_gain, clamp = cfg
    if need_bias and not torch.is_grad_enabled() and modtail_supported(y):
        if y.dtype == torch.float16:
            dz, _, s0, _ = kernels_f16.modtail_backward(g.detach().to(torch.float16), y, None, None, want_sums=True, want_noise=False, act=act,
                                                        gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
        else:
            dz, _, s0, _ = kernels.modtail_backward(g.detach().contiguous(), y, None, None, want_sums=True, want_noise=False, act=act, gain=gain,
                                                    alpha=alpha, act_gain=act_gain, clamp=clamp)
        return dz, s0.sum(0).to(bias_dtype)
    dz = _BiasActBwdFn.apply(g, y, cfg)
    return dz, (channel_sum(dz).to(bias_dtype) if need_bias else None)


def bias_act(x, bias=None, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    """y = lrelu_agc(x + bias[c]) (or (x + bias) * gain without activation), differentiable in x and bias.  x: [N,C,...]."""
    shape = x.shape
    if x.dtype == torch.float16:
        if x.ndim != 4:
            raise NotImplementedError('bias_act: fp16 tensors are rank-4 activations (dense layers stay fp32 in the reference too)')
        return _BiasActFn.apply(x, bias, bool(act), float(gain), float(alpha), float(act_gain), clamp)
    x4 = x.reshape(shape[0], shape[1], -1, 1) if x.ndim != 4 else x
    return _BiasActFn.apply(x4.contiguous(), bias, bool(act), float(gain), float(alpha), float(act_gain), clamp).reshape(shape)


class _ModTailFn(torch.autograd.Function):
    """y = A(t * d[n,c] + noise + bias[c]) in ONE kernel, and its first-order backward in ONE kernel + three tiny reductions -- float16
    NHWC activations (csrc/conv_f16.hip modtail kernels) and float32 NCHW ones (forward: the fused bias_act kernel; backward:
    modtail_backward_f32_kernel) -- instead of one tensor pass per operation and per gradient (the training-route fusion of
    stylegan.py:173,176-181,298-304).  Under ``create_graph`` (R1 / path-length regularisers) the backward is composed from the
    differentiable operators instead, so second derivatives keep working."""

    @staticmethod
    def forward(ctx, t, d, noise, bias, cfg):
        act, gain, alpha, act_gain, clamp = cfg
        if t.dtype == torch.float16:
            y = kernels_f16.modtail(t.detach(), None if d is None else d.detach(), None if noise is None else noise.detach(),
                                    None if bias is None else bias.detach(), act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
        else:
            y = kernels.bias_act(t.detach(), bias=None if bias is None else bias.detach(), scale=None if d is None else d.detach().reshape(-1),
                                 noise=None if noise is None else noise.detach(), act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
        ctx.save_for_backward(t, y, d, noise, bias)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, gy):
        t, y, d, noise, bias = ctx.saved_tensors
        act, gain, alpha, act_gain, clamp = ctx.cfg
        need_t, need_d, need_n, need_b = ctx.needs_input_grad[:4]
        n, c = y.shape[0], y.shape[1]
        if torch.is_grad_enabled() and CLOSED_TAIL_BACKWARD:
            # create_graph (R1 / path length): every gradient of the tail from the fused kernel, as ONE differentiable node.  (needs_input_grad
            # says which inputs CAN receive a gradient, not which ones this pass wants: the path-length pass asks for the latents only, and
            # the noise / bias sums it never uses must not cost tensor passes either.)
            gt, gd, s0, gnz = _ModTailBwdFn.apply(gy, y, t if d is not None else None, d, bool(noise is not None and need_n), ctx.cfg)
            gb = s0.sum(0).to(bias.dtype) if (bias is not None and need_b) else None
            gn = None
            if noise is not None and need_n:
                gn = (gnz.sum(0) if noise.numel() != gnz.numel() else gnz).reshape(noise.shape).to(noise.dtype)
            return (gt if need_t else None), (gd if (d is not None and need_d) else None), gn, gb, None
        if torch.is_grad_enabled():
            # create_graph: the same quantities from differentiable pieces (gz is linear in gy, piecewise constant in y)
            gz = _BiasActBwdFn.apply(gy, y, ctx.cfg)
            gt = gd = gn = gb = None
            if need_t:
                gt = gz if d is None else gz * d.to(gz.dtype).reshape(n, c, 1, 1)
            if d is not None and need_d:
                gd = (gz.float() * t.float()).sum([2, 3]).reshape(d.shape).to(d.dtype)
            if noise is not None and need_n:
                gn = gz.float().sum(1, keepdim=True)
                gn = (gn.sum(0) if noise.numel() != gn.numel() else gn).reshape(noise.shape).to(noise.dtype)
            if bias is not None and need_b:
                gb = channel_sum(gz).to(bias.dtype)
            return gt, gd, gn, gb, None
        want_sums = (d is not None and need_d) or (bias is not None and need_b)
        if y.dtype == torch.float16:
            gt, s1, s0, gnz = kernels_f16.modtail_backward(gy.detach().to(torch.float16), y, t if (d is not None and need_d) else None, d,
                                                           want_sums=want_sums, want_noise=(noise is not None and need_n), act=act, gain=gain,
                                                           alpha=alpha, act_gain=act_gain, clamp=clamp)
        else:
            gt, s1, s0, gnz = kernels.modtail_backward(gy.detach(), y, t if (d is not None and need_d) else None, d, want_sums=want_sums,
                                                       want_noise=(noise is not None and need_n), act=act, gain=gain, alpha=alpha,
                                                       act_gain=act_gain, clamp=clamp)
        gd = s1.reshape(d.shape).to(d.dtype) if (d is not None and need_d) else None
        gb = s0.sum(0).to(bias.dtype) if (bias is not None and need_b) else None
        gn = None
        if gnz is not None:
            gn = (gnz.sum(0) if noise.numel() != gnz.numel() else gnz).reshape(noise.shape).to(noise.dtype)
        return (gt if need_t else None), gd, gn, gb, None


CLOSED_TAIL_BACKWARD = True      # (A/B switch: False = the tensor-operator composition under create_graph)


def _tail_backward_kernel(gy, y, t, d, want_sums, cfg, u=None, e=None):
    """(A'(y) * (gy * d [+ u * e]), sum_hw gy * A'(y) * t) on the fused first-order kernel of either layout."""
    act, gain, alpha, act_gain, clamp = cfg
    if y.dtype == torch.float16:
        gt, s1, _, _ = kernels_f16.modtail_backward(gy.detach().to(torch.float16), y, None if t is None else t.detach().to(torch.float16), d,
                                                    want_sums=want_sums, want_noise=False, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp,
                                                    u=None if u is None else u.detach().to(torch.float16), e=e)
    else:
        gt, s1, _, _ = kernels.modtail_backward(gy.detach().contiguous(), y, None if t is None else t.detach().contiguous(), d, want_sums=want_sums,
                                                want_noise=False, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp,
                                                u=None if u is None else u.detach().contiguous(), e=e)
    return gt, s1


class _ModTailBwdFn(torch.autograd.Function):
    """(gt, gd, s0, gnoise) = (gz * d[n,c], sum_hw gz * t, sum_hw gz, sum_c gz) with gz = gy * A'(y): the first-order backward of the
    modulation tail as a node of its own, for ``create_graph`` passes (d / t may be None: the plain bias + activation tail).  Scaling by a
    per-(n, c) factor and the per-(n, c) dot product are each other's derivatives, so the node's own backward is the same kernel again
    (A' is piecewise constant: y carries no gradient):
        d/dgy = A'(y) * (ggt * d + t * ggd [+ ggs0 + ggn])      d/dt = gz * ggd      d/dd = sum_hw gz * ggt
    -- two passes of the fused kernel (its optional second product carries the sum) instead of ~10 broadcasting tensor operators per layer
    (stylegan_default_loss.py:72-88: the path-length regulariser differentiates every modulated layer of the synthesis network twice).
    Gradients arriving through the two sums (nothing in the shipped losses sends any) are added with tensor operators."""
    @staticmethod
    def forward(ctx, gy, y, t, d, want_noise, cfg):
        act, gain, alpha, act_gain, clamp = cfg
        kw = dict(want_sums=True, want_noise=want_noise, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
        d_ = None if d is None else d.detach()
        if y.dtype == torch.float16:
            gt, s1, s0, gnz = kernels_f16.modtail_backward(gy.detach().to(torch.float16), y, None if t is None else t.detach(), d_, **kw)
        else:
            gt, s1, s0, gnz = kernels.modtail_backward(gy.detach().contiguous(), y, None if t is None else t.detach(), d_, **kw)
        ctx.save_for_backward(gy, y, t, d)
        ctx.cfg = cfg
        gd = s1.reshape(d.shape).to(d.dtype) if (d is not None and t is not None) else None
        return gt, gd, s0, gnz

    @staticmethod
    def backward(ctx, ggt, ggd, ggs0, ggn):
        gy, y, t, d = ctx.saved_tensors
        cfg = ctx.cfg
        n, c = y.shape[0], y.shape[1]
        need_gy, _, need_t, need_d = ctx.needs_input_grad[:4]
        need_t, need_d = need_t and t is not None, need_d and d is not None
        if ggd is not None and t is None:
            ggd = None
        extra = None                              # gradients through the plain sums: broadcast back (tensor operators; differentiable)
        if ggs0 is not None:
            extra = ggs0.reshape(n, c, 1, 1).to(y.dtype).expand_as(y)
        if ggn is not None:
            e2 = ggn.reshape(n, 1, y.shape[2], y.shape[3]).to(y.dtype).expand_as(y)
            extra = e2 if extra is None else extra + e2
        if torch.is_grad_enabled():            # a third derivative: tensor operators (never taken by the shipped losses)
            gz = _BiasActBwdFn.apply(gy, y, cfg)
            g_gy = g_t = g_d = None
            ggd4 = None if ggd is None else ggd.reshape(n, c, 1, 1)
            if need_gy:
                v = 0 if extra is None else extra
                if ggt is not None:
                    v = v + (ggt if d is None else ggt * d.to(ggt.dtype).reshape(n, c, 1, 1))
                if ggd is not None:
                    v = v + t * ggd4.to(t.dtype)
                g_gy = _BiasActBwdFn.apply(v, y, cfg) if torch.is_tensor(v) else None
            if need_t and ggd is not None:
                g_t = gz * ggd4.to(gz.dtype)
            if need_d and ggt is not None:
                g_d = (gz.float() * ggt.float()).sum([2, 3]).reshape(d.shape).to(d.dtype)
            return g_gy, None, g_t, g_d, None, None
        g_gy = g_t = g_d = None
        ggd_f = None if ggd is None else ggd.detach().reshape(n, c).float().contiguous()
        if (need_t and ggd is not None) or (need_d and ggt is not None):
            # one pass: gz * ggd and sum_hw gz * ggt (a missing factor: the pass still gives the other quantity)
            want = need_d and ggt is not None
            gz_s, s1 = _tail_backward_kernel(gy, y, ggt if want else None, ggd_f, want, cfg)
            if need_t and ggd is not None:
                g_t = gz_s
            if want:
                g_d = s1.reshape(d.shape).to(d.dtype)
        if need_gy:
            d_f = None if d is None else d.detach().reshape(n, c).float().contiguous()
            a = None
            if ggt is not None and ggd is not None:            # A'(y) (ggt d + t ggd) in one pass
                a, _ = _tail_backward_kernel(ggt, y, None, d_f, False, cfg, u=t, e=ggd_f)
            elif ggt is not None:
                a, _ = _tail_backward_kernel(ggt, y, None, d_f, False, cfg)
            elif ggd is not None:
                a, _ = _tail_backward_kernel(t, y, None, ggd_f, False, cfg)
            if extra is not None:
                ex = extra.contiguous(memory_format=CL) if y.dtype == torch.float16 else extra.contiguous()
                b = _tail_backward_kernel(ex, y, None, None, False, cfg)[0]
                a = b if a is None else a.add_(b)
            g_gy = a
        return g_gy, None, g_t, g_d, None, None


def modconv_tail(t, d=None, noise=None, bias=None, act=False, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    """Fused modulation tail: ``lrelu_agc(t * d[n,c] + noise + bias[c])`` (or ``(..) * gain`` without activation).
    t [N,C,H,W] float16 (channels_last; C a multiple of 8 with C/8 a power of two <= 64) or float32 (NCHW; H*W a multiple of 4, C <= 512);
    d [N,C], noise [H,W] / [N,1,H,W], bias [C] float32, each optional (``modtail_supported`` tells; callers fall back to the
    per-operation form otherwise)."""
    if t.dtype == torch.float32:
        t = t.contiguous()
    return _ModTailFn.apply(t, d, noise, bias, (bool(act), float(gain), float(alpha), float(act_gain), clamp))


def modtail_supported(t):
    if t.ndim != 4:
        return False
    if t.dtype == torch.float32:            # NCHW kernel: 4 pixels per lane, <= 512 channels
        return (t.shape[2] * t.shape[3]) % 4 == 0 and t.shape[1] <= 512 and t.is_cuda
    c8 = t.shape[1] // 8
    return t.dtype == torch.float16 and t.shape[1] % 8 == 0 and 1 <= c8 <= 64 and (c8 & (c8 - 1)) == 0
