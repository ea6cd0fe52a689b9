"""StyleGAN2 building blocks of the SH-GAN generator, host side (PyTorch-ROCm nn.Modules whose
forward enqueues libshgan_hip kernels).  Mirrors the module/operator surface and the state_dict
schema of the reference's lib/model_zoo/stylegan.py (``conv2d`` :28, ``dense`` :66,
``modulated_conv2d`` :103, ``conv2d_layer`` :195, ``synthesis_layer`` :243, ``torgb_layer`` :306,
``Mapping`` :347, ``synthesis_block`` :436, ``Synthesis`` :523, ``Generator`` :581,
``discrim_block`` :624, ``discrim_epilogue`` :707) so released checkpoints load with strict=True.

MI355X-first differences:
  * modulation is executed in the "scale activations" algebra (stylegan.py:172-181) for every value
    of ``fused_modconv``: one batch-shared GEMM weight + per-sample in/out scales, instead of the
    groups=batch grouped convolution with N materialised weight copies;
  * weight layout transforms / demodulation tables are cached per layer and refreshed when the
    parameter changes (version counter), so an eval forward does no per-call weight work;
  * bias, noise, activation, skip-add, FIR and RGB-upsample are epilogues of the producing kernels.
float32 is the default and the only arithmetic of the shipped configs; the ``use_fp16`` options of the reference run on the NHWC
fp16-MFMA kernels (csrc/conv_f16.hip) through the non-fused algebra, with or without autograd."""
import math

import os

import numpy as np
import torch
import torch.nn as nn

from .. import kernels
from .common import utils
from .common.get_model import get_model, register
from .stylegan_utils import grad_ops, conv2d_gradfix, conv2d_resample, fma, misc, upfirdn2d  # noqa: F401

version = '0'
symbol = 'stylegan'


# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------

def _make_act(activation):
    return None if activation is None else utils.get_unit()(activation)()


def _act_kwargs(act_obj, gain=1.0):
    """kwargs for the fused epilogues from an activation object (only lrelu_agc can be fused)."""
    if act_obj is None:
        return dict(act=False, gain=gain)
    if isinstance(act_obj, utils.lrelu_agc):
        return dict(act=True, gain=gain, alpha=act_obj.alpha, act_gain=act_obj.gain, clamp=act_obj.clamp)
    return None


def _act_generic(y, ak):
    """lrelu_agc from its kwargs with tensor ops (any dtype / channel count): common/utils.py:135-143."""
    g = float(ak.get('gain', 1.0))
    if ak.get('act'):
        y = torch.nn.functional.leaky_relu(y, float(ak['alpha'])) * (float(ak['act_gain']) * g)
        c = ak.get('clamp')
        return y if c is None else y.clamp(-float(c) * g, float(c) * g)
    return y * g if g != 1.0 else y


FUSE_BLOCK_SUM = os.environ.get('SHG_FUSE_BLOCK_SUM', '1') == '1'       # (A/B switch: 0 = the residual blocks' sum as an elementwise pass)


def _add(a, b):
    """a + b: one fused kernel on the inference path, a differentiable tensor op on the training path."""
    if grad_ops.generic_route(a, b):
        return a + b
    return kernels.bias_act(a, residual=b, act=False)


class _ParamCache:
    """Derived device tensors (GEMM-layout weights, host copies of scalars) keyed by the identity and
    version counter of the parameters they were computed from.

    CONTRACT.  A write the version counter sees refreshes the entry by itself: ``p.copy_()``, ``p.add_()`` ... under ``no_grad``,
    optimiser steps, ``load_state_dict``, ``.to()`` / ``.half()`` (new storage).  A write THROUGH ``p.data`` (``p.data.copy_(v)``,
    ``p.data.mul_(b)``: some EMA loops and hand-rolled loaders) has a version counter of its own and is invisible here, as is a HIP
    graph replay that updates parameters: call ``_ParamCache.invalidate_all()`` (= ``stylegan.invalidate_param_caches()``) after it.
    ``_ParamCache.guard = True`` (or the environment variable SHGAN_PARAM_GUARD=1) makes every cache hit compare a fingerprint of
    the parameters' bytes with the one taken when the entry was built and raise on a stale entry -- a synchronising check, meant
    for bringing up a new training / loading loop, not for the timed path."""

    epoch = 0          # bumped by invalidate_all(): a replayed HIP graph changes parameters without touching their version counters
    guard = os.environ.get('SHGAN_PARAM_GUARD', '0') not in ('', '0')
    builds = 0         # entries (re)built so far: a stream pipeline reads it to learn whether a batch had to prepare weights

    def __init__(self):
        self.store = {}

    @classmethod
    def invalidate_all(cls):
        cls.epoch += 1

    @staticmethod
    def _fingerprint(params):
        """Exact for the guard's purpose: the sum of the raw bit patterns as integers (every changed bit changes it, up to wrap-around)
        plus the first element -- one small reduction and one host read per parameter."""
        out = []
        for p in params:
            t = p.detach().reshape(-1)
            if t.numel() == 0:
                out.append((0, 0))
                continue
            bits = t.view(torch.int32 if t.element_size() == 4 else torch.int16 if t.element_size() == 2 else torch.int64)
            out.append((int(bits.to(torch.int64).sum().item()), int(bits[0].item())))
        return tuple(out)

    def get(self, tag, params, builder):
        key = (_ParamCache.epoch,) + tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        hit = self.store.get(tag)
        if hit is None or hit[0] != key:
            _ParamCache.builds += 1
            hit = (key, builder(), self._fingerprint(params) if _ParamCache.guard else None)
            self.store[tag] = hit
        elif _ParamCache.guard and hit[2] is not None and hit[2] != self._fingerprint(params):
            raise RuntimeError(
                f'stale prepared weights for {tag!r}: a parameter was written without its version counter moving (a write through '
                '`.data`, or a HIP-graph replay). Write with `p.copy_()` under no_grad, or call stylegan.invalidate_param_caches() '
                'after the write.')
        return hit[1]

    def __deepcopy__(self, memo):
        return _ParamCache()


def invalidate_param_caches():
    """Drop every prepared weight of every module (see ``_ParamCache``): after writes through ``.data`` or a HIP-graph replay
    that updates parameters."""
    _ParamCache.invalidate_all()


def _cache_of(module):
    c = module.__dict__.get('_shg_cache')
    if c is None:
        c = _ParamCache()
        module.__dict__['_shg_cache'] = c
    return c


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------

class conv2d(nn.Conv2d):
    """nn.Conv2d with He-normal init and an optional runtime weight scale (stylegan.py:28-64);
    used by the SHU for its 1x1 spectral convolution.  Forward = MFMA conv kernel."""

    def __init__(self, *args, **kwargs):
        use_wscale = kwargs.pop('use_wscale', False)
        super().__init__(*args, **kwargs)
        in_channels = args[0] if len(args) > 0 else kwargs['in_channels']
        kernel_size = args[2] if len(args) > 2 else kwargs['kernel_size']
        he_std = 1.0 / np.sqrt(in_channels * kernel_size * kernel_size)
        self.weight_gain = he_std if use_wscale else 1
        self.bias_gain = 1
        nn.init.normal_(self.weight, mean=0.0, std=1.0 if use_wscale else he_std)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def prepped(self):
        return _cache_of(self).get('w', [self.weight], lambda: kernels.conv_weight_prep(
            self.weight.detach(), gain=self.weight_gain))

    def forward(self, x, relu=False):
        if self.padding_mode != 'zeros' or self.groups != 1 or tuple(self.dilation) != (1, 1):
            raise NotImplementedError('conv2d: only zero padding, groups=1, dilation=1 run on the HIP path')
        stride, pad = self.stride[0], self.padding[0]
        if grad_ops.wants_grad(x, self.weight, self.bias):
            from .stylegan_utils import conv2d_gradfix
            y = conv2d_gradfix.conv2d(x, self.weight * self.weight_gain, None if self.bias is None else self.bias * self.bias_gain,
                                      stride=stride, padding=pad)
            return torch.relu(y) if relu else y
        mode = kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2
        b = self.bias.detach() if self.bias is not None else None
        return kernels.conv2d(x, self.prepped(), mode=mode, pad=pad, bias=b, act=relu, alpha=0.0, act_gain=1.0, clamp=None)


class dense(nn.Module):
    """y = act(x @ (W * lr/sqrt(in))^T + b*lr)   (stylegan.py:66-101)."""

    def __init__(self, in_features, out_features, bias=True, bias_init=0, activation=None, lr_multi=1):
        super().__init__()
        self.activation = _make_act(activation)
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multi)
        self.bias = nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multi / np.sqrt(in_features)
        self.bias_gain = lr_multi
        self.lr_multi, self.bias_init = lr_multi, bias_init
        self.repr = 'dense({}, {}, bias={}, act={}, lr_multi={})'.format(in_features, out_features, bias, activation, lr_multi)

    def forward(self, x, out=None):
        if grad_ops.wants_grad(x, self.weight, self.bias):
            # training rows: the layer and its gradients on the dense kernels (dense_ops: x W^T, g W, g^T x -- a closed family, so the
            # path-length regulariser's second derivative needs nothing else), both learning-rate gains folded into the kernels
            if x.ndim == 2 and x.dtype == torch.float32:
                from .stylegan_utils import dense_ops
                y = dense_ops.linear(x, self.weight, self.bias, self.weight_gain, self.bias_gain)
                ak = _act_kwargs(self.activation)
                if ak is None:
                    return self.activation(y)
                return grad_ops.bias_act(y, None, **ak) if ak['act'] or ak.get('gain', 1.0) != 1.0 else y
            y = x.matmul((self.weight * self.weight_gain).t())
            ak = _act_kwargs(self.activation)
            b = None if self.bias is None else self.bias * self.bias_gain
            if ak is None:
                return self.activation(y if b is None else y + b)
            return grad_ops.bias_act(y, b, **ak)
        ak = _act_kwargs(self.activation)
        b = self.bias.detach() if self.bias is not None else None
        if ak is None:   # non-fusable activation object
            return self.activation(kernels.dense(x, self.weight.detach(), b, wgain=self.weight_gain, bgain=self.bias_gain))
        return kernels.dense(x, self.weight.detach(), b, wgain=self.weight_gain, bgain=self.bias_gain, out=out, **ak)

    def __repr__(self):
        return self.repr


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, _prepped=None, _epilogue=None, _tail=None):
    """Modulated (and demodulated) convolution, signature of stylegan.py:103-113.

    x [N,I,H,W], weight [O,I,k,k], styles [N,I], noise broadcastable to the output ([H',W'] or
    [N,1,H',W']).  float32: both values of ``fused_modconv`` run the same kernels: the activations are scaled by
    the normalised styles while the input tile is staged, the shared weight is pre-normalised
    (stylegan.py:146), and the demodulation coefficient (stylegan.py:155) is applied in the epilogue.
    float16 without autograd: ``fused_modconv`` selects, as in the reference, whether the per-sample WEIGHT (w * s * d, fused) or the
    ACTIVATION (x * s, then * d: non-fused) is what gets rounded to half.
    ``_prepped`` / ``_epilogue`` are used by the layer classes to pass cached weights and to fuse
    bias / activation / residual."""
    batch_size = x.shape[0]
    out_channels, in_channels, kh, kw = weight.shape
    misc.assert_shape(weight, [out_channels, in_channels, kh, kw])
    misc.assert_shape(x, [batch_size, in_channels, None, None])
    misc.assert_shape(styles, [batch_size, in_channels])
    if (fused_modconv and x.dtype == torch.float16 and x.is_cuda and down == 1 and not _epilogue and _tail is None
            and not grad_ops.wants_grad(x, weight, styles, noise)):
        # the reference's fused form on halves (stylegan.py:149-170,183-193): per-sample weights rounded to half, see _modulated_conv2d_half_infer
        y = _modulated_conv2d_half_infer(x, weight, styles, noise, up, padding, resample_filter, demodulate, flip_weight, None, _act_kwargs(None),
                                         None, fused=True)
        if y is not None:
            return y
    if grad_ops.generic_route(x, weight, styles, noise):
        if _epilogue:
            raise NotImplementedError('modulated_conv2d: the fused epilogue is an inference-path extension')
        return _modulated_conv2d_train(x, weight, styles, noise, up, down, padding, resample_filter, demodulate, flip_weight, _tail)
    ep = dict(_epilogue or {})
    if noise is not None and noise.ndim == 4 and noise.shape[0] == 1:
        noise = noise[0, 0]

    fast_plain = up == 1 and down == 1 and kh == kw and kh in (1, 3) and padding == kh // 2
    fast_up = (up == 2 and down == 1 and kh == kw == 3 and padding == 1 and resample_filter is not None
               and resample_filter.ndim == 2 and tuple(resample_filter.shape) == (4, 4))
    if fast_plain or fast_up:
        pw = _prepped
        if pw is None:
            # plain path: F.conv2d correlation unless flip_weight is False (conv2d_resample.py:32-33);
            # up path: conv2d_resample hands (not flip_weight) to the transposed wrapper (:137)
            pw = kernels.conv_weight_prep(weight, demod=demodulate, flip=(flip_weight if fast_up else not flip_weight))
        s, d = kernels.modconv_style_prep(styles, pw, demod=demodulate)
        if fast_plain:
            return kernels.conv2d(x, pw, mode=kernels.MODE_SAME, pad=padding, in_scale=s, out_scale=d, noise=noise,
                                  bias=ep.get('bias'), residual=ep.get('residual'),
                                  **{k: v for k, v in ep.items() if k in ('act', 'gain', 'alpha', 'act_gain', 'clamp')})
        mid = kernels.conv2d(x, pw, mode=kernels.MODE_UP2T, in_scale=s, planar=True)
        return kernels.upfir_planar(mid, resample_filter, scale=d.reshape(-1) if d is not None else None, noise=noise,
                                    bias=ep.get('bias'), residual=ep.get('residual'),
                                    **{k: v for k, v in ep.items() if k in ('act', 'gain', 'alpha', 'act_gain', 'clamp')})

    # generic geometry: explicit scale -> conv2d_resample -> demod/noise epilogue (stylegan.py:172-181)
    pw_tmp = kernels.conv_weight_prep(weight, demod=demodulate)   # only for wsq / the normalised weight scale
    s, d = kernels.modconv_style_prep(styles, pw_tmp, demod=demodulate)
    wn = weight
    if demodulate:
        wn = weight * weight.square().mean([1, 2, 3], keepdim=True).rsqrt()
    y = conv2d_resample.conv2d_resample(x=kernels.scale_channels(x, s.reshape(-1)), w=wn, f=resample_filter, up=up, down=down,
                                        padding=padding, flip_weight=flip_weight)
    return kernels.bias_act(y, scale=d.reshape(-1) if d is not None else None, noise=noise, bias=ep.get('bias'),
                            residual=ep.get('residual'), act=ep.get('act', False), gain=ep.get('gain', 1.0))


class _DemodWeightFn(torch.autograd.Function):
    """(wn, wsq) of ``_weight_factors`` for a demodulated float32 weight in ONE kernel each way (csrc/dense.hip); under ``create_graph``
    the backward is composed from tensor ops instead (never needed by the shipped losses: the path-length regulariser reaches the
    weights only in its second, ordinary backward pass)."""
    @staticmethod
    def forward(ctx, weight, prenorm):
        wn, wsq, sfac = kernels.demod_weight(weight.detach(), prenorm)
        ctx.save_for_backward(wn, sfac)
        ctx.mark_non_differentiable(sfac)
        return wn, wsq, sfac

    @staticmethod
    def backward(ctx, gwn, gwsq, _gs):
        wn, sfac = ctx.saved_tensors
        if torch.is_grad_enabled():
            g = 0
            if gwn is not None:
                g = g + gwn
            if gwsq is not None:
                g = g + 2 * gwsq[:, :, None, None] * wn
            s = sfac.reshape(-1, 1, 1, 1)
            return s * (g - wn * (g * wn).mean([1, 2, 3], keepdim=True)), None
        return kernels.demod_weight_backward(wn, sfac, None if gwn is None else gwn.contiguous(), None if gwsq is None else gwsq.contiguous()), None


FUSED_DEMOD_WEIGHT = os.environ.get('SHG_FUSED_DEMOD', '1') == '1'        # (A/B switch; SHG_FUSED_DEMOD=0: the tensor-op composition)


def _weight_factors(half, weight, demodulate):
    """The weight side of stylegan.py:136-155 (a function of the parameter alone: the no-grad routes cache it per parameter version):
    (normalised weight, sum_k w^2 [O,I] | None)."""
    if (FUSED_DEMOD_WEIGHT and demodulate and weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4
            and grad_ops.wants_grad(weight)):
        wn, wsq, _ = _DemodWeightFn.apply(weight.contiguous(), bool(half))
        return wn, wsq
    if half and demodulate:
        o, i, kh, kw = weight.shape
        weight = weight * (1 / np.sqrt(i * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))       # max_Ikk, :137
    wsq = None
    if demodulate:
        weight = weight * weight.square().mean([1, 2, 3], keepdim=True).rsqrt()          # stylegan.py:146
        wsq = weight.square().sum([2, 3])                                                  # [O, I]
    return weight, wsq


def _style_factors_composed(half, styles, wsq):
    """The style side of a demodulated layer (stylegan.py:138,147,155) from differentiable tensor operators."""
    if half:
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)                                              # max_I, :138
    styles = styles * styles.square().mean().rsqrt()                                       # :147
    if styles.is_cuda and styles.dtype == torch.float32 and styles.ndim == 2:
        from .stylegan_utils import dense_ops
        return styles, (dense_ops.nt(styles.square(), wsq) + 1e-8).rsqrt()                 # :155, [N,O] -- s^2 @ wsq^T on the dense kernels
    return styles, (styles.square().matmul(wsq.t()) + 1e-8).rsqrt()


class _StyleFactorsFn(torch.autograd.Function):
    """(normalised styles, demodulation coefficients) in ONE kernel, their first-order backward in two (csrc/dense.hip) -- instead of ~11
    + ~25 launches of 4 us per layer and pass.  Under ``create_graph`` (path-length regulariser: the gradient with respect to the styles is
    differentiated again) the backward re-derives both outputs from the saved INPUTS with the composed operators and differentiates that,
    so it stays a differentiable function of the styles, the weights and the incoming gradients."""
    @staticmethod
    def forward(ctx, styles, wsq, half):
        sn, d, aux = kernels.style_factors(styles.detach(), wsq.detach(), half)
        ctx.save_for_backward(styles, wsq, sn, d, aux)
        ctx.half = half
        return sn, d

    @staticmethod
    def backward(ctx, gsn, gd):
        styles, wsq, sn, d, aux = ctx.saved_tensors
        if torch.is_grad_enabled() and CLOSED_STYLE_FACTORS_BACKWARD:
            gs, gw = _StyleFactorsBwdFn.apply(styles, wsq, gsn, gd, sn, d, aux, ctx.half)
            return (gs if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), None
        if torch.is_grad_enabled():
            with torch.enable_grad():
                s_in = styles if styles.requires_grad else styles.detach().requires_grad_(True)
                w_in = wsq if wsq.requires_grad else wsq.detach().requires_grad_(True)
                sn2, d2 = _style_factors_composed(ctx.half, s_in, w_in)
                gs, gw = torch.autograd.grad([sn2, d2], [s_in, w_in], [gsn, gd], create_graph=True, allow_unused=True)
            return (gs if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), None
        gs, gw = kernels.style_factors_backward(sn, d, wsq, aux, gsn, gd, ctx.half, want_wsq=ctx.needs_input_grad[1])
        return (gs if ctx.needs_input_grad[0] else None), gw, None


class _StyleFactorsBwdFn(torch.autograd.Function):
    """The first-order backward of ``_StyleFactorsFn`` as ONE differentiable node (path-length regulariser: the gradient with respect to
    the styles is differentiated again).  Forward = the two backward kernels.  Backward = the closed form of the second derivative,
    about forty small launches on this package's kernels, instead of re-deriving (sn, d) from tensor operators under ``create_graph``
    and letting autograd differentiate ~36 recorded operators per layer twice (~1 300 launches of a Greg pass).

    With m = mean(s^2), r = m^-1/2, c = r^3 / M, sn = r s, u = sn^2, e = u W^T + eps, d = e^-1/2 and incoming (gsn, gd):
        q = -1/2 gd d^3,  t = gsn + 2 sn (q W),  gs = r t - c <t,s> s,  gW = q^T u                     (the first-order backward)
    and for cotangents (a, b) of (gs, gW), with alpha = <a,s>, tau = <t,s>, kappa = <t,a>, abar = r a - c alpha s:
        h = (2 sn abar) W^T + u b^T,  ggsn = abar,  ggd = -1/2 d^3 h,  p = 3/4 gd d^5 h,
        g_sn = 2 abar (q W) + 2 sn (q b) + 2 sn (p W),
        gs2 = r g_sn - c <g_sn,s> s - c kappa s + 3 r^5 / M^2 alpha tau s - c tau a - c alpha t,  gW2 = q^T (2 sn abar) + p^T u.
    float16 rows divide each sample's styles by their max-norm first (stylegan.py:138): the chain rule through s / max|s| (one-hot at the
    arg-max) is applied on both sides.  Checked against torch's float64 double backward (tests/test_gpu_train_graph.py)."""
    @staticmethod
    def forward(ctx, styles, wsq, gsn, gd, sn, d, aux, half):
        gs, gw = kernels.style_factors_backward(sn, d, wsq.detach(), aux, gsn.detach().contiguous(), gd.detach().contiguous(), half, want_wsq=True)
        ctx.save_for_backward(styles, wsq, gsn, gd)
        ctx.half = half
        return gs, gw

    @staticmethod
    def backward(ctx, a, b):
        styles, wsq, gsn, gd = ctx.saved_tensors
        if torch.is_grad_enabled():          # third order: differentiate the composed operators
            with torch.enable_grad():
                ins = [t if t.requires_grad else t.detach().requires_grad_(True) for t in (styles, wsq, gsn, gd)]
                sn2, d2 = _style_factors_composed(ctx.half, ins[0], ins[1])
                gs, gw = torch.autograd.grad([sn2, d2], ins[:2], [ins[2], ins[3]], create_graph=True)
                out = torch.autograd.grad([gs, gw], ins, [a, b], create_graph=True, allow_unused=True)
            return (*[o if need else None for o, need in zip(out, ctx.needs_input_grad[:4])], None, None, None, None)
        nt, nn = kernels.dense, kernels.matmul_nn
        s, W = styles.detach(), wsq.detach()
        gsn, gd = gsn.detach(), gd.detach()
        if ctx.half:
            mu, k = s.abs().max(dim=1, keepdim=True)
            sig = torch.gather(s, 1, k).sign()
            s0, a0, ak = s, a, torch.gather(a, 1, k)
            a = a / mu - s * (sig * ak / mu ** 2)
            s = s / mu
        m_count = s.numel()
        r = s.square().mean().rsqrt()
        c = r ** 3 / m_count
        sn = s * r
        u = sn * sn
        d = (nt(u, W) + 1e-8).rsqrt()
        d3 = d ** 3
        q = -0.5 * gd * d3
        qp_w = None
        alpha = (a * s).sum()
        abar = r * a - (c * alpha) * s
        sa2 = 2 * sn * abar
        h = nt(torch.cat([sa2, u], 1), torch.cat([W, b], 1))                    # (2 sn abar) W^T + u b^T in one launch
        ggd = -0.5 * d3 * h
        pq = 0.75 * gd * d3 * d * d * h
        qp_w = nn(torch.cat([q, pq]), W)                                      # [q; p] W
        n = s.shape[0]
        q_w, p_w = qp_w[:n], qp_w[n:]
        t = gsn + 2 * sn * q_w
        tau, kappa = (t * s).sum(), (t * a).sum()
        g_sn = 2 * (abar * q_w + sn * (nn(q, b) + p_w))
        gs2 = r * g_sn - (c * ((g_sn * s).sum() + kappa) - (3 * r ** 5 / m_count ** 2) * alpha * tau) * s - (c * tau) * a - (c * alpha) * t
        gw2, _ = kernels.matmul_tn(torch.cat([q, pq]), torch.cat([sa2, u]))   # q^T (2 sn abar) + p^T u
        if ctx.half:
            g1 = r * t - (c * tau) * s                                        # first-order gradient in the pre-normalised space
            corr = (-sig * ((gs2 * s0).sum(1, keepdim=True) + (a0 * g1).sum(1, keepdim=True)) / mu ** 2
                    + 2 * ak * (s0 * g1).sum(1, keepdim=True) / mu ** 3)
            gs2 = (gs2 / mu - (sig * ak / mu ** 2) * g1).scatter_add(1, k, corr)
        need = ctx.needs_input_grad
        return (gs2 if need[0] else None), (gw2 if need[1] else None), (abar if need[2] else None), (ggd if need[3] else None), None, None, None, None


CLOSED_STYLE_FACTORS_BACKWARD = os.environ.get('SHG_CLOSED_STYLE_BWD', '1') == '1'    # (A/B switch; 0: the composed double backward)
FUSED_STYLE_FACTORS = os.environ.get('SHG_FUSED_STYLE', '1') == '1'       # (A/B switch; SHG_FUSED_STYLE=0: the tensor-op composition)


def _modulation_factors(half, weight, styles, demodulate, wfac=None):
    """(normalised weight, normalised styles, demodulation coefficients [N,O] | None) of stylegan.py:136-155; ``wfac`` = a cached
    ``_weight_factors`` result."""
    dcoefs = None
    weight, wsq = wfac if wfac is not None else _weight_factors(half, weight, demodulate)
    if (FUSED_STYLE_FACTORS and demodulate and grad_ops.wants_grad(styles, wsq) and kernels.style_factors_supported(styles, wsq)
            and wsq.dtype == torch.float32):
        styles, dcoefs = _StyleFactorsFn.apply(styles.contiguous(), wsq.contiguous(), bool(half))
        return weight, styles, dcoefs
    if half and demodulate:
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)                                              # max_I, :138
    if demodulate:
        styles = styles * styles.square().mean().rsqrt()                                   # :147
        if styles.is_cuda and styles.dtype == torch.float32 and styles.ndim == 2:
            from .stylegan_utils import dense_ops
            dcoefs = (dense_ops.nt(styles.square(), wsq) + 1e-8).rsqrt()                      # :155, [N,O] -- s^2 @ wsq^T on the dense kernels
        else:
            dcoefs = (styles.square().matmul(wsq.t()) + 1e-8).rsqrt()
    return weight, styles, dcoefs


F16_INFER_FUSED = True       # half layers without autograd: modulation / tail fused into the fp16 convolution (False: the composed route)

# ---- the one place that decides which implementation a layer call takes ---------------------------------------------------------
#   'f32_fused'   float32 activations, no gradient requested: the fused inference kernels (weights prepared once per parameter version,
#                 styles / demodulation / noise / bias / lrelu_agc / skip-add inside the convolution or FIR kernel)
#   'f16_fused'   float16 activations (a `use_fp16` block), no gradient requested: the same fusion on the NHWC fp16-MFMA kernels
#   'generic'     a gradient is requested (training rows: the differentiable operators of stylegan_utils, first and second order), or a
#                 float16 call with F16_INFER_FUSED switched off
# ROUTE_TRACE: set to a list to record (module class, route, dtype, activation shape) per layer call -- "which kernel family ran"
# without a profiler (tests/test_gpu_fp16.py, tools/).
ROUTE_TRACE = None


def fused_modconv_rule(module, x, fused_modconv):
    """The blocks' choice of the modulated-convolution form (stylegan.py:488-490, comodgan.py:240-242,307-309): unless the caller fixes it,
    fused in eval for float32 blocks and for a batch of ONE image, else the non-fused algebra.  On this path float32 layers run the same
    kernels either way (``modulated_conv2d``); for float16 blocks the flag selects where the half rounding falls (weights or
    activations), as in the reference."""
    if fused_modconv is None:
        return (not module.training) and (x.dtype == torch.float32 or int(x.shape[0]) == 1)
    return bool(fused_modconv)


def layer_route(module, x, *tensors):
    half = isinstance(x, torch.Tensor) and x.dtype == torch.float16
    if grad_ops.wants_grad(x, *tensors):
        route = 'generic'
    elif half:
        route = 'f16_fused' if F16_INFER_FUSED else 'generic'
    else:
        route = 'f32_fused'
    if ROUTE_TRACE is not None:
        ROUTE_TRACE.append((type(module).__name__, route, str(x.dtype).replace('torch.', ''), tuple(x.shape)))
    return route


def _modulated_conv2d_half_infer(x, weight, styles, noise, up, padding, resample_filter, demodulate, flip_weight, bias, ak, residual, cache=None,
                                 fused=False):
    """Inference route of a float16 modulated layer (no autograd): the same algebra as ``_modulated_conv2d_train`` with the passes fused
    into the NHWC fp16 convolution -- ``x * styles`` while the patch is staged, demodulation / noise / bias / lrelu_agc / skip-add in its
    store pass (up = 1); the transposed form (up = 2) takes the style scale at staging, its tail follows the FIR in one modtail pass.
    Returns None when the geometry is not one of the layer forms (caller falls back to the composed route).
    ``fused`` = the reference's ``fused_modconv`` form on halves (stylegan.py:149-170,183-193; what its blocks take in eval for a batch of
    ONE image, :490): the weight is modulated and demodulated per sample in float32 and rounded to half once, the activations stay as
    they are -- there a grouped convolution with groups = N, here one launch per image with that image's weight (same arithmetic)."""
    from .. import kernels_f16
    k = weight.shape[2]
    if ak is None or weight.shape[2] != weight.shape[3] or k not in (1, 3) or padding != k // 2 or up not in (1, 2):
        return None
    if up == 2 and not (k == 3 and resample_filter is not None and resample_filter.ndim == 2 and tuple(resample_filter.shape) == (4, 4)):
        return None
    if weight.shape[0] % 8 and up == 2:
        return None              # the NHWC tail pass after the FIR (modtail) moves 8 channels per lane: narrower outputs take the composed route
                                 # (up = 1 stores any width: toRGB's 3 channels run fused, tests/test_gpu_fp16.py)
    # the weight side (normalisation, sum of squares, operand-order packing) depends on the parameter alone: cached per parameter version
    # when the caller hands its layer cache (the float32 route does the same with `prepped()`)
    def build():
        wn_, wsq_ = _weight_factors(True, weight, demodulate)
        if up == 1:
            pk = kernels_f16.pack_weight((wn_ if flip_weight else wn_.flip([2, 3])).to(torch.float16))
        else:
            pk = kernels_f16.pack_weight((wn_.flip([2, 3]) if flip_weight else wn_).transpose(0, 1).to(torch.float16), transposed=True)
        return wn_, wsq_, pk
    wn, wsq, pk = cache.get(f'w16_{up}_{int(bool(flip_weight))}_{int(bool(demodulate))}', [weight], build) if cache is not None else build()
    _, sn, d = _modulation_factors(True, weight, styles, demodulate, wfac=(wn, wsq))
    if fused:
        outs = []
        per_sample_noise = noise is not None and noise.ndim == 4 and noise.shape[0] == x.shape[0] and x.shape[0] > 1
        tail = ak if (ak.get('act') or float(ak.get('gain', 1.0)) != 1.0) else {}         # (the bare operator: no activation pass)
        for j in range(x.shape[0]):
            wj = wn * sn[j].reshape(1, -1, 1, 1)                                    # :150-151
            if d is not None:
                wj = wj * d[j].reshape(-1, 1, 1, 1)                                 # :169
            nz = noise[j:j + 1] if per_sample_noise else noise
            rj = None if residual is None else residual[j:j + 1]
            if up == 1:
                pkj = kernels_f16.pack_weight((wj if flip_weight else wj.flip([2, 3])).to(torch.float16))
                outs.append(kernels_f16.conv2d(x[j:j + 1], pkj, bias, 1, padding, noise=nz, residual=rj, **tail))
                continue
            pkj = kernels_f16.pack_weight((wj.flip([2, 3]) if flip_weight else wj).transpose(0, 1).to(torch.float16), transposed=True)
            mid = kernels_f16.conv_transpose2d(x[j:j + 1], pkj, None, 0, None)
            mid = kernels_f16.upfirdn2d(mid, resample_filter, padx0=1, padx1=1, pady0=1, pady1=1, gain=4.0)
            yj = kernels_f16.modtail(mid, noise=nz, bias=bias, **ak)
            outs.append(yj if rj is None else yj + rj)
        return outs[0] if len(outs) == 1 else torch.cat(outs).contiguous(memory_format=torch.channels_last)
    if up == 1:
        return kernels_f16.conv2d(x, pk, bias, 1, padding, in_scale=sn, out_scale=d, noise=noise, residual=residual, **ak)
    # conv2d_resample.py:122-142 with up = 2, padding = 1, a 4x4 filter: conv_transpose2d(stride 2, padding 0) -> FIR pad [1,1,1,1], gain 4
    mid = kernels_f16.conv_transpose2d(x, pk, None, 0, None, in_scale=sn)
    mid = kernels_f16.upfirdn2d(mid, resample_filter, padx0=1, padx1=1, pady0=1, pady1=1, gain=4.0)
    y = kernels_f16.modtail(mid, d=d, noise=noise, bias=bias, **ak)
    return y if residual is None else y + residual


def _modulated_conv2d_train(x, weight, styles, noise, up, down, padding, resample_filter, demodulate, flip_weight, tail=None):
    """Training rows and every float16 layer: the non-fused form of stylegan.py:172-181 (what the reference runs while training, and
    for fp16 batches in eval: ``fused_modconv = (not training) and (fp32 or N == 1)``, :490) on differentiable operators -- activations
    scaled by the styles, ONE shared-weight convolution (HIP forward / backward), demodulation coefficient and noise applied
    afterwards.  The coefficient d[n,o] = rsqrt(sum_{i,k} (W[o,i,k] s[n,i])^2 + 1e-8) is evaluated as rsqrt(s^2 @ (sum_k W^2)^T + 1e-8):
    the [N,O,I,k,k] tensor of stylegan.py:150-155 is never materialised.  float16: the pre-normalisation of :136-138 (weights by their
    max-norm and 1/sqrt(fan-in), styles by their max-norm -- it cancels under demodulation and keeps x*s and the accumulators inside
    the fp16 range), then styles / weights / coefficients / noise are cast to the activation dtype exactly where the reference casts."""
    n = x.shape[0]
    weight, styles, dcoefs = _modulation_factors(x.dtype == torch.float16, weight, styles, demodulate)
    fuse = FUSED_F16_TAIL and grad_ops.modtail_supported(x)
    x = grad_ops.modconv_tail(x, d=styles) if fuse else x * styles.to(x.dtype).reshape(n, -1, 1, 1)       # (one pass each way incl. d/ds)
    x = conv2d_resample.conv2d_resample(x=x, w=weight.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
    if tail is not None:
        # `tail` = (bias, activation kwargs) of the calling layer: demodulation, noise, bias and lrelu_agc in ONE pass each way
        # (grad_ops.modconv_tail) where the fused half kernels apply; else the per-operation form below + the layer's own bias_act
        bias, ak = tail
        if FUSED_F16_TAIL and grad_ops.modtail_supported(x) and ak is not None:
            return grad_ops.modconv_tail(x, d=dcoefs, noise=noise, bias=bias, **ak)
        y = _modulated_tail_unfused(x, n, dcoefs, noise, demodulate)
        if ak is None or (y.dtype == torch.float16 and y.shape[1] % 8):
            # activation objects the kernels do not know, and half tensors whose channel count is no multiple of 8 (custom widths: the
            # fp16 kernels move 8 channels per lane): the per-operation form, as conv2d_layer._forward_train does
            if bias is not None:
                y = y + bias.view(1, -1, 1, 1).to(y.dtype)
            return y if ak is None else _act_generic(y, ak)
        return grad_ops.bias_act(y, bias, **ak)
    return _modulated_tail_unfused(x, n, dcoefs, noise, demodulate)


FUSED_F16_TAIL = True        # (A/B switch: False = one tensor pass per operation, the round-3 first form; float16 and float32 training routes)


def _modulated_tail_unfused(x, n, dcoefs, noise, demodulate):
    if x.dtype == torch.float16:
        # (tensor ops that keep the NHWC layout of x: the product first, the broadcast noise added in place)
        if demodulate:
            x = x * dcoefs.to(x.dtype).reshape(n, -1, 1, 1)
        return x if noise is None else x.add(noise.to(x.dtype))
    if demodulate and noise is not None:
        return torch.addcmul(noise, x, dcoefs.reshape(n, -1, 1, 1))                        # fma.py:15
    if demodulate:
        return x * dcoefs.reshape(n, -1, 1, 1)
    return x if noise is None else x + noise


class conv2d_layer(nn.Module):
    """conv (+FIR up/down) -> +bias -> activation*gain   (stylegan.py:195-241)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation=None, up=1, down=1,
                 resample_filter=[1, 3, 3, 1]):
        super().__init__()
        self.up = up
        self.down = down
        if resample_filter is not None:
            self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        else:
            self.resample_filter = None
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.activation = _make_act(activation)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels])) if bias else None
        self.repr = 'conv2d_layer({}, {}, kernal_size={}, bias={}, up={}, down={}, act={})'.format(
            in_channels, out_channels, kernel_size, bias, up, down, activation)

    def prepped(self):
        # flip_weight = (up == 1): plain/strided convs are correlations, weights used as stored
        return _cache_of(self).get('w', [self.weight], lambda: kernels.conv_weight_prep(
            self.weight.detach(), gain=self.weight_gain))

    def _forward_train(self, x, gain=1, residual=None):
        """Differentiable composition (training rows): convolution / FIR / bias + activation through their autograd forms.  ``residual``: the
        other branch of a residual block, added in the store pass of this (linear) layer where that form exists, by ``_add`` otherwise."""
        if residual is not None:
            ak = _act_kwargs(self.activation, gain)
            if (FUSE_BLOCK_SUM and self.activation is None and self.up == 1 and self.down == 2 and self.padding == 0 and self.weight.shape[2] == 1
                    and self.bias is None and self.resample_filter is not None and tuple(self.resample_filter.shape) == (4, 4) and ak is not None
                    and not ak.get('act') and conv2d_gradfix.conv_bias_act_supported(x, self.weight, ak) and residual.dtype == x.dtype):
                xd = upfirdn2d.upfirdn2d(x=x, f=self.resample_filter, down=2, padding=[1, 1, 1, 1])
                if tuple(residual.shape) == (x.shape[0], self.weight.shape[0], xd.shape[2], xd.shape[3]):
                    return conv2d_gradfix.conv2d_bias_act(xd, grad_ops.scaled_weight(self.weight, self.weight_gain, x.dtype), None, 0, residual=residual, **ak)
            return _add(residual, self._forward_train(x, gain))
        ak = _act_kwargs(self.activation, gain)
        if self.up == 1 and self.down == 1 and conv2d_gradfix.conv_bias_act_supported(x, self.weight, ak):
            # 3x3 stride-1 layers: convolution + bias + activation as one node on one forward kernel
            return conv2d_gradfix.conv2d_bias_act(x, grad_ops.scaled_weight(self.weight, self.weight_gain, x.dtype), self.bias, self.padding, **ak)
        if (self.up == 1 and self.down == 2 and self.padding == 0 and self.weight.shape[2] == 1 and self.resample_filter is not None
                and tuple(self.resample_filter.shape) == (4, 4) and conv2d_gradfix.conv_bias_act_supported(x, self.weight, ak)):
            # the residual blocks' skip: decimate with the low-pass (conv2d_resample.py:104-108), then the 1x1 with its gain in the store pass
            xd = upfirdn2d.upfirdn2d(x=x, f=self.resample_filter, down=2, padding=[1, 1, 1, 1])
            return conv2d_gradfix.conv2d_bias_act(xd, grad_ops.scaled_weight(self.weight, self.weight_gain, x.dtype), self.bias, 0, **ak)
        if self.up == 1 and self.down == 2 and self.padding == 1:
            y = conv2d_resample.conv2d_down_bias_act(x, grad_ops.scaled_weight(self.weight, self.weight_gain, x.dtype), self.resample_filter, self.bias, ak)
            if y is not None:
                return y
        y = conv2d_resample.conv2d_resample(x=x, w=grad_ops.scaled_weight(self.weight, self.weight_gain, x.dtype), f=self.resample_filter, up=self.up,
                                            down=self.down, padding=self.padding, flip_weight=(self.up == 1))
        if ak is None or (y.dtype == torch.float16 and y.shape[1] % 8):
            if self.bias is not None:
                y = grad_ops.add_channel_bias(y, self.bias)
            return self.activation(y, gain=gain) if self.activation is not None else y * gain
        return grad_ops.bias_act(y, self.bias, **ak)

    def _forward_half_infer(self, x, gain):
        """float16 activations without autograd: bias + lrelu_agc fused into the store pass of the fp16 convolution (plain and FIR-filtered
        stride-2 3x3 forms, 1x1); None for the other geometries (the composed route takes them)."""
        from .. import kernels_f16
        ak = _act_kwargs(self.activation, gain)
        k = self.weight.shape[2]
        if ak is None or self.up != 1 or k not in (1, 3) or self.weight.shape[0] % 8:
            return None
        w = _cache_of(self).get('w16', [self.weight], lambda: kernels_f16.pack_weight((self.weight.detach() * self.weight_gain).to(torch.float16)))
        b = None if self.bias is None else self.bias.detach()
        if self.down == 1:
            return kernels_f16.conv2d(x, w, b, 1, self.padding, **ak)
        f = self.resample_filter
        if self.down == 2 and k == 3 and self.padding == 1 and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4):
            y = kernels_f16.upfirdn2d(x, f, padx0=2, padx1=2, pady0=2, pady1=2)             # conv2d_resample.py:116-120
            return kernels_f16.conv2d(y, w, b, 2, 0, **ak)
        return None

    def forward(self, x, gain=1, residual=None):
        if residual is not None:                  # (training rows of the critic's residual blocks; elsewhere the plain sum)
            if layer_route(self, x, self.weight, self.bias) == 'generic':
                return self._forward_train(x, gain, residual=residual)
            return _add(residual, self.forward(x, gain))
        route = layer_route(self, x, self.weight, self.bias)
        if route == 'f16_fused':
            y = self._forward_half_infer(x, gain)
            if y is not None:
                return y
        if route != 'f32_fused':                 # 'generic', or a half geometry the fused half route does not serve
            return self._forward_train(x, gain)
        ak = _act_kwargs(self.activation, gain)
        b = self.bias.detach() if self.bias is not None else None
        k = self.weight.shape[2]
        fusable = ak is not None and self.up == 1 and self.down in (1, 2) and k in (1, 3)
        if fusable and self.down == 2 and not (k == 3 and self.resample_filter is not None and self.resample_filter.ndim == 2):
            fusable = False
        if fusable:
            if self.down == 1:
                if k == 1 and self.weight.shape[1] <= 8 and (x.shape[2] * x.shape[3]) % 4 == 0:
                    return kernels.conv1x1_thin_in(x, self.weight.detach().reshape(self.weight.shape[0], -1), b,
                                                   wgain=self.weight_gain, **ak)
                return kernels.conv2d(x, self.prepped(), mode=kernels.MODE_SAME, pad=self.padding, bias=b, **ak)
            # low-pass with the resample filter, then the stride-2 convolution (conv2d_resample.py:116-120)
            f = self.resample_filter
            fw, fh = f.shape[1], f.shape[0]
            p = self.padding
            pads = [p + (fw - 1) // 2, p + (fw - 2) // 2, p + (fh - 1) // 2, p + (fh - 2) // 2]
            if tuple(f.shape) == (4, 4) and p == 1 and kernels.down_poly_supported(x, self.prepped()):
                return kernels.fir_conv_down2(x, f, self.prepped(), bias=b, **ak)      # polyphase-Winograd form
            y = kernels.upfirdn2d(x, f, padx0=pads[0], padx1=pads[1], pady0=pads[2], pady1=pads[3])
            return kernels.conv2d(y, self.prepped(), mode=kernels.MODE_DOWN2, pad=0, bias=b, **ak)
        # generic composition
        w = self.weight.detach() * self.weight_gain
        y = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        if ak is None:
            if b is not None:
                y = kernels.bias_act(y, bias=b, act=False)
            return self.activation(y, gain=gain)
        return kernels.bias_act(y, bias=b, **ak)

    def __repr__(self):
        return self.repr


class synthesis_layer(conv2d_layer):
    """affine(w) -> modulated conv (x2 upsampling when up == 2) -> +noise -> +bias -> activation
    (stylegan.py:243-304).  ``residual`` (extension) is added after the activation so the skip
    connection of comodgan.py:320-327 costs no extra pass."""

    def __init__(self, in_channels, out_channels, kernel_size, w_dim, resolution, bias=True,
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2)', up=1, resample_filter=[1, 3, 3, 1], use_noise=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, activation=activation, up=up, down=1,
                         resample_filter=resample_filter)
        self.affine = dense(w_dim, in_channels, bias=True, bias_init=1, activation=None)
        self.resolution = resolution
        self.use_noise = use_noise
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))
        self.repr = 'synthesis_layer({}, {}, kernal_size={}, bias={}, up={}, act={}, noise={})'.format(
            in_channels, out_channels, kernel_size, bias, up, activation, use_noise)

    def prepped(self):
        # demodulated layers ignore weight_gain (it cancels, stylegan.py:289-294); the up path is a
        # transposed conv called with flip_weight=False -> no explicit flip (conv2d_resample.py:137)
        return _cache_of(self).get('w', [self.weight], lambda: kernels.conv_weight_prep(self.weight.detach(), demod=True))

    def _noise_strength_host(self):
        return _cache_of(self).get('ns', [self.noise_strength], lambda: float(self.noise_strength.detach().cpu()))

    def forward(self, x, w, fused_modconv=True, gain=1, noise_mode='random', residual=None, styles_sd=None):
        """``styles_sd`` (extension): the (normalised styles, demodulation coefficients) pair of this layer when the
        caller has already computed it for all layers at once (comodgan.Synthesis); ``w`` is then unused."""
        if noise_mode not in ('random', 'const', 'none'):
            raise AssertionError(f'bad noise_mode {noise_mode!r}')
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)
        elif self.use_noise and noise_mode == 'const':
            noise = self.noise_const
        ak = _act_kwargs(self.activation, gain)
        if ak is None or self.up not in (1, 2) or self.weight.shape[2] != 3:
            raise NotImplementedError('synthesis_layer: HIP path needs lrelu_agc, 3x3 kernels and up in {1,2}')
        route = layer_route(self, x, w, self.weight, self.bias, self.affine.weight, residual)
        if route == 'f16_fused':
            y = _modulated_conv2d_half_infer(x, self.weight.detach(), self.affine(w), None if noise is None else noise * self.noise_strength.detach(),
                                             self.up, self.padding, self.resample_filter, True, self.up == 1, self.bias.detach(), ak, residual,
                                             cache=_cache_of(self), fused=bool(fused_modconv))
            if y is not None:
                return y
        if route != 'f32_fused':
            # training rows and float16 layers (stylegan.py:276-304): styles from the affine layer, noise scaled by its learnt strength, the
            # non-fused modulated convolution, bias + activation; the skip tensor (extension) is added last
            y = modulated_conv2d(x=x, weight=self.weight, styles=self.affine(w), noise=None if noise is None else noise * self.noise_strength,
                                 up=self.up, padding=self.padding, resample_filter=self.resample_filter, flip_weight=(self.up == 1),
                                 _tail=(self.bias, ak))
            return y if residual is None else y + residual
        ns = 0.0
        if noise is not None:
            if x.is_cuda and torch.cuda.is_current_stream_capturing():
                # inside a HIP-graph capture (train_stage.PhaseGraphs: the generator's no-grad pass of Dmain) a kernel argument is
                # frozen and a host read is illegal: the learnt strength is applied on the device instead
                noise, ns = noise * self.noise_strength.detach(), 1.0
            else:
                ns = self._noise_strength_host()
        pw = self.prepped()
        if styles_sd is not None:
            s, d = styles_sd
        else:
            s, d = kernels.modconv_style_prep(self.affine(w), pw, demod=True)
        b = self.bias.detach()
        if self.up == 1:
            return kernels.conv2d(x, pw, mode=kernels.MODE_SAME, pad=self.padding, in_scale=s, out_scale=d, noise=noise,
                                  noise_strength=ns, bias=b, residual=residual, **ak)
        mid = kernels.conv2d(x, pw, mode=kernels.MODE_UP2T, in_scale=s, planar=True)      # four phase planes
        return kernels.upfir_planar(mid, self.resample_filter, scale=d.reshape(-1), noise=noise, noise_strength=ns, bias=b,
                                    residual=residual, **ak)


class torgb_layer(conv2d_layer):
    """Modulated 1x1 conv to RGB without demodulation (stylegan.py:306-337).  ``base_img`` /
    ``base_filter`` (extension) fuse ``upsample2d(img) + torgb(x)`` of the skip architecture."""

    def __init__(self, in_channels, out_channels, kernel_size, w_dim, activation=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=True, activation=activation, up=1, down=1,
                         resample_filter=None)
        self.affine = dense(w_dim, in_channels, bias=True, bias_init=1, activation=None)

    def forward(self, x, w, fused_modconv=True, base_img=None, base_filter=None, styles_sd=None):
        if self.activation is not None or self.weight.shape[2] != 1 or self.weight.shape[0] > 4:
            raise NotImplementedError('torgb_layer: HIP path is the 1x1, <=4-channel, linear form')
        route = layer_route(self, x, w, self.weight, self.bias, self.affine.weight, base_img)
        if route == 'f16_fused':
            from .. import kernels_f16
            if fused_modconv:         # stylegan.py:149-151,183-193 without demodulation: per-sample weights w * s rounded to half, one image per launch
                st = self.affine(w) * self.weight_gain
                ys = [kernels_f16.conv2d(x[j:j + 1], (self.weight.detach() * st[j].reshape(1, -1, 1, 1)).to(torch.float16), self.bias.detach(), 1, 0)
                      for j in range(x.shape[0])]
                y = ys[0] if len(ys) == 1 else torch.cat(ys)
            else:
                y = kernels_f16.conv2d(x, self.weight.detach().to(torch.float16), self.bias.detach(), 1, 0, in_scale=self.affine(w) * self.weight_gain)
            y = y.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            return y if base_img is None else upfirdn2d.upsample2d(base_img, base_filter) + y
        if route != 'f32_fused':
            # training rows and float16 blocks (stylegan.py:325-337) + the skip architecture's upsample2d(img) + y (comodgan.py:331-338);
            # the RGB branch itself is float32 (`y.to(torch.float32)`, comodgan.py:337)
            y = modulated_conv2d(x=x, weight=self.weight, styles=self.affine(w) * self.weight_gain, demodulate=False)
            y = grad_ops.add_channel_bias(y, self.bias)
            y = y.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            return y if base_img is None else upfirdn2d.upsample2d(base_img, base_filter) + y
        if styles_sd is not None:
            s = styles_sd[0]
        else:
            s, _ = kernels.modconv_style_prep(self.affine(w), None, demod=False, pre_gain=self.weight_gain)
        wmat = self.weight.detach().reshape(self.weight.shape[0], -1)
        return kernels.torgb(x, wmat, s, self.bias.detach(), base_up=base_img, f=base_filter)


# ------------------------------------------------------------------------------------------------
# mapping network
# ------------------------------------------------------------------------------------------------

def normalize_2nd_moment(x, dim=1, eps=1e-8):
    if grad_ops.wants_grad(x):
        return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()                 # stylegan.py:343-344
    if dim != 1 or x.ndim != 2:
        raise NotImplementedError('normalize_2nd_moment: HIP path handles [N,K] along dim 1')
    return kernels.normalize_2nd_moment(x, eps)


@register('stylegan2_mapping', version)
class Mapping(nn.Module):
    """z (and optional label c) -> w, broadcast to num_ws rows, optional truncation (stylegan.py:347-430)."""

    def __init__(self, z_dim=512, c_dim=0, w_dim=512, num_ws=14, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers, self.w_avg_beta = num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = dense(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', dense(widths[idx], widths[idx + 1], activation=activation, lr_multi=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.w_avg_beta is None:
                raise AssertionError('truncation needs w_avg')
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


# ------------------------------------------------------------------------------------------------
# plain StyleGAN2 synthesis (kept for API completeness; the SH-GAN path uses comodgan.Synthesis)
# ------------------------------------------------------------------------------------------------

class synthesis_block(nn.Module):
    def __init__(self, ic_n, oc_n, w_dim, resolution, rgb_n=None, resample_filter=[1, 3, 3, 1],
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', res_link=False, use_fp16=False):
        super().__init__()
        self.w_dim, self.resolution, self.use_fp16, self.res_link = w_dim, resolution, use_fp16, res_link
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        self.const = None
        self.conv0 = None
        if ic_n == 0:
            self.const = nn.Parameter(torch.randn([oc_n, resolution, resolution]))
        else:
            self.conv0 = synthesis_layer(ic_n, oc_n, 3, w_dim=w_dim, resolution=resolution, up=2, activation=activation,
                                         resample_filter=resample_filter, use_noise=True)
            self.num_conv += 1
        self.conv1 = synthesis_layer(oc_n, oc_n, 3, w_dim=w_dim, resolution=resolution, up=1, activation=activation,
                                     resample_filter=None, use_noise=True)
        self.num_conv += 1
        self.torgb = None
        if rgb_n is not None:
            self.torgb = torgb_layer(oc_n, rgb_n, 1, w_dim=w_dim, activation=None)
            self.num_torgb += 1
        if ic_n != 0 and res_link:
            self.skip = conv2d_layer(ic_n, oc_n, kernel_size=1, bias=False, up=2, down=1, resample_filter=resample_filter)

    def forward(self, x, img, ws, fused_modconv=None, noise_mode='random'):
        if self.const is not None:
            x = (self.const if grad_ops.wants_grad(self.const) else self.const.detach()).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        x = grad_ops.to_block_dtype(x, self.use_fp16)                            # stylegan.py:486-495
        fm = fused_modconv_rule(self, x, fused_modconv)                          # :488-490
        if self.res_link:
            y = self.skip(x, gain=np.sqrt(0.5))
        w_iter = iter(ws.unbind(dim=1))
        if self.conv0 is not None:
            x = self.conv0(x, next(w_iter).contiguous(), fused_modconv=fm, noise_mode=noise_mode)
        if self.res_link:
            x = self.conv1(x, next(w_iter).contiguous(), fused_modconv=fm, gain=np.sqrt(0.5), noise_mode=noise_mode, residual=y)
        else:
            x = self.conv1(x, next(w_iter).contiguous(), fused_modconv=fm, noise_mode=noise_mode)
        if self.torgb is not None:
            img = self.torgb(x, next(w_iter).contiguous(), fused_modconv=fm, base_img=img, base_filter=self.resample_filter)
        elif img is not None:
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        return x, img


@register('stylegan2_synthesis', version)
class Synthesis(nn.Module):
    def __init__(self, w_dim=512, resolution=256, rgb_n=3, ch_base=16384, ch_max=512, use_fp16_after_res=16,
                 resample_filter=[1, 3, 3, 1], activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)'):
        super().__init__()
        log2res = int(np.log2(resolution))
        if 2 ** log2res != resolution:
            raise ValueError
        self.w_dim, self.resolution, self.rgb_n = w_dim, resolution, rgb_n
        self.block_res = [2 ** i for i in range(2, log2res + 1)]
        self.num_ws = 0
        for resi, resj in zip([None] + self.block_res[:-1], self.block_res):
            ci = min(ch_base // resi, ch_max) if resi is not None else 0
            cj = min(ch_base // resj, ch_max)
            block = synthesis_block(ci, cj, w_dim=w_dim, resolution=resj, rgb_n=rgb_n, resample_filter=resample_filter,
                                    activation=activation, res_link=False,
                                    use_fp16=(use_fp16_after_res is not None and resj > use_fp16_after_res))          # stylegan.py:550
            self.num_ws += block.num_conv
            if resj == self.block_res[-1]:
                self.num_ws += block.num_torgb
            setattr(self, 'b{}'.format(resj), block)

    def forward(self, ws, noise_mode='random'):
        ws = ws.to(torch.float32)
        x = img = None
        w_idx = 0
        for res in self.block_res:
            block = getattr(self, f'b{res}')
            x, img = block(x, img, ws.narrow(1, w_idx, block.num_conv + block.num_torgb), noise_mode=noise_mode)
            w_idx += block.num_conv
        return img


@register('stylegan2_generator', version)
class Generator(nn.Module):
    """mapping + synthesis; sub-networks are modules or registry configs (stylegan.py:581-606)."""

    def __init__(self, mapping, synthesis):
        super().__init__()
        self.mapping = mapping if isinstance(mapping, nn.Module) else get_model()(mapping)
        self.synthesis = synthesis if isinstance(synthesis, nn.Module) else get_model()(synthesis)
        if self.synthesis.num_ws != self.mapping.num_ws:
            raise ValueError
        self.num_ws = self.mapping.num_ws
        self.z_dim, self.c_dim, self.w_dim = self.mapping.z_dim, self.mapping.c_dim, self.mapping.w_dim
        self.img_resolution = self.synthesis.resolution
        self.img_channels = self.synthesis.rgb_n

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, **synthesis_kwargs)


# ------------------------------------------------------------------------------------------------
# discriminator-style down blocks (the co-modulation encoder is built from these)
# ------------------------------------------------------------------------------------------------

JOIN_INPUT_GRADS = os.environ.get('SHG_JOIN_GRADS', '1') == '1'       # (A/B switch: grad_ops.InputGradJoin in the residual down blocks)


class discrim_block(nn.Module):
    """[fromrgb] -> conv0 3x3 -> conv1 3x3 stride-2 with FIR pre-filter (stylegan.py:624-684)."""

    def __init__(self, ic_n, mc_n, oc_n, rgb_n=None, resample_filter=[1, 3, 3, 1],
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', reslink=False, use_fp16=False):
        super().__init__()
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.fromrgb = None
        if rgb_n is not None:
            self.fromrgb = conv2d_layer(rgb_n, mc_n, 1, bias=True, activation=activation, up=1, down=1, resample_filter=None)
        self.conv0 = conv2d_layer(ic_n, mc_n, 3, bias=True, activation=activation, up=1, down=1, resample_filter=None)
        self.conv1 = conv2d_layer(mc_n, oc_n, 3, bias=True, activation=activation, up=1, down=2,
                                  resample_filter=resample_filter)
        self.reslink = reslink
        if reslink:
            self.skip = conv2d_layer(mc_n, oc_n, 1, bias=False, activation=None, up=1, down=2, resample_filter=resample_filter)
        self.use_fp16 = use_fp16

    def forward(self, x, img):
        x = grad_ops.to_block_dtype(x, self.use_fp16)                            # stylegan.py:659-663
        if self.fromrgb is not None:
            y = self.fromrgb(grad_ops.to_block_dtype(img, self.use_fp16))        # :665-670
            x = _add(x, y) if x is not None else y
        img = None
        if self.reslink:
            if JOIN_INPUT_GRADS and grad_ops.wants_grad(x) and x.is_cuda:
                # training rows: x feeds the skip branch and conv0; the skip branch (called last = first in backward) leaves its input
                # gradient with the join and conv0's input-gradient kernel adds it in its store pass (grad_ops.InputGradJoin)
                join = grad_ops.InputGradJoin(x)
                with grad_ops.InputGradJoin.consumer(join):
                    h = self.conv0(x)
                h = self.conv1(h, gain=np.sqrt(0.5))
                # (the block's sum h + skip(x) in the store pass of the skip's 1x1 GEMM: round 6)
                return self.skip(grad_ops.stash_input_grad(x, join), gain=np.sqrt(0.5), residual=h), None
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv1(self.conv0(x), gain=np.sqrt(0.5))
            x = _add(x, y)
        else:
            x = self.conv1(self.conv0(x))
        return x, img


class minibatch_std_layer(nn.Module):
    """Appends the per-group standard-deviation statistic as ``num_channels`` extra channels (stylegan.py:686-704)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x, segments=1):
        if segments > 1:                     # independent sub-batches stacked along the batch axis: the statistic never mixes them
            if x.shape[0] % segments:
                raise ValueError('minibatch_std_layer: the batch does not split into %d equal segments' % segments)
            return torch.cat([self.forward(part) for part in x.chunk(segments)], dim=0)
        if grad_ops.wants_grad(x):
            # training rows: the statistic is a handful of reductions over a 4x4 map -- composed from differentiable tensor ops
            n, c, h, w = x.shape
            g = n if self.group_size is None else min(int(self.group_size), n)
            f = self.num_channels
            y = x.reshape(g, -1, f, c // f, h, w)
            y = (y - y.mean(dim=0)).square().mean(dim=0)
            y = (y + 1e-8).sqrt().mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1).repeat(g, 1, h, w)
            return torch.cat([x, y], dim=1)
        return kernels.minibatch_std(x, self.group_size, self.num_channels)


class discrim_epilogue(nn.Module):
    """4x4 tail: [minibatch-std] -> conv 3x3 -> fc -> out (stylegan.py:707-755)."""

    def __init__(self, ic_n, resolution, cmap_dim, rgb_n=None, mbstd_group_size=4, mbstd_c_n=1,
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', reslink=True):
        super().__init__()
        self.ic_n, self.cmap_dim, self.resolution, self.rgb_n, self.reslink = ic_n, cmap_dim, resolution, rgb_n, reslink
        self.fromrgb = None
        if rgb_n is not None:
            self.fromrgb = conv2d_layer(rgb_n, ic_n, 1, bias=True, activation=activation, up=1, down=1, resample_filter=None)
        self.mbstd = minibatch_std_layer(group_size=mbstd_group_size, num_channels=mbstd_c_n) if mbstd_c_n > 0 else None
        self.conv = conv2d_layer(ic_n + mbstd_c_n, ic_n, 3, bias=True, activation=activation, up=1, down=1,
                                 resample_filter=None)
        self.fc = dense(ic_n * (resolution ** 2), ic_n, activation=activation)
        self.out = dense(ic_n, 1 if cmap_dim is None else cmap_dim, activation=None)

    def forward(self, x, img=None, cmap=None, segments=1):
        x = grad_ops.to_block_dtype(x, False)                                    # stylegan.py:744: the tail is always float32
        if self.fromrgb is not None:
            x = _add(x, self.fromrgb(img.to(torch.float32)))
        if self.mbstd is not None:
            x = self.mbstd(x, segments=segments)
        x = self.conv(x)
        x = self.out(self.fc(x.flatten(1)))
        if self.cmap_dim is not None:        # conditional projection (stylegan.py:752-753): [N, cmap_dim] x [N, cmap_dim] -> [N, 1]
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        return x


@register('stylegan2_discriminator', version)
class Discriminator(nn.Module):
    """The training-time critic, forward only (stylegan.py:757-838): residual down blocks (3x3 -> 3x3 stride 2, plus the
    FIR-decimated 1x1 skip of conv2d_resample.py:104-108, both scaled by sqrt(.5)) and the minibatch-std epilogue.
    SH-GAN feeds it cat([mask-0.5, image]) (ic_n = 4, comodgan.yaml:51-58)."""

    def __init__(self, resolution=256, ic_n=3, ch_base=16384, ch_max=512, use_fp16_before_res=16, resample_filter=[1, 3, 3, 1],
                 activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', mbstd_group_size=4, mbstd_c_n=1, c_dim=None,
                 cmap_dim=None):
        super().__init__()
        log2res = int(np.log2(resolution))
        if 2 ** log2res != resolution:
            raise ValueError
        self.encode_res = [2 ** i for i in range(log2res, 1, -1)]
        self.ic_n, self.ch_base, self.ch_max = ic_n, ch_base, ch_max
        self.resample_filter, self.activation = resample_filter, activation
        for idx, (ri, rj) in enumerate(zip(self.encode_res[:-1], self.encode_res[1:])):
            ci, cj = min(ch_base // ri, ch_max), min(ch_base // rj, ch_max)
            setattr(self, 'b{}'.format(ri), discrim_block(ci, ci, cj, rgb_n=(ic_n if idx == 0 else None),
                                                          resample_filter=resample_filter, activation=activation, reslink=True,
                                                          use_fp16=(use_fp16_before_res is not None and ri > use_fp16_before_res)))   # stylegan.py:788
        self.mapping = None
        if c_dim is not None and c_dim > 0:
            # the reference cannot build this either: its Mapping passes an unknown keyword to `dense` when c_dim > 0
            # (stylegan.py:377), and its critic builds the epilogue without the projection (cmap_dim=None, :818-825)
            raise NotImplementedError('label-conditioned critics: the reference constructor fails for c_dim > 0 (stylegan.py:377)')
        c4 = min(ch_base // self.encode_res[-1], ch_max)
        self.b4 = discrim_epilogue(c4, resolution=4, cmap_dim=None, activation=activation,
                                   mbstd_group_size=mbstd_group_size, mbstd_c_n=mbstd_c_n)

    def forward(self, img, c, segments=1, **kwargs):
        """``segments`` (not in the reference): ``img`` is that many independently judged sub-batches stacked along the batch axis.
        Every layer but the minibatch statistic is per-sample, and that one is then taken per segment, so
        ``D(cat([a, b]), c, segments=2) == cat([D(a), D(b)])`` -- one pass over the weights instead of two (losses.py, Dmain)."""
        x = None
        for res in self.encode_res[0:-1]:
            x, img = getattr(self, 'b{}'.format(res))(x, img)
        cmap = self.mapping(None, c) if self.mapping is not None else None
        return self.b4(x, img, cmap, segments=segments)
