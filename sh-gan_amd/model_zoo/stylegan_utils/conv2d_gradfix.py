"""``conv2d`` / ``conv_transpose2d`` with the call signature of
lib/model_zoo/stylegan_utils/conv2d_gradfix.py:35-43, forward AND first-order backward on the HIP kernels:

* forward: the fp32-MFMA convolution kernels (``kernels.conv2d``);
* input gradient: the opposite operator with the same weights, as in the reference (conv2d_gradfix.py:118-128): a stride-1
  convolution's is a stride-1 convolution with the flipped, channel-transposed weights, a stride-2 convolution's is the
  stride-2 transposed convolution (cropped / zero-extended by the ``output_padding`` rule of :96-105) and vice versa;
* weight gradient: ``shg_conv2d_wgrad_f32`` (replaces the cuDNN backward-weight call of :140-146), bias gradient: a sum.

Supported geometry is what the generator / discriminator need: 3x3 / 1x1 kernels, stride 1 or 2, dilation 1, symmetric
padding, groups = 1 under autograd (grouped forward only), and the stride-2 3x3 transposed form.  The backward passes are
themselves written with these operators (and ``_WgradFn`` has the derivatives of ``Conv2dGradWeight.backward``,
conv2d_gradfix.py:148-163), so gradients of gradients work: the R1 and path-length regularisers differentiate twice."""
import os

import torch

from ... import kernels, kernels_f16
from . import grad_ops

PLANAR_CONVT = True                  # transposed convolutions: phase planes + interleave pass (False: the direct interleaved kernel)
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


def _dense(t):
    """Dense in the layout the kernels of its dtype take: NCHW for float32, NHWC (channels_last) for float16."""
    return t.contiguous(memory_format=torch.channels_last) if t.dtype == torch.float16 and t.ndim == 4 else t.contiguous()


def _fit(y, h, w, lo):
    """Rows/cols [lo, lo+h) x [lo, lo+w) of y, zero-extended where y ends earlier (output_padding)."""
    y = y[:, :, lo:lo + h, lo:lo + w]
    dh, dw = h - y.shape[2], w - y.shape[3]
    if dh or dw:
        y = torch.nn.functional.pad(y, (0, dw, 0, dh))
    return _dense(y)


THIN_1X1 = True                      # (A/B switch: thin 1x1 convolutions on the pointwise kernels)
_ONES = {}


def _ones(n, i, device):
    t = _ONES.get((n, i, str(device)))
    if t is None:
        t = _ONES[(n, i, str(device))] = torch.ones(n, i, device=device)
    return t


def _conv_fwd(x, weight, bias, stride, padding, groups):
    if x.dtype == torch.float16:             # the reference's fp16 blocks (stylegan.py:486,660-667): NHWC fp16-MFMA kernels, fp32 accumulation
        if groups != 1:
            # grouped halves (the reference's fused modulated form reshapes the batch into groups, stylegan.py:187-190): one launch per group
            cg, og = x.shape[1] // groups, weight.shape[0] // groups
            ys = [kernels_f16.conv2d(x[:, j * cg:(j + 1) * cg].contiguous(memory_format=torch.channels_last), weight[j * og:(j + 1) * og].to(torch.float16),
                                     None if bias is None else bias[j * og:(j + 1) * og], stride, padding) for j in range(groups)]
            return torch.cat(ys, 1)
        return kernels_f16.conv2d(x, weight.to(torch.float16), bias, stride, padding)
    n, c, h, w = x.shape
    if THIN_1X1 and groups == 1 and stride == 1 and padding == 0 and tuple(weight.shape[2:]) == (1, 1) and n <= 65535:
        # thin 1x1 convolutions (RGB branch: 64 -> 3 and its input gradient 3 -> 64; fromrgb 4 -> 64 and back): HBM-bound pointwise
        # kernels instead of the MFMA kernel with 3 of its 64 output (or 32 input) lanes in use
        o, i = weight.shape[0], weight.shape[1]
        if o <= 4 and 16 * i + 4096 <= 65536:
            return kernels.torgb(x, weight.reshape(o, i), styles=_ones(n, i, x.device), bias=bias)
        if i <= 8 and (h * w) % 4 == 0:
            return kernels.conv1x1_thin_in(x, weight.reshape(o, i), bias, act=False)
    pw = kernels.conv_weight_prep(weight, groups=groups)
    y = kernels.conv2d(x.reshape(n * groups, c // groups, h, w), pw, mode=kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2,
                       pad=padding, bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y


def _convt_fwd(x, weight, bias, padding, groups, out_hw=None):
    """conv_transpose2d(x, weight [Cin, Cout/g, 3, 3], stride 2) cropped by ``padding`` on every side; with ``out_hw`` = (h, w) the
    rows / columns [padding, padding + h) x [padding, padding + w) instead, zero where the result ends earlier (``_fit``)."""
    if x.dtype == torch.float16:
        if groups != 1:
            cg, og = x.shape[1] // groups, weight.shape[1]          # weight [Cin, Cout/g, 3, 3]: group j owns input rows j*cg ...
            ys = [kernels_f16.conv_transpose2d(x[:, j * cg:(j + 1) * cg].contiguous(memory_format=torch.channels_last),
                                               weight[j * cg:(j + 1) * cg].to(torch.float16), None if bias is None else bias[j * og:(j + 1) * og],
                                               padding, out_hw) for j in range(groups)]
            return torch.cat(ys, 1)
        return kernels_f16.conv_transpose2d(x, weight.to(torch.float16), bias, padding, out_hw)
    n, c, h, w = x.shape
    ci_g, co_g = weight.shape[0] // groups, weight.shape[1]
    # torch layout [Cin, Cout/g, kh, kw] -> per group [Cout/g, Cin/g, kh, kw]
    wg = weight.reshape(groups, ci_g, co_g, 3, 3).transpose(1, 2).reshape(groups * co_g, ci_g, 3, 3).contiguous()
    pw = kernels.conv_weight_prep(wg, groups=groups)
    if groups == 1 and PLANAR_CONVT and n * co_g <= 65535:
        # the four sub-pixel phases as planes (polyphase-Winograd kernel where its geometry allows), then one pass that interleaves,
        # crops / zero-extends and adds the bias
        mid = kernels.conv2d(x, pw, mode=kernels.MODE_UP2T, planar=True)
        oh, ow = out_hw if out_hw is not None else (2 * h + 1 - 2 * padding, 2 * w + 1 - 2 * padding)
        return kernels.planes_to_image(mid, padding, oh, ow, bias=bias)
    if out_hw is not None:
        return _fit(_convt_fwd(x, weight, bias, 0, groups), out_hw[0], out_hw[1], padding)
    y = kernels.conv2d(x.reshape(n * groups, ci_g, h, w), pw, mode=kernels.MODE_UP2T, bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if padding:
        y = y[:, :, padding:y.shape[2] - padding, padding:y.shape[3] - padding].contiguous()
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y


def _conv_input_grad(g, weight, x_shape, stride, padding, residual=None):
    """dL/dx of y = conv2d(x, weight, stride, padding) from g = dL/dy, through the public (differentiable) operators: a stride-1
    convolution with the flipped, channel-transposed weights, or the stride-2 transposed convolution cropped / zero-extended to
    the input extent (the output_padding rule of conv2d_gradfix.py:96-105)."""
    k = weight.shape[2]
    if residual is not None:
        # ``residual`` (grad_ops.InputGradJoin: the other consumer's input gradient, first-order passes only) is added in the store pass of
        # the kernel where the geometry has one (3x3 stride 1), by a tensor op otherwise
        if stride == 1 and k == 3 and g.is_cuda and not _wants_grad(g, weight, residual) and tuple(residual.shape) == tuple(x_shape):
            if g.dtype == torch.float16:
                pw = kernels_f16.pack_weight(weight.detach().to(torch.float16), transposed=True, flip=True)
                return kernels_f16.conv2d(g, pw, None, 1, k - 1 - padding, residual=_dense(residual.to(torch.float16)))
            pw = kernels.conv_weight_prep(weight.detach().transpose(0, 1).flip(2, 3).contiguous())
            return kernels.conv2d(g, pw, mode=kernels.MODE_SAME, pad=k - 1 - padding, residual=residual.contiguous())
        return _conv_input_grad(g, weight, x_shape, stride, padding) + residual
    if stride == 1:
        if g.dtype == torch.float16 and g.is_cuda and not _wants_grad(g, weight):
            # first-order pass of a half layer: the rotated, channel-transposed weight is built by the pack kernel itself (one gather
            # instead of flip + permute-copy + pack)
            pw = kernels_f16.pack_weight(weight.detach().to(torch.float16), transposed=True, flip=True)
            return kernels_f16.conv2d(g, pw, None, 1, k - 1 - padding)
        return conv2d(g, weight.transpose(0, 1).flip(2, 3), stride=1, padding=k - 1 - padding)
    if k != 3:
        raise NotImplementedError('conv2d backward: 1x1 stride-2 convolutions (the forward decimates with upfirdn2d first)')
    if not _wants_grad(g, weight):
        return _convt_fwd(g, weight, None, padding, 1, out_hw=(x_shape[2], x_shape[3]))     # crop / extension fused into the kernel
    return _fit(conv_transpose2d(g, weight, stride=2, padding=0), x_shape[2], x_shape[3], padding)


class _WgradFn(torch.autograd.Function):
    """dw[o,i,ky,kx] = sum g[n,o,oy,ox] * x[n,i,oy*s-p+ky,ox*s-p+kx] -- bilinear in (g, x); its own derivatives are again a
    convolution and an input gradient (Conv2dGradWeight.backward, conv2d_gradfix.py:148-163), which makes the convolution
    twice differentiable (R1 / path-length regularisers)."""

    @staticmethod
    def forward(ctx, g, x, k, stride, padding):
        ctx.save_for_backward(g, x)
        ctx.geom = (k, stride, padding)
        if x.dtype == torch.float16:
            return kernels_f16.conv2d_wgrad(x.detach(), g.detach().to(torch.float16), k, stride, padding)
        return kernels.conv2d_wgrad(x.detach().contiguous(), g.detach().contiguous(), k, k, stride, padding)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        k, stride, padding = ctx.geom
        gg = gx = None
        if ctx.needs_input_grad[0]:
            gg = conv2d(x, ggw, stride=stride, padding=padding)
        if ctx.needs_input_grad[1]:
            gx = _conv_input_grad(g, ggw, x.shape, stride, padding)
        return gg, gx, None, None, None


class _Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, w, b, stride, padding), groups = 1."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, padding, bias is not None)
        ctx.join = grad_ops.InputGradJoin.adopt(x, ctx.needs_input_grad[0])      # (a residual block's first convolution: see grad_ops.InputGradJoin)
        return _conv_fwd(x.detach(), weight.detach(), None if bias is None else bias.detach(), stride, padding, 1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, has_bias = ctx.geom
        g = _dense(g)
        gx = gw = gb = None
        other = ctx.join.take() if ctx.join is not None else None
        if ctx.needs_input_grad[0]:
            gx = _conv_input_grad(g, weight, x.shape, stride, padding, residual=other)
        elif other is not None:
            raise RuntimeError('InputGradJoin: a gradient was stashed for a convolution whose input needs none')
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            gw = _WgradFn.apply(g, x, weight.shape[2], stride, padding)
        if has_bias and ctx.needs_input_grad[2]:
            gb = grad_ops.channel_sum(g).to(g.dtype)
        return gx, gw, gb, None, None


class _ConvBiasActFn(torch.autograd.Function):
    """y = lrelu_agc(conv2d(x, w, stride 1 | 2, padding) + b) (or ``(..) * gain`` without activation) as ONE node on ONE forward kernel: the
    convolution kernels apply bias and activation in their store pass, as on the inference route.  The composed form (``_Conv2dFn``, then
    ``grad_ops._BiasActFn``) wrote the convolution result and read it back for the activation -- a tensor nobody needs again: the backward
    pass takes the activation's slope from the OUTPUT (``bias_act_grads``) and the convolution's gradients from x and w.  Backward is built
    from the same differentiable pieces as the two nodes it replaces, so R1 differentiates it twice."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, cfg, residual=None):
        act, gain, alpha, act_gain, clamp = cfg
        b = None if bias is None else bias.detach()
        if residual is not None and act:
            raise NotImplementedError('_ConvBiasActFn: a residual is added in the store pass of LINEAR layers only (the backward pass reads the '
                                      "activation's slope from the saved output)")
        res = None if residual is None else residual.detach()
        if x.dtype == torch.float16:
            y = kernels_f16.conv2d(x.detach(), weight.detach().to(torch.float16), None if b is None else b.to(torch.float32), stride, padding, act=act,
                                   gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp, residual=res)
        elif (residual is None and THIN_1X1 and weight.shape[2] == 1 and stride == 1 and padding == 0 and weight.shape[1] <= 8 and (x.shape[2] * x.shape[3]) % 4 == 0
              and x.shape[0] <= 65535):
            # fromrgb (4 -> 64): the HBM-bound pointwise kernel, bias + activation in the same pass
            y = kernels.conv1x1_thin_in(x.detach().contiguous(), weight.detach().reshape(weight.shape[0], weight.shape[1]), b, act=act, gain=gain,
                                        alpha=alpha, act_gain=act_gain, clamp=clamp)
        else:
            pw = kernels.conv_weight_prep(weight.detach())
            y = kernels.conv2d(x.detach().contiguous(), pw, mode=kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2, pad=padding, bias=b, act=act, gain=gain, alpha=alpha,
                               act_gain=act_gain, clamp=clamp, residual=None if res is None else res.contiguous())
        ctx.save_for_backward(x, weight, y)
        ctx.cfg, ctx.padding, ctx.stride = cfg, padding, stride
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.join = grad_ops.InputGradJoin.adopt(x, ctx.needs_input_grad[0])      # (a residual block's first convolution: see grad_ops.InputGradJoin)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        need_b = ctx.bias_dtype is not None and ctx.needs_input_grad[2]
        act, gain = ctx.cfg[0], ctx.cfg[1]
        if not act and LINEAR_GAIN_ON_WEIGHTS:
            # a linear layer, y = (conv(x, w) + b) * gain [+ r]: dL/dz = gain * dL/dy needs no pass over the activation gradient -- the gain
            # goes onto the (small) weight for the input gradient and onto the (small) weight / bias gradients (the critic's skip branches:
            # 14 whole-tensor passes per training step)
            gz, wz = _dense(gy), weight * gain
            gb = (grad_ops.channel_sum(gz) * gain).to(ctx.bias_dtype) if need_b else None
        else:
            gz, gb = grad_ops.bias_act_grads(_dense(gy), y, ctx.cfg, need_b, ctx.bias_dtype)
            wz, gain = weight, None
        other = ctx.join.take() if ctx.join is not None else None
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv_input_grad(gz, wz, x.shape, ctx.stride, ctx.padding, residual=other)
        elif other is not None:
            raise RuntimeError('InputGradJoin: a gradient was stashed for a convolution whose input needs none')
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            gw = _WgradFn.apply(gz, x, weight.shape[2], ctx.stride, ctx.padding)
            if gain is not None:
                gw = gw * gain
        # (a residual added in the store pass of a linear layer: y = conv * gain + r, dL/dr = dL/dy)
        return gx, gw, gb, None, None, None, (gy if len(ctx.needs_input_grad) > 6 and ctx.needs_input_grad[6] else None)


LINEAR_GAIN_ON_WEIGHTS = os.environ.get('SHG_LINEAR_GAIN_ON_WEIGHTS', '1') == '1'      # (A/B switch: 0 = dz = gain * dy as an elementwise pass)
FUSED_CONV_ACT = os.environ.get('SHG_FUSED_CONV_ACT', '1') == '1'        # (A/B switch: 0 = convolution and bias / activation as two nodes)


def conv_bias_act_supported(x, weight, act_kwargs):
    return (FUSED_CONV_ACT and act_kwargs is not None and x.is_cuda and x.ndim == 4 and tuple(weight.shape[2:]) in ((3, 3), (1, 1))
            and x.dtype in (torch.float16, torch.float32) and (x.dtype == torch.float32 or weight.shape[0] % 8 == 0)
            and _wants_grad(x, weight))


def conv2d_bias_act(x, weight, bias, padding, stride=1, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0, residual=None):
    """lrelu_agc(conv2d(x, weight, stride, padding) + bias) for 3x3 (stride 1 | 2) and 1x1 (stride 1) layers under autograd (see _ConvBiasActFn).
    ``residual`` (linear layers only, ``act=False``): (conv + bias) * gain + residual in the same store pass -- the sum of a residual block's two
    branches (stylegan.py:676-680: ``x = y.add_(x)``) without an elementwise pass of its own; its gradient is the output gradient."""
    return _ConvBiasActFn.apply(x, weight, bias, int(stride), int(padding), (bool(act), float(gain), float(alpha), float(act_gain), clamp), residual)


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
        g = _dense(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = conv2d(g, weight, stride=2, padding=padding)                     # conv2d_gradfix.py:124-127, transpose flipped
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            # dw[ci,co,ky,kx] = sum x[n,ci,y,x] * g[n,co,2y-p+ky,2x-p+kx]: the conv weight gradient with the tensors exchanged
            gw = _WgradFn.apply(x, g, 3, 2, padding)
        if has_bias and ctx.needs_input_grad[2]:
            gb = grad_ops.channel_sum(g).to(g.dtype)
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
