"""2-D convolution with optional FIR up/down-sampling on the HIP kernels -- same call signature and
padding arithmetic as lib/model_zoo/stylegan_utils/conv2d_resample.py:57-154."""
import torch

from ... import kernels
from . import conv2d_gradfix, grad_ops, upfirdn2d
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
        if groups != 1 and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
            raise NotImplementedError('grouped convolutions are forward only')
        wg = w if flip_weight else w.flip([2, 3])
        if transpose:
            # torch layout [Cin, Cout/groups, kh, kw] (conv2d_resample.py:40-47)
            wt = wg.transpose(0, 1) if groups == 1 else wg.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2).reshape(groups * icg, oc // groups, kh, kw)
            return conv2d_gradfix.conv_transpose2d(x, wt, stride=stride, padding=padding, groups=groups)
        return conv2d_gradfix.conv2d(x, wg, stride=stride, padding=padding, groups=groups)
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


def _fusable(x, w, f, groups):
    """float32 training rows whose resampling layer has the model's shape: 3x3 weights, the 4x4 low-pass, one group."""
    return (x.dtype == torch.float32 and x.is_cuda and groups == 1 and tuple(w.shape[2:]) == (3, 3) and f is not None and f.ndim == 2
            and tuple(f.shape) == (4, 4) and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
            and x.shape[0] * max(w.shape[0], w.shape[1]) <= 65535)


class _UpConvFirFn(torch.autograd.Function):
    """The up path of conv2d_resample.py:122-142 for float32 training rows as ONE autograd node: transposed convolution as four phase
    planes (polyphase-Winograd kernel) -> FIR straight from the planes (``upfir_planar``, the pair the inference route uses).  The
    generic composition interleaved the planes into the (2H+1)^2 image only for the FIR to read it back (``planes_to_image``: 5.7 ms
    of a 170 ms float32 step).  Backward = the composition of the public differentiable operators (FIR transpose, strided
    convolution, weight gradient with the tensors exchanged: conv2d_gradfix.py:118-146), so second derivatives keep working."""
    @staticmethod
    def forward(ctx, x, wt, f, flip_filter):
        ctx.save_for_backward(x, wt, f)
        ctx.flip = flip_filter
        pw = kernels.conv_weight_prep(wt.detach().transpose(0, 1).contiguous())
        mid = kernels.conv2d(x.detach().contiguous(), pw, mode=kernels.MODE_UP2T, planar=True)
        return kernels.upfir_planar(mid, f, fir_gain=4.0, flip=flip_filter)

    @staticmethod
    def backward(ctx, gy):
        x, wt, f = ctx.saved_tensors
        n, _, h, w = x.shape
        g_mid = upfirdn2d.upfirdn2d_backward(gy.contiguous(), f, (n, wt.shape[1], 2 * h + 1, 2 * w + 1), padding=[1, 1, 1, 1],
                                             flip_filter=ctx.flip, gain=4)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = conv2d_gradfix.conv2d(g_mid, wt, stride=2, padding=0)
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            gw = conv2d_gradfix._WgradFn.apply(x, g_mid, 3, 2, 0)
        return gx, gw, None, None


def conv2d_down_bias_act(x, w, f, bias, ak, flip_filter=False):
    """The down path (pad-2 low-pass, stride-2 3x3 convolution: conv2d_resample.py:116-120) with bias + activation in the convolution's store
    pass, under autograd; None when the geometry is not the model's (callers then compose ``conv2d_resample`` and ``bias_act``)."""
    if not (conv2d_gradfix.conv_bias_act_supported(x, w, ak) and tuple(w.shape[2:]) == (3, 3) and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
        return None
    cfg = (bool(ak.get('act', False)), float(ak.get('gain', 1.0)), float(ak.get('alpha', 0.2)), float(ak.get('act_gain', kernels.SQRT2)),
           ak.get('clamp', 256.0))
    if FUSED_TRAIN_RESAMPLE and _fusable(x, w, f, 1):
        return _FirDownConvFn.apply(x, w, f, flip_filter, bias, cfg)
    xf = upfirdn2d.upfirdn2d(x=x, f=f, padding=[2, 2, 2, 2], flip_filter=flip_filter)
    return conv2d_gradfix._ConvBiasActFn.apply(xf, w, bias, 2, 0, cfg)


class _FirDownConvFn(torch.autograd.Function):
    """The down path (conv2d_resample.py:116-120: pad-2 low-pass, then the stride-2 convolution) for float32 training rows as one
    node.  Forward: the two kernels as before.  First-order backward: the input gradient is conv_transpose2d followed by the FIR's
    transpose -- exactly the phase-plane pair of the up path with gain 1 -- instead of planes -> interleaved image -> same-size FIR."""
    @staticmethod
    def forward(ctx, x, w, f, flip_filter, bias=None, act_cfg=None):
        xf = upfirdn2d.upfirdn2d(x.detach(), f, padding=[2, 2, 2, 2], flip_filter=flip_filter)
        ctx.cfg = (tuple(x.shape), flip_filter)
        ctx.act_cfg = act_cfg
        ctx.join = grad_ops.InputGradJoin.adopt(x, ctx.needs_input_grad[0])      # (an encoder block's feature map: its other consumer is the synthesis network)
        ctx.bias_dtype = None if bias is None else bias.dtype
        if act_cfg is None:
            ctx.save_for_backward(xf, w, f, x)
            return conv2d_gradfix.conv2d(xf, w.detach(), stride=2, padding=0)
        # bias + activation in the store pass of the strided convolution (``conv2d_down_bias_act``); the output is saved for their backward
        act, gain, alpha, act_gain, clamp = act_cfg
        pw = kernels.conv_weight_prep(w.detach())
        y = kernels.conv2d(xf, pw, mode=kernels.MODE_DOWN2, pad=0, bias=None if bias is None else bias.detach(), act=act, gain=gain, alpha=alpha,
                           act_gain=act_gain, clamp=clamp)
        ctx.save_for_backward(xf, w, f, x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.act_cfg is not None:
            xf, w, f, x, y = ctx.saved_tensors
            g, gb = grad_ops.bias_act_grads(g.contiguous(), y, ctx.act_cfg, ctx.bias_dtype is not None and ctx.needs_input_grad[4], ctx.bias_dtype)
        else:
            xf, w, f, x = ctx.saved_tensors
        x_shape, flip = ctx.cfg
        g = g.contiguous()
        gx = gw = None
        other = ctx.join.take() if ctx.join is not None else None
        if ctx.needs_input_grad[0]:
            if torch.is_grad_enabled():          # create_graph: compose the differentiable operators
                gm = conv2d_gradfix._conv_input_grad(g, w, xf.shape, 2, 0)
                gx = upfirdn2d.upfirdn2d_backward(gm, f, x_shape, padding=[2, 2, 2, 2], flip_filter=flip, gain=1)
            else:
                pw = kernels.conv_weight_prep(w.detach().transpose(0, 1).contiguous())
                mid = kernels.conv2d(g, pw, mode=kernels.MODE_UP2T, planar=True)
                gx = kernels.upfir_planar(mid, f, fir_gain=1.0, flip=not flip, residual=None if other is None else other.contiguous())
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            if torch.is_grad_enabled() and ctx.needs_input_grad[0]:
                # create_graph with weight gradients: the saved ``xf`` is detached from x, and the reference's Conv2dGradWeight is differentiable
                # in the input (conv2d_gradfix.py:140-146) -- filter the saved input again, on the differentiable operator (x itself is an
                # input of this node: saving it holds no extra memory).  The shipped regularisers run under no_weight_gradients and skip this.
                xf = upfirdn2d.upfirdn2d(x=x, f=f, padding=[2, 2, 2, 2], flip_filter=flip)
            gw = conv2d_gradfix._WgradFn.apply(g, xf, 3, 2, 0)
        return gx, gw, None, None, gb, None


FUSED_TRAIN_RESAMPLE = True      # (A/B switch for the two nodes above)


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
        if (FUSED_TRAIN_RESAMPLE and down == 2 and pads == [2, 2, 2, 2] and _fusable(x, w, f, groups) and x.shape[2] % 2 == 0
                and x.shape[3] % 2 == 0):
            return _FirDownConvFn.apply(x, w if flip_weight else w.flip([2, 3]), f, flip_filter)
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=pads, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                   # transposed strided convolution, then low-pass
        px0, px1, py0, py1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        if (FUSED_TRAIN_RESAMPLE and up == 2 and down == 1 and pxt == 0 and pyt == 0 and [px0, px1, py0, py1] == [1, 1, 1, 1]
                and _fusable(x, w, f, groups)):
            wg = w.flip([2, 3]) if flip_weight else w           # (_conv2d_wrapper is called with flip_weight negated for the transposed form)
            return _UpConvFirFn.apply(x, wg.transpose(0, 1), f, flip_filter)
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
