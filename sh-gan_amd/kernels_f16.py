"""Host wrappers of the fp16 entry points of libshgan_hip.so (csrc/conv_f16.hip; include/shgan_hip.h "fp16 route") -- the kernels
behind the reference's ``use_fp16`` branches (stylegan.py:136-138,486,660-667; comodgan.py:40-47,305).

Every fp16 activation is a ``[N,C,H,W]`` torch tensor in ``torch.channels_last`` memory format (= dense NHWC in HBM): the module
API keeps the reference's logical shapes, the kernels get 8 consecutive channels per 16-byte MFMA operand.  A tensor that arrives in
another layout is converted (one pass); outputs are channels_last.  fp32 accumulation everywhere, one rounding to fp16.
No CPU path, no fallback: a non-HIP tensor raises."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib, kernels
from ._lib import check

CL = torch.channels_last


def _h(L, t, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ShgError(f'{name} must reside on a HIP (cuda) device: libshgan_hip has no CPU path')
    if t.dtype != torch.float16 or t.ndim != 4:
        raise _lib.ShgError(f'{name} must be a rank-4 float16 tensor (got {t.dtype}, rank {t.ndim})')
    L._own(t, name)
    if not t.is_contiguous(memory_format=CL):
        t = t.contiguous(memory_format=CL)
    L.keep.append(t)
    return t


def _new_cl(L, n, c, h, w, zero=False):
    y = torch.empty((n, c, h, w), device=L.dev, dtype=torch.float16, memory_format=CL)
    return y.zero_() if zero else y


def _pad_channels(x, mult):
    """Zero-pad the channel dimension of an NHWC tensor to a multiple of ``mult`` (thin inputs: fromrgb has 4 channels)."""
    c = x.shape[1]
    cp = (c + mult - 1) // mult * mult
    if cp == c:
        return x
    y = torch.empty((x.shape[0], cp, x.shape[2], x.shape[3]), device=x.device, dtype=x.dtype, memory_format=CL).zero_()
    y[:, :c] = x
    return y


PACK_DIRECT = True        # (A/B switch) pack straight from the torch-layout tensor; False: permute-copy + pack as before


def _taps_weight(L, w_oihw, i_mult, flip=False):
    """[O,I,kh,kw] halves -> the MFMA-operand-order tensor shg_conv2d_f16 takes ([ceil(O/32)][kh*kw][Ip/16][64][8], input channels
    zero-padded to a multiple of ``i_mult``).  A contiguous tensor, or the ``transpose(0, 1)`` view of one, is packed by ONE gather kernel
    (``flip``: taps reversed -- the rotated weight of an input gradient); anything else through a [kh*kw][O][Ip] staging copy."""
    o, i, kh, kw = w_oihw.shape
    lib = _lib.get_lib()
    if PACK_DIRECT and i_mult == 32 and kh == kw and kh in (1, 3):
        src, tr = None, 0
        if w_oihw.is_contiguous():
            src = w_oihw
        elif w_oihw.transpose(0, 1).is_contiguous():
            src, tr = w_oihw.transpose(0, 1), 1
        if src is not None:
            ip = (i + 31) // 32 * 32
            wp = torch.empty(lib.shg_conv2d_f16_packed_weight_elems(kh * kw, o, ip), device=w_oihw.device, dtype=torch.float16)
            with L:
                check(lib.shg_conv2d_f16_pack_weight_oihw(kernels._ptr(src), kernels._ptr(wp), kh * kw, o, i, tr, int(bool(flip)), L.stream()),
                      'conv2d_f16_pack_weight_oihw')
            return wp
    if flip:
        w_oihw = w_oihw.flip(2, 3)
    wt = w_oihw.permute(2, 3, 0, 1)
    ip = (i + i_mult - 1) // i_mult * i_mult
    if ip != i:
        wt = F.pad(wt, (0, ip - i))
    wt = wt.contiguous()
    wp = torch.empty(lib.shg_conv2d_f16_packed_weight_elems(kh * kw, o, ip), device=w_oihw.device, dtype=torch.float16)
    with L:
        check(lib.shg_conv2d_f16_pack_weight(kernels._ptr(wt), kernels._ptr(wp), kh * kw, o, ip, L.stream()), 'conv2d_f16_pack_weight')
    return wp


class PackedWeight:
    """A convolution weight already in MFMA operand order (``pack_weight``): what ``conv2d`` / ``conv_transpose2d`` build per call from a
    [O,I,k,k] half tensor.  The no-grad route of the modules caches one per parameter version."""
    __slots__ = ('wp', 'o', 'i', 'k')

    def __init__(self, wp, o, i, k):
        self.wp, self.o, self.i, self.k = wp, o, i, k


def pack_weight(weight, transposed=False, flip=False):
    """weight [O,I,k,k] halves (``transposed``: the torch conv_transpose2d layout [Cin,Cout,3,3]; ``flip``: taps rotated by 180 degrees --
    ``pack_weight(w, transposed=True, flip=True)`` is the weight of the input gradient of ``conv2d(x, w)``) -> PackedWeight."""
    if weight.dtype != torch.float16 or weight.ndim != 4 or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (1, 3):
        raise _lib.ShgError('pack_weight: weight must be float16 [O,I,k,k] with k = 1 or 3')
    L = kernels._Launch()
    L._own(weight, 'weight')
    w = weight.detach().transpose(0, 1) if transposed else weight.detach()
    return PackedWeight(_taps_weight(L, w, 32, flip=flip), w.shape[0], w.shape[1], w.shape[2])


def conv2d(x, weight, bias=None, stride=1, padding=0, in_scale=None, out_scale=None, noise=None, noise_strength=1.0, act=None, gain=1.0,
           alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0, residual=None):
    """F.conv2d(x, weight, bias, stride, padding) on halves: x [N,I,H,W], weight [O,I,k,k] (k = 1 | 3) -> [N,O,OH,OW].
    Inference-route extension (shg_conv2d_f16_fused): ``in_scale`` [N,I] multiplies x while it is staged; ``out_scale`` [N,O], ``noise``
    ([OH,OW] / [N,1,OH,OW]) * ``noise_strength``, ``bias``, the activation (``act`` True / False; None = no tail) and ``residual`` form
    y = A(conv * out_scale + noise + bias) + residual in the store pass."""
    L = kernels._Launch()
    x = _h(L, x, 'x')
    packed = weight if isinstance(weight, PackedWeight) else None
    if packed is None:
        if weight.dtype != torch.float16 or weight.ndim != 4 or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (1, 3):
            raise _lib.ShgError('conv2d_f16: weight must be float16 [O,I,k,k] with k = 1 or 3')
        L._own(weight, 'weight')
    wo, wi, k = (packed.o, packed.i, packed.k) if packed is not None else (weight.shape[0], weight.shape[1], weight.shape[2])
    if wi != x.shape[1]:
        raise _lib.ShgError(f'conv2d_f16: weight expects {wi} input channels, x has {x.shape[1]}')
    n, i, h, w = x.shape
    o = wo
    oh, ow = (h + 2 * padding - k) // stride + 1, (w + 2 * padding - k) // stride + 1
    if oh < 1 or ow < 1:
        raise _lib.ShgError('conv2d_f16: empty output')
    xp = _pad_channels(x, 32)
    wt = packed.wp if packed is not None else _taps_weight(L, weight.detach(), 32)
    L.view(wt, 'weight')
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    y = _new_cl(L, n, o, oh, ow)
    fused = in_scale is not None or out_scale is not None or noise is not None or act is not None or residual is not None
    with kernels._timed(L, 'conv_f16', 2.0 * n * o * i * k * k * oh * ow):
        if not fused:
            check(_lib.get_lib().shg_conv2d_f16(kernels._ptr(xp), kernels._ptr(wt), kernels._ptr(b), kernels._ptr(y), n, xp.shape[1], o, h, w, k,
                                                stride, padding, 0, 0, oh, ow, L.stream()), 'conv2d_f16')
            return y
        s_in = None
        if in_scale is not None:
            s_in = L.req(in_scale.detach().to(torch.float32).reshape(n, i), 'in_scale')
            if xp.shape[1] != i:
                s_in = F.pad(s_in, (0, xp.shape[1] - i)).contiguous()
        s_out = None if out_scale is None else L.req(out_scale.detach().to(torch.float32).reshape(n, o), 'out_scale')
        nz, mode = _noise_flat(L, noise, n, oh * ow)
        res = _h(L, residual, 'residual')
        if res is not None and tuple(res.shape) != (n, o, oh, ow):
            raise _lib.ShgError('conv2d_f16: residual shape mismatch')
        a, al, g, cl = kernels._act_args(bool(act), gain, alpha, act_gain, clamp)
        check(_lib.get_lib().shg_conv2d_f16_fused(kernels._ptr(xp), kernels._ptr(wt), kernels._ptr(y), n, xp.shape[1], o, h, w, k, stride, padding, 0, 0,
                                                  oh, ow, kernels._ptr(s_in), kernels._ptr(s_out), kernels._ptr(nz), mode, float(noise_strength),
                                                  kernels._ptr(b), a, al, g, cl, kernels._ptr(res), L.stream()), 'conv2d_f16_fused')
    return y


def conv_transpose2d(x, weight, bias=None, padding=0, out_hw=None, in_scale=None):
    """Rows / columns [padding, padding + oh) of F.conv_transpose2d(x, weight [Cin,Cout,3,3], stride 2) on halves, zero where the
    (2H+1) x (2W+1) result ends earlier (the ``output_padding`` rule of conv2d_gradfix.py:96-105 when used as an input gradient).
    ``in_scale`` [N,Cin] (inference route): x * in_scale while the patch is staged."""
    L = kernels._Launch()
    x = _h(L, x, 'x')
    packed = weight if isinstance(weight, PackedWeight) else None            # (pack_weight(w, transposed=True))
    if packed is None:
        if weight.dtype != torch.float16 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[0] != x.shape[1]:
            raise _lib.ShgError('conv_transpose2d_f16: weight must be float16 [Cin,Cout,3,3] matching x')
        L._own(weight, 'weight')
    elif packed.k != 3 or packed.i != x.shape[1]:
        raise _lib.ShgError('conv_transpose2d_f16: packed weight does not match x')
    n, i, h, w = x.shape
    o = packed.o if packed is not None else weight.shape[1]
    oh, ow = out_hw if out_hw is not None else (2 * h + 1 - 2 * padding, 2 * w + 1 - 2 * padding)
    xp = _pad_channels(x, 32)
    wt = packed.wp if packed is not None else _taps_weight(L, weight.detach().transpose(0, 1), 32)
    L.view(wt, 'weight')
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    lib = _lib.get_lib()
    y = _new_cl(L, n, o, oh, ow, zero=bool(lib.shg_conv2d_f16_needs_clear(h, w, padding, oh, ow)))
    with kernels._timed(L, 'conv_f16_up', 2.0 * n * o * i * 9 * h * w):
        if in_scale is None:
            check(lib.shg_conv2d_f16(kernels._ptr(xp), kernels._ptr(wt), kernels._ptr(b), kernels._ptr(y), n, xp.shape[1], o, h, w, 3, 2, 0, 1,
                                     padding, oh, ow, L.stream()), 'conv_transpose2d_f16')
        else:
            s_in = L.req(in_scale.detach().to(torch.float32).reshape(n, i), 'in_scale')
            if xp.shape[1] != i:
                s_in = F.pad(s_in, (0, xp.shape[1] - i)).contiguous()
            check(lib.shg_conv2d_f16_fused(kernels._ptr(xp), kernels._ptr(wt), kernels._ptr(y), n, xp.shape[1], o, h, w, 3, 2, 0, 1, padding, oh, ow,
                                           kernels._ptr(s_in), None, None, 0, 0.0, kernels._ptr(b), 0, 0.0, 1.0, -1.0, None, L.stream()),
                  'conv_transpose2d_f16_fused')
    return y


def conv2d_wgrad(x, g, k, stride, padding):
    """dL/dweight [O,I,k,k] (float16, as the weight it belongs to) of y = conv2d(x, weight, stride, padding) from g = dL/dy; the sum
    over pixels runs in fp32 on the device and is rounded once."""
    L = kernels._Launch()
    x, g = _h(L, x, 'x'), _h(L, g, 'g')
    n, i, h, w = x.shape
    o, oh, ow = g.shape[1], g.shape[2], g.shape[3]
    if g.shape[0] != n or oh != (h + 2 * padding - k) // stride + 1 or ow != (w + 2 * padding - k) // stride + 1:
        raise _lib.ShgError(f'conv2d_wgrad_f16: g {tuple(g.shape)} is not the output extent of x {tuple(x.shape)} (k={k}, stride={stride}, padding={padding})')
    xp, gp = _pad_channels(x, 8), _pad_channels(g, 8)
    ip, op = xp.shape[1], gp.shape[1]
    lib = _lib.get_lib()
    nbytes = lib.shg_conv2d_wgrad_f16_workspace_bytes(n, ip, op, oh, ow, k)
    ws = torch.empty(max(nbytes // 4, 1), device=L.dev, dtype=torch.float32)
    dw = torch.empty((k * k, op, ip), device=L.dev, dtype=torch.float32)
    with kernels._timed(L, 'conv_wgrad_f16', 2.0 * n * o * i * k * k * oh * ow):
        check(lib.shg_conv2d_wgrad_f16(kernels._ptr(xp), kernels._ptr(gp), kernels._ptr(dw), n, ip, op, h, w, oh, ow, k, stride, padding,
                                       kernels._ptr(ws), ctypes.c_size_t(nbytes), L.stream()), 'conv2d_wgrad_f16')
    return dw[:, :o, :i].reshape(k, k, o, i).permute(2, 3, 0, 1).to(torch.float16).contiguous()


def upfirdn2d(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip=False, gain=1.0):
    """``upfirdn2d_plugin.upfirdn2d`` for halves (upfirdn2d.cpp:59 dispatches the same op for at::Half): f float32 [fh,fw]."""
    L = kernels._Launch()
    x = _h(L, x, 'x')
    f = L.req(f, 'f')
    if f.ndim != 2:
        raise _lib.ShgError('f must be rank 2')
    n, c, h, w = x.shape
    fh, fw = f.shape
    oh, ow = kernels.upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
    if oh < 1 or ow < 1:
        raise _lib.ShgError('upfirdn2d: output must be at least 1x1')
    xp = _pad_channels(x, 8)
    y = _new_cl(L, n, xp.shape[1], oh, ow)
    with kernels._timed(L, 'upfirdn2d_f16', 2.0 * (x.numel() + n * c * oh * ow)):
        check(_lib.get_lib().shg_upfirdn2d_f16(kernels._ptr(xp), kernels._ptr(f), kernels._ptr(y), n, xp.shape[1], h, w, fh, fw, upx, upy,
                                               downx, downy, padx0, padx1, pady0, pady1, int(bool(flip)), float(gain), L.stream()), 'upfirdn2d_f16')
    return y if xp.shape[1] == c else y[:, :c].contiguous(memory_format=CL)


def relayout_supported(x):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.ndim == 4 and (x.shape[1] % 8 == 0 or x.shape[1] <= 16) and x.shape[0] <= 65535 and x.numel() > 0
            and ((x.dtype == torch.float32 and x.is_contiguous()) or (x.dtype == torch.float16 and x.is_contiguous(memory_format=CL))))


def relayout(x):
    """float32 NCHW -> float16 channels_last, or float16 channels_last -> float32 NCHW (the other layout of the other dtype), one pass."""
    L = kernels._Launch()
    n, c, h, w = x.shape
    to_half = x.dtype == torch.float32
    x = L.req(x, 'x') if to_half else _h(L, x, 'x')
    y = _new_cl(L, n, c, h, w) if to_half else torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    with kernels._timed(L, 'relayout', 6.0 * x.numel()):
        check(_lib.get_lib().shg_relayout_f32_f16(kernels._ptr(x), kernels._ptr(y), n, c, h * w, int(to_half), L.stream()), 'relayout')
    return y


def bias_act(x, bias=None, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    L = kernels._Launch()
    x = _h(L, x, 'x')
    n, c, h, w = x.shape
    if c % 8:
        raise _lib.ShgError('bias_act_f16: channel count must be a multiple of 8')
    b = None if bias is None else L.req(bias.detach().to(torch.float32), 'bias')
    a, al, g, cl = kernels._act_args(act, gain, alpha, act_gain, clamp)
    y = _new_cl(L, n, c, h, w)
    with kernels._timed(L, 'bias_act_f16', 2.0 * 2 * x.numel()):
        check(_lib.get_lib().shg_bias_act_f16(kernels._ptr(x), kernels._ptr(b), kernels._ptr(y), n * h * w, c, a, al, g, cl, L.stream()), 'bias_act_f16')
    return y


def bias_act_backward(g, y, act=True, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    L = kernels._Launch()
    g, y = _h(L, g, 'g'), _h(L, y, 'y')
    if g.shape != y.shape or g.numel() % 8:
        raise _lib.ShgError('bias_act_backward_f16: g and y must have the same shape, a multiple of 8 elements')
    a, al, gn, cl = kernels._act_args(act, gain, alpha, act_gain, clamp)
    dx = _new_cl(L, *g.shape)
    with L:
        check(_lib.get_lib().shg_bias_act_backward_f16(kernels._ptr(g), kernels._ptr(y), kernels._ptr(dx), g.numel(), a, al, gn, cl, L.stream()),
              'bias_act_backward_f16')
    return dx


def _noise_flat(L, noise, n, hw):
    """noise: None | [H,W] | [1,1,H,W] (shared) | [N,1,H,W] (per sample), float32 -> (flat tensor, mode)."""
    if noise is None:
        return None, 0
    t = L.req(noise.detach().to(torch.float32), 'noise')
    if t.numel() == hw:
        return t.reshape(-1), 1
    if t.numel() == n * hw:
        return t.reshape(-1), 2
    raise _lib.ShgError(f'modtail_f16: noise of {tuple(noise.shape)} does not match {n} x {hw} pixels')


def modtail(t, d=None, noise=None, bias=None, act=False, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0):
    """y = A(t * d[n,c] + noise + bias[c]) in one pass (csrc/conv_f16.hip modtail_f16_kernel)."""
    L = kernels._Launch()
    t = _h(L, t, 't')
    n, c, h, w = t.shape
    if c % 8:
        raise _lib.ShgError('modtail_f16: channel count must be a multiple of 8')
    d32 = None if d is None else L.req(d.detach().to(torch.float32).reshape(n, c), 'd')
    b32 = None if bias is None else L.req(bias.detach().to(torch.float32), 'bias')
    nz, mode = _noise_flat(L, noise, n, h * w)
    a, al, g, cl = kernels._act_args(act, gain, alpha, act_gain, clamp)
    y = _new_cl(L, n, c, h, w)
    with kernels._timed(L, 'modtail_f16', 2.0 * 2 * t.numel()):
        check(_lib.get_lib().shg_modtail_f16(kernels._ptr(t), kernels._ptr(d32), kernels._ptr(nz), mode, kernels._ptr(b32), kernels._ptr(y), n, h * w, c,
                                             a, al, g, cl, L.stream()), 'modtail_f16')
    return y


def modtail_backward(gy, y, t=None, d=None, want_sums=True, want_noise=False, act=False, gain=1.0, alpha=0.2, act_gain=kernels.SQRT2, clamp=256.0,
                     u=None, e=None):
    """-> (gt [N,C,H,W] halves, s1 [N,C] fp32 = sum_hw gz*t | None, s0 [N,C] fp32 = sum_hw gz | None, gnoise [N,1,H,W] fp32 | None) with
    gz = gy * A'(y): the whole first-order backward of ``modtail`` in one pass over gy / y / t.  With ``u`` (halves like gy) / ``e`` [N,C]:
    gt = A'(y) * (gy * d + u * e) (the tail's double backward)."""
    L = kernels._Launch()
    gy, y = _h(L, gy, 'gy'), _h(L, y, 'y')
    t, u = _h(L, t, 't'), _h(L, u, 'u')
    n, c, h, w = y.shape
    d32 = None if d is None else L.req(d.detach().to(torch.float32).reshape(n, c), 'd')
    e32 = None if e is None else L.req(e.detach().to(torch.float32).reshape(n, c), 'e')
    lib = _lib.get_lib()
    nblk = lib.shg_modtail_backward_f16_blocks(h * w, c)
    part = torch.empty((n, nblk, 2, c), device=L.dev, dtype=torch.float32) if want_sums else None
    gnoise = torch.empty((n, 1, h, w), device=L.dev, dtype=torch.float32) if want_noise else None
    a, al, g, cl = kernels._act_args(act, gain, alpha, act_gain, clamp)
    gt = _new_cl(L, n, c, h, w)
    with kernels._timed(L, 'modtail_bwd_f16', 2.0 * (3 + (t is not None) + (u is not None)) * y.numel()):
        check(lib.shg_modtail_backward_f16(kernels._ptr(gy), kernels._ptr(y), kernels._ptr(t), kernels._ptr(d32), kernels._ptr(u), kernels._ptr(e32),
                                           kernels._ptr(gt), kernels._ptr(part), kernels._ptr(gnoise), n, h * w, c, a, al, g, cl, L.stream()),
              'modtail_backward_f16')
    s1 = s0 = None
    if want_sums:
        sums = kernels.sum_partials(part)     # fixed order over the workgroup partials: deterministic
        s1, s0 = sums[:, 0], sums[:, 1]
    return gt, s1, s0, gnoise
