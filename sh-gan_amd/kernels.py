"""Torch-tensor front end of the C-ABI (include/shgan_hip.h).

PyTorch is used for device memory and the current HIP stream only; every function here validates
its tensors, allocates the output and enqueues exactly the HIP kernels of libshgan_hip.so on
``torch.cuda.current_stream()``.  CPU tensors are rejected (there is no CPU path in the product)."""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import check

SQRT2 = math.sqrt(2.0)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _addr(t):
    return t.data_ptr() if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ShgError(f'{name} must reside on a HIP (cuda) device: libshgan_hip has no CPU path')
    if t.dtype != dtype:
        raise _lib.ShgError(f'{name} must be {dtype} (got {t.dtype})')
    return t if t.is_contiguous() else t.contiguous()


def _req_rows(t, name):
    """Like _req, but accepts 2-D views with a row stride (the C entry point takes a row pitch): never copies."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.ndim != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise _lib.ShgError(f'{name} must be a float32 HIP matrix with contiguous rows')
    return t


def _act_args(act, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0):
    """(act_flag, alpha, total_gain, total_clamp) following common/utils.py:135-143:
    total gain = act_gain*gain; clamp = clamp*gain; without activation the layer does x*gain."""
    if act:
        return 1, float(alpha), float(act_gain * gain), float(clamp * gain) if clamp is not None else -1.0
    return 0, 0.0, float(gain), -1.0


def _noise_args(noise, n):
    """noise: None | [OH,OW] | [1,1,OH,OW] (shared) | [N,1,OH,OW] (per sample)  ->  (tensor, mode)."""
    if noise is None:
        return None, 0
    noise = _req(noise, 'noise')
    if noise.ndim == 2:
        return noise, 1
    if noise.ndim == 4 and noise.shape[1] == 1 and noise.shape[0] in (1, n):
        return noise, (2 if noise.shape[0] == n and n > 1 else 1)
    raise _lib.ShgError(f'noise must be [H,W] or [N,1,H,W] (got {tuple(noise.shape)})')


# ------------------------------------------------------------------------------------------------
# upfirdn2d
# ------------------------------------------------------------------------------------------------

def upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
    oh, ow = ctypes.c_int(), ctypes.c_int()
    check(_lib.get_lib().shg_upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
                                                ctypes.byref(oh), ctypes.byref(ow)), 'upfirdn2d_out_size')
    return oh.value, ow.value


def upfirdn2d(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip=False, gain=1.0,
              epilogue=None):
    """Mirror of ``upfirdn2d_plugin.upfirdn2d`` (upfirdn2d.cpp:16).  ``epilogue`` (dict) fuses
    scale/bias/noise/act/residual -- see shg_upfirdn2d_epilogue_f32."""
    x = _req(x, 'x')
    f = _req(f, 'f')
    if x.ndim != 4:
        raise _lib.ShgError('x must be rank 4')
    if f.ndim != 2:
        raise _lib.ShgError('f must be rank 2')
    n, c, h, w = x.shape
    fh, fw = f.shape
    oh, ow = upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
    if oh < 1 or ow < 1:
        raise _lib.ShgError('upfirdn2d: output must be at least 1x1')
    y = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32)
    lib = _lib.get_lib()
    if epilogue is None:
        check(lib.shg_upfirdn2d_f32(_ptr(x), _ptr(f), _ptr(y), n, c, h, w, fh, fw, upx, upy, downx, downy, padx0, padx1,
                                    pady0, pady1, int(bool(flip)), float(gain), _stream()), 'upfirdn2d')
        return y
    e = epilogue
    scale = _req(e.get('scale'), 'scale')
    bias = _req(e.get('bias'), 'bias')
    residual = _req(e.get('residual'), 'residual')
    noise, nmode = _noise_args(e.get('noise'), n)
    act, alpha, g, clamp = _act_args(e.get('act', False), e.get('gain', 1.0), e.get('alpha', 0.2), e.get('act_gain', SQRT2),
                                      e.get('clamp', 256.0))
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('upfirdn2d epilogue: residual shape mismatch')
    check(lib.shg_upfirdn2d_epilogue_f32(_ptr(x), _ptr(f), _ptr(y), n, c, h, w, fh, fw, upx, upy, downx, downy, padx0, padx1,
                                         pady0, pady1, int(bool(flip)), float(gain), _ptr(scale), _ptr(bias), _ptr(noise),
                                         nmode, float(e.get('noise_strength', 1.0)), act, alpha, g, clamp, _ptr(residual),
                                         _stream()), 'upfirdn2d_epilogue')
    return y


# ------------------------------------------------------------------------------------------------
# pointwise
# ------------------------------------------------------------------------------------------------

def bias_act(x, bias=None, scale=None, noise=None, noise_strength=1.0, residual=None, act=True, gain=1.0, alpha=0.2,
             act_gain=SQRT2, clamp=256.0, out=None):
    x = _req(x, 'x')
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // max(n * c, 1)
    y = out if out is not None else torch.empty_like(x)
    noise, nmode = _noise_args(noise, n)
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    check(_lib.get_lib().shg_bias_act_f32(_ptr(x), _ptr(y), _ptr(_req(scale, 'scale')), _ptr(_req(bias, 'bias')), _ptr(noise),
                                          nmode, float(noise_strength), _ptr(_req(residual, 'residual')), n, c, hw, a, al, g,
                                          cl, _stream()), 'bias_act')
    return y


def fma(a, b, c):
    a, b, c = torch.broadcast_tensors(_req(a, 'a'), _req(b, 'b'), _req(c, 'c'))
    a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
    y = torch.empty_like(a)
    check(_lib.get_lib().shg_fma_f32(_ptr(a), _ptr(b), _ptr(c), _ptr(y), a.numel(), _stream()), 'fma')
    return y


def scale_channels(x, s):
    x = _req(x, 'x')
    s = _req(s, 's')
    n, c = x.shape[:2]
    y = torch.empty_like(x)
    check(_lib.get_lib().shg_scale_channels_f32(_ptr(x), _ptr(s), _ptr(y), n * c, x.numel() // (n * c), _stream()),
          'scale_channels')
    return y


def composite_u8(x4, img):
    x4 = _req(x4, 'x')
    img = _req(img, 'img')
    n, _, h, w = x4.shape
    out = torch.empty((n, 3, h, w), device=x4.device, dtype=torch.uint8)
    check(_lib.get_lib().shg_composite_u8(_ptr(x4), _ptr(img), _ptr(out), n, h, w, _stream()), 'composite_u8')
    return out


def assemble_input(real, mask):
    """real [N,3,H,W] in [-1,1], mask [N,1,H,W] or [N,H,W] in {0,1} -> x [N,4,H,W] = cat([mask-0.5, real*mask])."""
    real = _req(real, 'real')
    mask = _req(mask, 'mask')
    n, c, h, w = real.shape
    if c != 3 or mask.numel() != n * h * w:
        raise _lib.ShgError('assemble_input: real must be [N,3,H,W] and mask [N,(1,)H,W]')
    x = torch.empty((n, 4, h, w), device=real.device, dtype=torch.float32)
    check(_lib.get_lib().shg_assemble_input_f32(_ptr(real), _ptr(mask), _ptr(x), n, h, w, _stream()), 'assemble_input')
    return x


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------

class PreppedWeight:
    """GEMM-layout weights produced by shg_conv_weight_prep_f32 (+ the demodulation table wsq).  ``wu`` (Winograd
    F(2x2,3x3) layout, shg_conv_weight_prep_wino_f32) is built on first use by a stride-1 3x3 convolution."""
    __slots__ = ('wt', 'wsq', 'o', 'i', 'op', 'kh', 'kw', 'groups', 'wu', '_w', '_wscale', '_flip')

    def __init__(self, wt, wsq, o, i, op, kh, kw, groups=1, w=None, wscale=None, flip=False):
        self.wt, self.wsq, self.o, self.i, self.op, self.kh, self.kw, self.groups = wt, wsq, o, i, op, kh, kw, groups
        self.wu, self._w, self._wscale, self._flip = None, w, wscale, flip

    def wino(self):
        if self.wu is None:
            if self._w is None or self.groups != 1 or self.kh != 3 or self.kw != 3:
                raise _lib.ShgError('PreppedWeight.wino: needs an ungrouped 3x3 weight')
            kc = int(_lib.get_lib().shg_conv_wino_chunk())
            nchunk = (self.i + kc - 1) // kc
            self.wu = torch.empty((self.op // 64) * nchunk * 16 * 64 * kc, device=self.wt.device, dtype=torch.float32)
            check(_lib.get_lib().shg_conv_weight_prep_wino_f32(_ptr(self._w), _ptr(self._wscale), _ptr(self.wu), self.o, self.i,
                                                               self.op, int(bool(self._flip)), _stream()), 'conv_weight_prep_wino')
            self._w = None          # the transformed copy is all that is needed from here on
        return self.wu


def conv_weight_prep(w, demod=False, gain=1.0, flip=False, groups=1):
    """w [O,I,kh,kw] (or [G*Og, I, kh, kw] with ``groups``) -> PreppedWeight (one layout for all conv modes)."""
    w = _req(w, 'w')
    o_all, i, kh, kw = w.shape
    o = o_all // groups
    op = (o + 63) // 64 * 64                     # 64-column weight blocks
    kk = kh * kw
    ip = (i + 31) // 32 * 32                     # channel rows are zero padded to whole K-chunks
    wt = torch.empty((groups, ip * kk * op), device=w.device, dtype=torch.float32)
    wsq = torch.empty((groups, i * op), device=w.device, dtype=torch.float32) if demod else None
    wscale = torch.empty((o,), device=w.device, dtype=torch.float32)
    lib = _lib.get_lib()
    for g in range(groups):
        wg = w[g * o:(g + 1) * o]
        check(lib.shg_conv_weight_prep_f32(_ptr(wg), _ptr(wt[g]), _ptr(wscale), _ptr(wsq[g]) if demod else None, o, i, kh, kw,
                                           op, int(bool(demod)), float(gain), int(bool(flip)), _stream()), 'conv_weight_prep')
    keep = groups == 1 and kh == 3 and kw == 3
    return PreppedWeight(wt, wsq, o, i, op, kh, kw, groups, w=w if keep else None, wscale=wscale if keep else None, flip=flip)


# Winograd F(2x2,3x3) path for stride-1 3x3 'same' convolutions on images of at least WINO_MIN pixels per side (8 x 32 pixel tiles from 32 columns up, 16 x 16 below)
# (SHG_WINO=0 keeps everything on the direct implicit-GEMM kernel).
WINO = os.environ.get('SHG_WINO', '1') != '0'
WINO_MIN = int(os.environ.get('SHG_WINO_MIN', '16'))


MODE_SAME, MODE_DOWN2, MODE_UP2T = 0, 1, 2


class KernelTimer:
    """Optional live instrumentation used by bench.py: brackets every launch of a kernel class with HIP
    events on the launch stream and accumulates the algorithmic work (flops or bytes) it was given."""

    def __init__(self):
        self.records = {}       # class -> list of (start_event, end_event, work)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, cls, start, work):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.records.setdefault(cls, []).append((start, ev, work))

    def summary(self):
        """-> {class: dict(calls, ms, work)} ; call after torch.cuda.synchronize()."""
        out = {}
        for cls, recs in self.records.items():
            out[cls] = dict(calls=len(recs), ms=sum(a.elapsed_time(b) for a, b, _ in recs), work=sum(w for _, _, w in recs))
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def conv2d(x, pw, mode=MODE_SAME, pad=0, in_scale=None, out_scale=None, bias=None, noise=None, noise_strength=1.0,
           act=False, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0, residual=None, planar=False):
    """x [NB, I, H, W] (for grouped weights NB = N*groups slots) -> y [NB, O, OH, OW].
    MODE_UP2T with ``planar`` returns the four sub-pixel phase planes [4, NB, O, H+1, W+1] for ``upfir_planar``."""
    x = _req(x, 'x')
    nb, i, h, w = x.shape
    if i != pw.i:
        raise _lib.ShgError(f'conv2d: x has {i} channels, weights expect {pw.i}')
    if mode == MODE_SAME:
        oh, ow = h + 2 * pad - pw.kh + 1, w + 2 * pad - pw.kw + 1
    elif mode == MODE_DOWN2:
        oh, ow = (h + 2 * pad - pw.kh) // 2 + 1, (w + 2 * pad - pw.kw) // 2 + 1
    else:
        oh, ow = 2 * h + 1, 2 * w + 1
    lib = _lib.get_lib()
    if mode == MODE_UP2T and planar:
        y = torch.empty((4, nb, pw.o, h + 1, w + 1), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((nb, pw.o, oh, ow), device=x.device, dtype=torch.float32)
    noise, nmode = _noise_args(noise, nb)
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    residual = _req(residual, 'residual')
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('conv2d: residual shape mismatch')
    ws, ws_bytes = None, 0
    if mode != MODE_UP2T or planar:      # split-K of the transposed conv is wired for the planar output only
        ws_bytes = int(lib.shg_conv2d_workspace_bytes(nb, i, pw.o, h, w, pw.kh, pw.kw, mode, pad, pw.groups))
        if ws_bytes:
            ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32)
    if (WINO and mode == MODE_SAME and pad == 1 and pw.kh == 3 and pw.kw == 3 and pw.groups == 1 and h >= WINO_MIN and w >= WINO_MIN and w % 4 == 0 and x.data_ptr() % 16 == 0 and i <= 1024
            and (pw.wu is not None or pw._w is not None)):
        wu = pw.wino()
        t0 = _timer.begin() if _timer is not None else None
        check(lib.shg_conv2d_wino_f32(
            _ptr(x), _ptr(wu), _ptr(y), nb, i, pw.o, pw.op, h, w, _ptr(_req(in_scale, 'in_scale')), _ptr(_req(out_scale, 'out_scale')),
            _ptr(_req(bias, 'bias')), _ptr(noise), nmode, float(noise_strength), a, al, g, cl, _ptr(residual), _stream()), 'conv2d_wino')
        if t0 is not None:
            _timer.end('conv_wino', t0, 2.0 * nb * pw.o * i * 9 * oh * ow)      # direct-form (algorithmic) flops
        return y
    t0 = _timer.begin() if _timer is not None else None
    check(lib.shg_conv2d_f32(
        _ptr(x), _ptr(pw.wt), _ptr(y), nb, i, pw.o, pw.op, h, w, pw.kh, pw.kw, mode, pad, pw.groups, pw.wt.shape[1],
        _ptr(_req(in_scale, 'in_scale')), _ptr(_req(out_scale, 'out_scale')), _ptr(_req(bias, 'bias')), _ptr(noise), nmode,
        float(noise_strength), a, al, g, cl, _ptr(residual), 1 if planar else 0, _ptr(ws), ws_bytes, _stream()), 'conv2d')
    if t0 is not None:
        # algorithmic MACs: every (input pixel, tap) pair of the reference convolution, x2 flops
        taps = pw.kh * pw.kw
        pix = (h * w) if mode == MODE_UP2T else (oh * ow)
        _timer.end('conv_mfma', t0, 2.0 * nb * pw.o * i * taps * pix)
    return y


def upfir_planar(mid, f, scale=None, bias=None, noise=None, noise_strength=1.0, residual=None, act=False, gain=1.0,
                 alpha=0.2, act_gain=SQRT2, clamp=256.0, fir_gain=4.0, flip=False):
    """mid [4,N,C,H+1,W+1] (phase planes of the transposed conv) -> y [N,C,2H,2W]: 4x4 FIR (pad 1) + fused layer tail."""
    mid = _req(mid, 'mid')
    f = _req(f, 'f')
    if mid.ndim != 5 or mid.shape[0] != 4 or tuple(f.shape) != (4, 4):
        raise _lib.ShgError('upfir_planar: mid must be [4,N,C,H+1,W+1] and f 4x4')
    _, n, c, hp, wp = mid.shape
    h, w = hp - 1, wp - 1
    y = torch.empty((n, c, 2 * h, 2 * w), device=mid.device, dtype=torch.float32)
    noise, nmode = _noise_args(noise, n)
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    residual = _req(residual, 'residual')
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('upfir_planar: residual shape mismatch')
    check(_lib.get_lib().shg_upfir_planar_f32(_ptr(mid), _ptr(f), _ptr(y), n, c, h, w, int(bool(flip)), float(fir_gain),
                                              _ptr(_req(scale, 'scale')), _ptr(_req(bias, 'bias')), _ptr(noise), nmode,
                                              float(noise_strength), a, al, g, cl, _ptr(residual), _stream()), 'upfir_planar')
    return y


def conv1x1_thin_in(x, w, bias=None, wgain=1.0, act=True, gain=1.0):
    x = _req(x, 'x')
    w = _req(w, 'w')
    n, i, h, wd = x.shape
    o = w.shape[0]
    y = torch.empty((n, o, h, wd), device=x.device, dtype=torch.float32)
    a, al, g, cl = _act_args(act, gain)
    check(_lib.get_lib().shg_conv1x1_thin_in_f32(_ptr(x), _ptr(w), _ptr(_req(bias, 'bias')), _ptr(y), n, i, o, h * wd,
                                                 float(wgain), a, al, g, cl, _stream()), 'conv1x1_thin_in')
    return y


def torgb(x, w, styles=None, bias=None, base_up=None, f=None):
    x = _req(x, 'x')
    w = _req(w, 'w')
    n, i, h, wd = x.shape
    o = w.shape[0]
    y = torch.empty((n, o, h, wd), device=x.device, dtype=torch.float32)
    check(_lib.get_lib().shg_torgb_f32(_ptr(x), _ptr(w), _ptr(_req(styles, 'styles')), _ptr(_req(bias, 'bias')),
                                       _ptr(_req(base_up, 'base_up')), _ptr(_req(f, 'f')), _ptr(y), n, i, o, h, wd, _stream()),
          'torgb')
    return y


# ------------------------------------------------------------------------------------------------
# dense / style
# ------------------------------------------------------------------------------------------------

def dense(x, w, b=None, wgain=1.0, bgain=1.0, act=False, gain=1.0, out=None):
    x = _req(x, 'x')
    w = _req(w, 'w')
    n, k = x.shape
    o = w.shape[0]
    if w.shape[1] != k:
        raise _lib.ShgError(f'dense: x has {k} features, weight expects {w.shape[1]}')
    y = out if out is not None else torch.empty((n, o), device=x.device, dtype=torch.float32)
    a, al, g, cl = _act_args(act, gain)
    check(_lib.get_lib().shg_dense_f32(_ptr(x), _ptr(w), _ptr(_req(b, 'b')), _ptr(y), n, k, o, x.stride(0), y.stride(0),
                                       float(wgain), float(bgain), a, al, g, cl, _stream()), 'dense')
    return y


def normalize_2nd_moment(x, eps=1e-8):
    x = _req(x, 'x')
    y = torch.empty_like(x)
    check(_lib.get_lib().shg_normalize_2nd_moment_f32(_ptr(x), _ptr(y), x.shape[0], x.shape[1], float(eps), _stream()),
          'normalize_2nd_moment')
    return y


def modconv_style_prep(styles, pw=None, demod=True, pre_gain=1.0):
    """-> (s [N,I], dcoef [N,O] or None)."""
    styles = _req(styles, 'styles')
    n, i = styles.shape
    s = torch.empty((n, i), device=styles.device, dtype=torch.float32)
    d = None
    o = op = 0
    wsq = None
    if demod:
        o, op, wsq = pw.o, pw.op, pw.wsq
        d = torch.empty((n, o), device=styles.device, dtype=torch.float32)
    check(_lib.get_lib().shg_modconv_style_prep_f32(_ptr(styles), styles.stride(0), _ptr(wsq), _ptr(s), _ptr(d), n, i, o, op,
                                                    int(bool(demod)), float(pre_gain), _stream()), 'modconv_style_prep')
    return s, d


MAX_GROUPS = 32


def dense_grouped(items):
    """items: list of dicts(x1, x2|None, w, b|None, y, wgain, bgain): y = [x1 | x2] @ (w*wgain)^T + b*bgain, all in one
    launch per 32 groups.  Rows of x1 / x2 / y may be strided (views); all share the batch size."""
    if not items:
        return
    n = items[0]['x1'].shape[0]
    for lo in range(0, len(items), MAX_GROUPS):
        chunk = items[lo:lo + MAX_GROUPS]
        arr = (_lib.DenseGroup * len(chunk))()
        for g, it in zip(arr, chunk):
            x1, x2, w, y = _req_rows(it['x1'], 'x1'), _req_rows(it.get('x2'), 'x2'), _req(it['w'], 'w'), _req_rows(it['y'], 'y')
            k1 = x1.shape[1]
            k2 = 0 if x2 is None else x2.shape[1]
            if x1.shape[0] != n or y.shape[0] != n or w.shape[1] != k1 + k2 or y.shape[1] != w.shape[0]:
                raise _lib.ShgError('dense_grouped: inconsistent shapes')
            g.x1, g.x2, g.w, g.b, g.y = _addr(x1), _addr(x2), _addr(w), _addr(it.get('b')), _addr(y)
            g.ld1, g.ld2, g.K1, g.K2, g.O, g.ldy = x1.stride(0), (x2.stride(0) if x2 is not None else 0), k1, k2, w.shape[0], y.stride(0)
            g.wgain, g.bgain = float(it.get('wgain', 1.0)), float(it.get('bgain', 1.0))
        check(_lib.get_lib().shg_dense_grouped_f32(arr, len(chunk), n, _stream()), 'dense_grouped')


def modconv_style_prep_grouped(items):
    """items: list of dicts(styles [N,I], pw|None, demod, pre_gain, s [N,I] out, d [N,O] out|None) -- one launch per 32."""
    if not items:
        return
    n = items[0]['styles'].shape[0]
    for lo in range(0, len(items), MAX_GROUPS):
        chunk = items[lo:lo + MAX_GROUPS]
        arr = (_lib.StyleGroup * len(chunk))()
        for g, it in zip(arr, chunk):
            st, s_out = _req_rows(it['styles'], 'styles'), _req(it['s'], 's')
            demod = bool(it.get('demod', True))
            pw = it.get('pw')
            g.styles, g.s_out = _addr(st), _addr(s_out)
            g.wsq = _addr(pw.wsq) if demod else None
            g.dcoef = _addr(_req(it['d'], 'd')) if demod else None
            g.ld, g.I = st.stride(0), st.shape[1]
            g.O, g.OP = (pw.o, pw.op) if demod else (0, 0)
            g.demod, g.pre_gain = int(demod), float(it.get('pre_gain', 1.0))
        check(_lib.get_lib().shg_modconv_style_prep_grouped_f32(arr, len(chunk), n, _stream()), 'modconv_style_prep_grouped')


# ------------------------------------------------------------------------------------------------
# SHU
# ------------------------------------------------------------------------------------------------

def shu_rfft2_shift(x):
    """x: [N,C,64,64] view whose channel planes are contiguous (a channel slice of an NCHW tensor is fine)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and tuple(x.shape[2:]) == (64, 64)):
        raise _lib.ShgError('shu_rfft2_shift: x must be a float32 HIP tensor [N,C,64,64]')
    if x.stride(3) != 1 or x.stride(2) != 64 or x.stride(1) != 4096:
        x = x.contiguous()
    n, c = x.shape[:2]
    t = torch.empty((n, 2 * c, 64, 33), device=x.device, dtype=torch.float32)
    check(_lib.get_lib().shg_shu_rfft2_shift_f32(_ptr(x), x.stride(0), _ptr(t), n, c, _stream()), 'shu_rfft2_shift')
    return t


def shu_split_irfft2(y, cw, gauss, outs, accumulate):
    """y: [N, 2C*B, 64, 33]; cw: [B,64,33] or None; gauss: list of 5 tables (r=4..64);
    outs: list of 5 tensors/views [N,C,r,r] with contiguous channel planes (or None to skip)."""
    y = _req(y, 'y')
    bands = cw.shape[0] if cw is not None else 1
    n = y.shape[0]
    c = y.shape[1] // (2 * bands)
    g_arr = (ctypes.c_void_p * 5)()
    o_arr = (ctypes.c_void_p * 5)()
    s_arr = (ctypes.c_long * 5)()
    for l in range(5):
        r = 4 << l
        g = _req(gauss[l], 'gauss')
        g_arr[l] = g.data_ptr()
        o = outs[l]
        if o is None:
            o_arr[l] = None
            continue
        if not (o.is_cuda and o.dtype == torch.float32 and tuple(o.shape) == (n, c, r, r)
                and o.stride(3) == 1 and o.stride(2) == r and o.stride(1) == r * r):
            raise _lib.ShgError(f'shu_split_irfft2: out[{l}] must be a float32 [N,{c},{r},{r}] view with contiguous planes')
        o_arr[l] = o.data_ptr()
        s_arr[l] = o.stride(0)
    check(_lib.get_lib().shg_shu_split_irfft2_f32(_ptr(y), _ptr(_req(cw, 'cw')), g_arr, o_arr, s_arr, n, c, bands,
                                                  int(bool(accumulate)), _stream()), 'shu_split_irfft2')
    return outs
