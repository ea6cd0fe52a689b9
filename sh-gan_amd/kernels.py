"""Torch-tensor front end of the C-ABI (include/shgan_hip.h).

PyTorch is used for device memory and the current HIP stream only; every function here validates
its tensors, allocates the output and enqueues exactly the HIP kernels of libshgan_hip.so on the
current stream of the device that holds the operands.  CPU tensors are rejected (there is no CPU
path in the product), and so are operands that live on different devices."""
import contextlib
import ctypes
import math

import torch

from . import _lib
from ._lib import check

SQRT2 = math.sqrt(2.0)
_NULLCTX = contextlib.nullcontext()


class _Launch:
    """Operand bookkeeping of ONE C-ABI call: every tensor handed to the kernel goes through ``req`` (HIP device, dtype,
    contiguity -- a non-contiguous operand is copied and the copy is kept alive until the launch has been enqueued), all
    operands must share one device (the reference gets this from ATen's device guard, upfirdn2d.cpp:31), and the call
    itself runs under ``with launch:`` = that device made current, with ``launch.stream()`` = its current stream."""
    __slots__ = ('dev', 'keep', '_ctx')

    def __init__(self):
        self.dev = None
        self.keep = []
        self._ctx = None

    def _own(self, t, name):
        if self.dev is None:
            self.dev = t.device
        elif t.device != self.dev:
            raise _lib.ShgError(f'{name} lives on {t.device}, the other operands of this call on {self.dev}')

    def req(self, t, name, dtype=torch.float32):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.ShgError(f'{name} must reside on a HIP (cuda) device: libshgan_hip has no CPU path')
        if t.dtype != dtype:
            raise _lib.ShgError(f'{name} must be {dtype} (got {t.dtype})')
        self._own(t, name)
        if not t.is_contiguous():
            t = t.contiguous()
        self.keep.append(t)
        return t

    def rows(self, t, name):
        """2-D views with a row stride (the C entry point takes a row pitch): never copies."""
        if t is None:
            return None
        if (not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.ndim != 2
                or (t.shape[1] > 1 and t.stride(1) != 1)):
            raise _lib.ShgError(f'{name} must be a float32 HIP matrix with contiguous rows')
        self._own(t, name)
        return t

    def view(self, t, name):
        """An operand whose strides the caller has already validated (channel-slice views of NCHW tensors)."""
        self._own(t, name)
        return t

    def new(self, shape, dtype=torch.float32):
        return torch.empty(shape, device=self.dev, dtype=dtype)

    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def __enter__(self):
        self._ctx = _NULLCTX if self.dev is None or self.dev.index == torch.cuda.current_device() else torch.cuda.device(self.dev)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _addr(t):
    return t.data_ptr() if t is not None else None


def _act_args(act, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0):
    """(act_flag, alpha, total_gain, total_clamp) following common/utils.py:135-143:
    total gain = act_gain*gain; clamp = clamp*gain; without activation the layer does x*gain."""
    if act:
        return 1, float(alpha), float(act_gain * gain), float(clamp * gain) if clamp is not None else -1.0
    return 0, 0.0, float(gain), -1.0


def _noise_args(L, noise, n):
    """noise: None | [OH,OW] | [1,1,OH,OW] (shared) | [N,1,OH,OW] (per sample)  ->  (tensor, mode)."""
    if noise is None:
        return None, 0
    noise = L.req(noise, 'noise')
    if noise.ndim == 2:
        return noise, 1
    if noise.ndim == 4 and noise.shape[1] == 1 and noise.shape[0] in (1, n):
        return noise, (2 if noise.shape[0] == n and n > 1 else 1)
    raise _lib.ShgError(f'noise must be [H,W] or [N,1,H,W] (got {tuple(noise.shape)})')


class KernelTimer:
    """Optional live instrumentation used by bench.py's second pass: brackets every launch of a kernel class with HIP
    events on the launch stream and accumulates the algorithmic work (flops or bytes) it was given."""

    def __init__(self):
        self.records = {}       # class -> list of (start_event, end_event, work)

    def begin(self, stream):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def end(self, cls, start, work, stream, executed=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        self.records.setdefault(cls, []).append((start, ev, work, work if executed is None else executed))

    def summary(self):
        """-> {class: dict(calls, ms, work, executed)} ; call after torch.cuda.synchronize().  ``work`` = algorithmic
        (direct-form) flops or bytes, ``executed`` = flops the matrix cores actually run (Winograd forms run fewer)."""
        out = {}
        for cls, recs in self.records.items():
            out[cls] = dict(calls=len(recs), ms=sum(r[0].elapsed_time(r[1]) for r in recs), work=sum(r[2] for r in recs),
                            executed=sum(r[3] for r in recs))
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


@contextlib.contextmanager
def _timed(L, cls, work, executed=None):
    """``with _timed(L, 'class', work):`` -- device guard of the launch + (when a KernelTimer is installed) HIP events
    around it on the launch stream."""
    with L:
        if _timer is None:
            yield
        else:
            st = torch.cuda.current_stream(L.dev)
            t0 = _timer.begin(st)
            yield
            _timer.end(cls, t0, work, st, executed)


# ------------------------------------------------------------------------------------------------
# upfirdn2d
# ------------------------------------------------------------------------------------------------

def upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
    oh, ow = ctypes.c_int(), ctypes.c_int()
    check(_lib.get_lib().shg_upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
                                                ctypes.byref(oh), ctypes.byref(ow)), 'upfirdn2d_out_size')
    return oh.value, ow.value


FIR_MARCH = True        # separable 4x4 filters take the row-marching FIR kernels (csrc/fir_march.h)


def sep_taps(f):
    """Host taps {fx[0..3], fy[0..3]} (ctypes float[8]) when the 4x4 filter ``f`` is an outer product fy (x) fx -- every filter
    ``upfirdn2d.setup_filter`` builds from a 1-D kernel (upfirdn2d.py:61-95) -- else None.  One device read per filter tensor
    and version, cached on the tensor."""
    if not FIR_MARCH or f is None or tuple(f.shape) != (4, 4):
        return None
    key = (f.data_ptr(), f._version)
    hit = getattr(f, '_shg_sep', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    import numpy as np
    fh = f.detach().to('cpu', torch.float64).numpy()
    tot = fh.sum()
    taps = None
    if np.isfinite(fh).all() and abs(tot) > 1e-30:
        fy, fx = fh.sum(1), fh.sum(0) / tot
        if np.abs(np.outer(fy, fx) - fh).max() <= 5e-7 * np.abs(fh).max():      # (fp32 rounding of an outer product: ~1e-7)
            taps = (ctypes.c_float * 8)(*[float(v) for v in fx], *[float(v) for v in fy])
    try:
        f._shg_sep = (key, taps)
    except Exception:
        pass
    return taps


_STRIDED_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def upfirdn2d_strided(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip=False, gain=1.0):
    """``upfirdn2d_plugin.upfirdn2d`` over the plugin's whole operand range (upfirdn2d.cpp:38-59): x of dtype float32 / float16 / float64
    with ANY strides, read in place; y in ``x``'s suggested memory format (channels_last for a channels_last x, else NCHW), as :37."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda or not isinstance(f, torch.Tensor) or not f.is_cuda:
        raise _lib.ShgError('upfirdn2d: x and f must reside on a HIP (cuda) device: libshgan_hip has no CPU path')
    if x.dtype not in _STRIDED_DTYPES:
        raise _lib.ShgError(f'upfirdn2d: x must be float32, float16 or float64 (got {x.dtype})')
    if f.dtype != torch.float32:
        raise _lib.ShgError('upfirdn2d: f must be float32')
    if x.ndim != 4:
        raise _lib.ShgError('x must be rank 4')
    if f.ndim != 2:
        raise _lib.ShgError('f must be rank 2')
    L = _Launch()
    L._own(x, 'x')
    L._own(f, 'f')
    n, c, h, w = x.shape
    fh, fw = f.shape
    oh, ow = upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
    if oh < 1 or ow < 1:
        raise _lib.ShgError('upfirdn2d: output must be at least 1x1')
    cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    y = torch.empty((n, c, oh, ow), device=x.device, dtype=x.dtype, memory_format=torch.channels_last if cl else torch.contiguous_format)
    sx, sy = (ctypes.c_long * 4)(*x.stride()), (ctypes.c_long * 4)(*y.stride())
    with _timed(L, 'upfirdn2d', float(x.element_size()) * (x.numel() + y.numel())):
        check(_lib.get_lib().shg_upfirdn2d_strided(_ptr(x), _ptr(f), _ptr(y), _STRIDED_DTYPES[x.dtype], n, c, h, w, sx, sy, fh, fw, f.stride(0),
                                                   f.stride(1), upx, upy, downx, downy, padx0, padx1, pady0, pady1, int(bool(flip)), float(gain),
                                                   L.stream()), 'upfirdn2d_strided')
    return y


def upfirdn2d(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip=False, gain=1.0,
              epilogue=None):
    """Mirror of ``upfirdn2d_plugin.upfirdn2d`` (upfirdn2d.cpp:16).  ``epilogue`` (dict) fuses
    scale/bias/noise/act/residual -- see shg_upfirdn2d_epilogue_f32."""
    L = _Launch()
    x = L.req(x, 'x')
    f = L.req(f, 'f')
    if x.ndim != 4:
        raise _lib.ShgError('x must be rank 4')
    if f.ndim != 2:
        raise _lib.ShgError('f must be rank 2')
    n, c, h, w = x.shape
    fh, fw = f.shape
    oh, ow = upfirdn2d_out_size(h, w, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1)
    if oh < 1 or ow < 1:
        raise _lib.ShgError('upfirdn2d: output must be at least 1x1')
    y = L.new((n, c, oh, ow))
    lib = _lib.get_lib()
    work = 4.0 * (x.numel() + y.numel())                       # algorithmic bytes: input once, output once
    if epilogue is None:
        taps = None
        if (upx, upy, downx, downy) == (1, 1, 1, 1) and (padx0, padx1, pady0, pady1) == (2, 2, 2, 2) and (fh, fw) == (4, 4) \
                and lib.shg_fir_pad2_sep_supported(h, w, 0):
            taps = sep_taps(f)
        rs_up = 0
        if taps is None and (fh, fw) == (4, 4) and x.data_ptr() % 16 == 0:
            if (upx, upy, downx, downy) == (1, 1, 2, 2) and (padx0, padx1, pady0, pady1) == (1, 1, 1, 1):
                rs_up = 1
            elif (upx, upy, downx, downy) == (2, 2, 1, 1) and (padx0, padx1, pady0, pady1) == (2, 1, 2, 1):
                rs_up = 2
            if rs_up and lib.shg_fir_resample2_sep_supported(h, w, rs_up):
                taps = sep_taps(f)
            if taps is None:
                rs_up = 0
        with _timed(L, 'upfirdn2d', work):
            if rs_up:
                check(lib.shg_fir_resample2_sep_f32(_ptr(x), taps, _ptr(y), n, c, h, w, rs_up, int(bool(flip)), float(gain), L.stream()),
                      'fir_resample2_sep')
                return y
            if taps is not None:
                check(lib.shg_fir_pad2_sep_f32(_ptr(x), taps, _ptr(y), n, c, h, w, 0, int(bool(flip)), float(gain), L.stream()), 'fir_pad2_sep')
                return y
            check(lib.shg_upfirdn2d_f32(_ptr(x), _ptr(f), _ptr(y), n, c, h, w, fh, fw, upx, upy, downx, downy, padx0, padx1,
                                        pady0, pady1, int(bool(flip)), float(gain), L.stream()), 'upfirdn2d')
        return y
    e = epilogue
    scale = L.req(e.get('scale'), 'scale')
    bias = L.req(e.get('bias'), 'bias')
    residual = L.req(e.get('residual'), 'residual')
    noise, nmode = _noise_args(L, e.get('noise'), n)
    act, alpha, g, clamp = _act_args(e.get('act', False), e.get('gain', 1.0), e.get('alpha', 0.2), e.get('act_gain', SQRT2),
                                      e.get('clamp', 256.0))
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('upfirdn2d epilogue: residual shape mismatch')
    with _timed(L, 'upfirdn2d', work + (4.0 * y.numel() if residual is not None else 0.0)):
        check(lib.shg_upfirdn2d_epilogue_f32(_ptr(x), _ptr(f), _ptr(y), n, c, h, w, fh, fw, upx, upy, downx, downy, padx0, padx1,
                                             pady0, pady1, int(bool(flip)), float(gain), _ptr(scale), _ptr(bias), _ptr(noise),
                                             nmode, float(e.get('noise_strength', 1.0)), act, alpha, g, clamp, _ptr(residual),
                                             L.stream()), 'upfirdn2d_epilogue')
    return y


# ------------------------------------------------------------------------------------------------
# pointwise
# ------------------------------------------------------------------------------------------------

def bias_act(x, bias=None, scale=None, noise=None, noise_strength=1.0, residual=None, act=True, gain=1.0, alpha=0.2,
             act_gain=SQRT2, clamp=256.0, out=None):
    L = _Launch()
    x = L.req(x, 'x')
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // max(n * c, 1)
    y = L.req(out, 'out') if out is not None else torch.empty_like(x)
    if y is not out and out is not None:
        raise _lib.ShgError('bias_act: out must be contiguous')
    noise, nmode = _noise_args(L, noise, n)
    scale, bias, residual = L.req(scale, 'scale'), L.req(bias, 'bias'), L.req(residual, 'residual')
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    with _timed(L, 'bias_act', 4.0 * x.numel() * (3 if residual is not None else 2)):
        check(_lib.get_lib().shg_bias_act_f32(_ptr(x), _ptr(y), _ptr(scale), _ptr(bias), _ptr(noise), nmode, float(noise_strength),
                                              _ptr(residual), n, c, hw, a, al, g, cl, L.stream()), 'bias_act')
    return y


def bias_act_backward(g, y, act=True, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0):
    """dL/dx of y = bias_act(x, bias, act, gain, alpha, act_gain, clamp) from g = dL/dy and the forward output y."""
    L = _Launch()
    g, y = L.req(g, 'g'), L.req(y, 'y')
    if g.shape != y.shape:
        raise _lib.ShgError('bias_act_backward: g and y differ in shape')
    a, al, gn, cl = _act_args(act, gain, alpha, act_gain, clamp)
    dx = torch.empty_like(g)
    with L:
        check(_lib.get_lib().shg_bias_act_backward_f32(_ptr(g), _ptr(y), _ptr(dx), g.numel(), a, al, gn, cl, L.stream()), 'bias_act_backward')
    return dx


def modtail_backward(gy, y, t=None, d=None, want_sums=True, want_noise=False, act=True, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0, u=None, e=None):
    """First-order backward of y = A(t * d[n,c] + noise + bias[c]) (float32 NCHW) in one pass: -> (gt, s1 [N,C] = sum_hw gz*t | None,
    s0 [N,C] = sum_hw gz | None, gnoise [N,1,H,W] | None) with gz = gy * A'(y); with ``u`` (like gy) / ``e`` [N,C]: gt = A'(y) * (gy * d + u * e)."""
    L = _Launch()
    gy, y, t, u = L.req(gy, 'gy'), L.req(y, 'y'), L.req(t, 't'), L.req(u, 'u')
    n, c, h, w = y.shape
    d = L.req(None if d is None else d.reshape(n, c), 'd')
    e = L.req(None if e is None else e.reshape(n, c), 'e')
    lib = _lib.get_lib()
    nblk = lib.shg_modtail_backward_f32_blocks(h * w)
    part = torch.empty((n, nblk, 2, c), device=L.dev, dtype=torch.float32) if want_sums else None
    zs = lib.shg_modtail_backward_f32_cslices(n, c, h * w)
    gnoise = torch.empty((zs, n, 1, h, w), device=L.dev, dtype=torch.float32) if want_noise else None
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    gt = torch.empty_like(y)
    with _timed(L, 'modtail_bwd', 4.0 * (3 + (t is not None) + (u is not None)) * y.numel()):
        check(lib.shg_modtail_backward_f32(_ptr(gy), _ptr(y), _ptr(t), _ptr(d), _ptr(u), _ptr(e), _ptr(gt), _ptr(part), _ptr(gnoise), n, c, h * w, a, al, g,
                                           cl, L.stream()), 'modtail_backward_f32')
    s1 = s0 = None
    if want_sums:
        sums = sum_partials(part)             # fixed order over the workgroup partials: deterministic
        s1, s0 = sums[:, 0], sums[:, 1]
    if gnoise is not None:
        gnoise = gnoise[0] if zs == 1 else gnoise.sum(0)
    return gt, s1, s0, gnoise


def scale_cast(x, gain, to_half):
    """half(x * gain) of a float32 tensor (``to_half``) or float(x) * gain of a float16 one, contiguous, one launch (shg_scale_cast_f32_f16)."""
    L = _Launch()
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == (torch.float32 if to_half else torch.float16)):
        raise _lib.ShgError('scale_cast: a HIP tensor of the source dtype is expected')
    x = x.contiguous()
    L._own(x, 'x')
    y = torch.empty(x.shape, dtype=torch.float16 if to_half else torch.float32, device=x.device)
    with L:
        check(_lib.get_lib().shg_scale_cast_f32_f16(_ptr(x), _ptr(y), x.numel(), float(gain), int(bool(to_half)), L.stream()), 'scale_cast')
    return y


def sum_partials(part):
    """part [N, B, ...] float32 -> [N, ...] = sum over B in block order (shg_sum_partials_f32)."""
    L = _Launch()
    part = L.req(part, 'part')
    n, b = part.shape[:2]
    out = L.new((n,) + tuple(part.shape[2:]))
    with L:
        check(_lib.get_lib().shg_sum_partials_f32(_ptr(part), _ptr(out), n, b, out.numel() // n, L.stream()), 'sum_partials')
    return out


def _collapse(shape, strides_list):
    """Drop size-1 dimensions and merge neighbours that every operand walks as one run (stride[i] == stride[i+1] * size[i+1], a
    broadcast operand has 0 on both): the kernels decompose an index over at most six dimensions."""
    dims = [(int(n), tuple(int(st[i]) if n > 1 else 0 for st in strides_list)) for i, n in enumerate(shape) if n != 1]
    out = []
    for n, st in dims:
        if out and all(ps == cs * n for ps, cs in zip(out[-1][1], st)):
            out[-1] = (out[-1][0] * n, st)
        else:
            out.append((n, st))
    return [n for n, _ in out], [[st[k] for _, st in out] for k in range(len(strides_list))]


def _larr(v):
    return (ctypes.c_long * max(len(v), 1))(*v)


def _fma_operands(L, ts, names):
    dt = ts[0].dtype
    if dt not in (torch.float32, torch.float64):
        raise _lib.ShgError(f'fma: float32 or float64 operands (got {dt})')
    for t, name in zip(ts, names):
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.ShgError(f'{name} must reside on a HIP (cuda) device: libshgan_hip has no CPU path')
        if t.dtype != dt:
            raise _lib.ShgError(f'{name} must be {dt} like the first operand (got {t.dtype})')
        L._own(t, name)
    return dt


def fma(a, b, c=None):
    """a * b + c with NumPy broadcasting (stylegan_utils/fma.py:15): operands are read through their strides, nothing is expanded.
    c = None: the product alone (fma.py:41,44).  float32 or float64."""
    L = _Launch()
    dt = _fma_operands(L, (a, b, c), 'abc')
    ts = [a, b] + ([c] if c is not None else [])
    shape = torch.broadcast_shapes(*[t.shape for t in ts])
    ex = [t.expand(shape) for t in ts]
    y = torch.empty(shape, dtype=dt, device=L.dev)
    cshape, cst = _collapse(shape, [t.stride() for t in ex])
    if len(cshape) > 6:
        ex = [t.contiguous() for t in ex]
        cshape, cst = [y.numel()], [[1]] * len(ex)
    with L:
        check(_lib.get_lib().shg_fma_bcast(_ptr(a), _ptr(b), _ptr(c), _ptr(y), len(cshape), _larr(cshape), _larr(cst[0]), _larr(cst[1]),
                                           _larr(cst[2]) if c is not None else None, int(dt == torch.float64), L.stream()), 'fma')
    del ex
    return y


def mul_reduce(g, b, shape):
    """``_unbroadcast(g * b, shape)`` (fma.py:48-58; b = None: ``_unbroadcast(g, shape)``) in one pass: the product is summed over
    the dimensions along which an operand of ``shape`` was broadcast to ``g.shape``; returns a tensor of ``shape``."""
    L = _Launch()
    dt = _fma_operands(L, (g, b), 'gb')
    shape = tuple(shape)
    extra = g.ndim - len(shape)
    if extra < 0:
        raise _lib.ShgError('mul_reduce: the gradient has fewer dimensions than the operand')
    full = (1,) * extra + shape
    bx = b.expand(g.shape) if b is not None else None
    red = [i for i in range(g.ndim) if g.shape[i] > 1 and full[i] == 1]
    for i in range(g.ndim):
        if full[i] not in (1, g.shape[i]):
            raise _lib.ShgError(f'mul_reduce: {shape} does not broadcast to {tuple(g.shape)}')
    out = torch.empty(shape, dtype=dt, device=L.dev)
    kept = [i for i in range(g.ndim) if i not in red]
    ops = [g] + ([bx] if bx is not None else [])
    kshape, kst = _collapse([g.shape[i] for i in kept], [[t.stride(i) for i in kept] for t in ops])
    rshape, rst = _collapse([g.shape[i] for i in red], [[t.stride(i) for i in red] for t in ops])
    if len(kshape) + len(rshape) > 6:
        g = g.contiguous()
        bx = bx.contiguous() if bx is not None else None
        ops = [g] + ([bx] if bx is not None else [])
        kshape, kst = _collapse([g.shape[i] for i in kept], [[t.stride(i) for i in kept] for t in ops])
        rshape, rst = _collapse([g.shape[i] for i in red], [[t.stride(i) for i in red] for t in ops])
        if len(kshape) + len(rshape) > 6:
            raise _lib.ShgError('mul_reduce: more than six non-mergeable dimensions')
    sg = kst[0] + rst[0]
    sb = (kst[1] + rst[1]) if bx is not None else None
    # the fastest-varying dimension of g (smallest non-zero stride): kept -> thread per output, reduced -> workgroup per output
    def fastest(st):
        v = [abs(x) for x in st if x]
        return min(v) if v else float('inf')
    inner_kept = int(not rshape or (bool(kshape) and fastest(kst[0]) < fastest(rst[0])))
    with L:
        check(_lib.get_lib().shg_mul_reduce(_ptr(g), _ptr(bx), _ptr(out), len(kshape) + len(rshape), len(kshape), _larr(kshape + rshape),
                                            _larr(sg), _larr(sb) if sb is not None else None, inner_kept, int(dt == torch.float64),
                                            L.stream()), 'mul_reduce')
    return out


def scale_channels(x, s):
    L = _Launch()
    x = L.req(x, 'x')
    s = L.req(s, 's')
    n, c = x.shape[:2]
    y = torch.empty_like(x)
    with L:
        check(_lib.get_lib().shg_scale_channels_f32(_ptr(x), _ptr(s), _ptr(y), n * c, x.numel() // (n * c), L.stream()),
              'scale_channels')
    return y


def planes_to_image(mid, lo, out_h, out_w, bias=None):
    """mid [4,N,C,H+1,W+1] (phase planes of a stride-2 transposed convolution) -> [N,C,out_h,out_w] = rows / columns
    [lo, lo+out) of the interleaved (2H+1) x (2W+1) result, zero where it ends earlier, + bias[c]."""
    L = _Launch()
    mid = L.req(mid, 'mid')
    bias = L.req(bias, 'bias')
    _, n, c, hp, wp = mid.shape
    y = L.new((n, c, out_h, out_w))
    if n * c > 65535:
        raise _lib.ShgError('planes_to_image: N*C > 65535')
    with _timed(L, 'planes_to_image', 4.0 * 2 * y.numel()):
        check(_lib.get_lib().shg_planes_to_image_f32(_ptr(mid), _ptr(bias), _ptr(y), n, c, hp - 1, wp - 1, int(lo), int(out_h), int(out_w),
                                                     L.stream()), 'planes_to_image')
    return y


def composite_u8(x4, img, out=None):
    """shgan_default.py:257-262: uint8 [N,3,H,W]; ``out`` = a caller-owned destination (a slice of a result buffer: the evaluation loop
    streams every batch into its place instead of concatenating afterwards)."""
    L = _Launch()
    x4 = L.req(x4, 'x')
    img = L.req(img, 'img')
    n, _, h, w = x4.shape
    if out is None:
        out = L.new((n, 3, h, w), torch.uint8)
    else:
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.uint8 and tuple(out.shape) == (n, 3, h, w) and out.is_contiguous()):
            raise _lib.ShgError('composite_u8: out must be a contiguous uint8 HIP tensor [N,3,H,W]')
        L._own(out, 'out')
    with _timed(L, 'composite_u8', 4.0 * (x4.numel() + img.numel()) + out.numel()):
        check(_lib.get_lib().shg_composite_u8(_ptr(x4), _ptr(img), _ptr(out), n, h, w, L.stream()), 'composite_u8')
    return out


def minibatch_std(x, group_size, num_channels=1):
    """stylegan.py:686-704: x [N,C,H,W] -> [N,C+F,H,W] with the per-group standard-deviation statistic appended."""
    L = _Launch()
    x = L.req(x, 'x')
    n, c, h, w = x.shape
    g = n if group_size is None else min(int(group_size), n)
    f = int(num_channels)
    y = L.new((n, c + f, h, w))
    stat = L.new((max(n // max(g, 1), 1) * f,))
    with L:
        check(_lib.get_lib().shg_minibatch_std_f32(_ptr(x), _ptr(y), _ptr(stat), n, c, h, w, g, f, L.stream()), 'minibatch_std')
    return y


_U8_LUT = {}


def u8_value_table(device):
    """float value of every uint8 code as the dataset route computes it on the host: ToTensor (/255) then the formatter's ``* 2 - 1``
    (ds_ffhq.py:318-326,338), evaluated with the same torch CPU float32 operations -- the device hand-off is bit-identical to it."""
    key = str(device)
    if key not in _U8_LUT:
        _U8_LUT[key] = (torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255) * 2 - 1).to(device)
    return _U8_LUT[key]


def assemble_input(real, mask, lut=None):
    """real [N,3,H,W] in [-1,1] (float32) or decoded uint8 pixels, mask [N,1,H,W] or [N,H,W] in {0,1} -> x [N,4,H,W] =
    cat([mask-0.5, real*mask]); uint8 pixels take their float value from ``lut`` [256] (default ``u8_value_table``)."""
    L = _Launch()
    u8 = isinstance(real, torch.Tensor) and real.dtype == torch.uint8
    real = L.req(real, 'real', dtype=torch.uint8 if u8 else torch.float32)
    mask = L.req(mask, 'mask')
    n, c, h, w = real.shape
    if c != 3 or mask.numel() != n * h * w:
        raise _lib.ShgError('assemble_input: real must be [N,3,H,W] and mask [N,(1,)H,W]')
    x = L.new((n, 4, h, w))
    with L:
        if u8:
            lut = L.req(u8_value_table(L.dev) if lut is None else lut, 'lut')
            if lut.numel() != 256:
                raise _lib.ShgError('assemble_input: lut must hold 256 floats')
            check(_lib.get_lib().shg_assemble_input_u8(_ptr(real), _ptr(mask), _ptr(lut), _ptr(x), n, h, w, L.stream()), 'assemble_input_u8')
        else:
            check(_lib.get_lib().shg_assemble_input_f32(_ptr(real), _ptr(mask), _ptr(x), n, h, w, L.stream()), 'assemble_input')
    return x


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------

class PreppedWeight:
    """GEMM-layout weights produced by shg_conv_weight_prep_f32 (+ the demodulation table wsq).  ``wu`` (Winograd
    F(2x2,3x3) layout, shg_conv_weight_prep_wino_f32) is built on first use by a stride-1 3x3 convolution."""
    __slots__ = ('wt', 'wsq', 'o', 'i', 'op', 'kh', 'kw', 'groups', 'wu', 'wu4', 'wu_up', 'wu_down', '_w', '_wscale', '_flip')

    def __init__(self, wt, wsq, o, i, op, kh, kw, groups=1, w=None, wscale=None, flip=False):
        self.wt, self.wsq, self.o, self.i, self.op, self.kh, self.kw, self.groups = wt, wsq, o, i, op, kh, kw, groups
        self.wu, self.wu4, self.wu_up, self.wu_down, self._w, self._wscale, self._flip = None, None, None, None, w, wscale, flip

    def _poly(self, entry):
        if self._w is None or self.groups != 1 or self.kh != 3 or self.kw != 3:
            raise _lib.ShgError('PreppedWeight: polyphase forms need an ungrouped 3x3 weight')
        L = _Launch()
        w, ws = L.req(self._w, 'w'), L.req(self._wscale, 'wscale')
        size = (self.op // 64) * ((self.i + 7) // 8) * 16 * 64 * 8
        wa, wb = L.new((size,)), L.new((size,))
        with L:
            check(getattr(_lib.get_lib(), entry)(_ptr(w), _ptr(ws), _ptr(wa), _ptr(wb), self.o, self.i, self.op,
                                                 int(bool(self._flip)), L.stream()), entry)
        return wa, wb

    def down_poly(self):
        """(wu_a, wu_b): polyphase-Winograd weights of the stride-2 convolution (shg_conv_weight_prep_down_poly_f32)."""
        if self.wu_down is None:
            self.wu_down = self._poly('shg_conv_weight_prep_down_poly_f32')
        return self.wu_down

    def up_poly(self):
        """(wu_a, wu_b): polyphase-Winograd weights of the stride-2 transposed convolution (shg_conv_weight_prep_up_poly_f32)."""
        if self.wu_up is None:
            self.wu_up = self._poly('shg_conv_weight_prep_up_poly_f32')
        return self.wu_up

    def wino(self):
        if self.wu is None:
            if self._w is None or self.groups != 1 or self.kh != 3 or self.kw != 3:
                raise _lib.ShgError('PreppedWeight.wino: needs an ungrouped 3x3 weight')
            L = _Launch()
            w, ws = L.req(self._w, 'w'), L.req(self._wscale, 'wscale')
            kc = int(_lib.get_lib().shg_conv_wino_chunk())
            nchunk = (self.i + kc - 1) // kc
            self.wu = L.new(((self.op // 64) * nchunk * 16 * 64 * kc,))
            with L:
                check(_lib.get_lib().shg_conv_weight_prep_wino_f32(_ptr(w), _ptr(ws), _ptr(self.wu), self.o, self.i, self.op,
                                                                   int(bool(self._flip)), L.stream()), 'conv_weight_prep_wino')
        return self.wu

    def wino4(self):
        """Winograd F(4x4,3x3) weights (shg_conv_weight_prep_wino4_f32), built on first use."""
        if self.wu4 is None:
            if self._w is None or self.groups != 1 or self.kh != 3 or self.kw != 3:
                raise _lib.ShgError('PreppedWeight.wino4: needs an ungrouped 3x3 weight')
            L = _Launch()
            w, ws = L.req(self._w, 'w'), L.req(self._wscale, 'wscale')
            self.wu4 = L.new((int(_lib.get_lib().shg_conv_wino4_weight_elems(self.op, self.i)),))
            with L:
                check(_lib.get_lib().shg_conv_weight_prep_wino4_f32(_ptr(w), _ptr(ws), _ptr(self.wu4), self.o, self.i, self.op,
                                                                    int(bool(self._flip)), L.stream()), 'conv_weight_prep_wino4')
        return self.wu4


def conv_weight_prep(w, demod=False, gain=1.0, flip=False, groups=1):
    """w [O,I,kh,kw] (or [G*Og, I, kh, kw] with ``groups``) -> PreppedWeight (one layout for all conv modes)."""
    L = _Launch()
    w = L.req(w, 'w')
    o_all, i, kh, kw = w.shape
    o = o_all // groups
    op = (o + 63) // 64 * 64                     # 64-column weight blocks
    kk = kh * kw
    ip = (i + 31) // 32 * 32                     # channel rows are zero padded to whole K-chunks
    wt = L.new((groups, ip * kk * op))
    wsq = L.new((groups, i * op)) if demod else None
    wscale = L.new((o,))
    lib = _lib.get_lib()
    with L:
        for g in range(groups):
            wg = w[g * o:(g + 1) * o]
            check(lib.shg_conv_weight_prep_f32(_ptr(wg), _ptr(wt[g]), _ptr(wscale), _ptr(wsq[g]) if demod else None, o, i, kh, kw,
                                               op, int(bool(demod)), float(gain), int(bool(flip)), L.stream()), 'conv_weight_prep')
    keep = groups == 1 and kh == 3 and kw == 3
    return PreppedWeight(wt, wsq, o, i, op, kh, kw, groups, w=w if keep else None, wscale=wscale if keep else None, flip=flip)


# Winograd F(2x2,3x3) path for stride-1 3x3 'same' convolutions on images of at least WINO_MIN pixels per side (8 x 32 pixel
# tiles from 32 columns up, 16 x 16 below).  Module attributes, not environment switches: tools/ and the route-agreement
# test set ``kernels.WINO = False`` to keep everything on the direct implicit-GEMM kernel.
WINO = True
WINO_MIN = 16
WINO4 = True            # F(4x4,3x3) (conv_wino4.hip) where it is served and the image has at least WINO4_MIN rows; else F(2x2,3x3)
WINO4_MIN = 32
WINO_SPLIT = True       # (A/B switch: Winograd launches whose grid does not fill the chip split along the input channels)
UP_POLY = True          # stride-2 transposed 3x3 convolutions in the polyphase-Winograd form (conv_wino_poly.hip)
DOWN_POLY = True        # FIR-filtered stride-2 3x3 convolutions likewise (fir_down_planar + conv2d_down_poly)

MODE_SAME, MODE_DOWN2, MODE_UP2T = 0, 1, 2
_CONV_CLASS = {MODE_SAME: 'conv_mfma_s1', MODE_DOWN2: 'conv_mfma_s2', MODE_UP2T: 'conv_mfma_up'}


def conv2d(x, pw, mode=MODE_SAME, pad=0, in_scale=None, out_scale=None, bias=None, noise=None, noise_strength=1.0,
           act=False, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0, residual=None, planar=False):
    """x [NB, I, H, W] (for grouped weights NB = N*groups slots) -> y [NB, O, OH, OW].
    MODE_UP2T with ``planar`` returns the four sub-pixel phase planes [4, NB, O, H+1, W+1] for ``upfir_planar``."""
    L = _Launch()
    x = L.req(x, 'x')
    nb, i, h, w = x.shape
    if i != pw.i:
        raise _lib.ShgError(f'conv2d: x has {i} channels, weights expect {pw.i}')
    L.view(pw.wt, 'weights')
    if mode == MODE_SAME:
        oh, ow = h + 2 * pad - pw.kh + 1, w + 2 * pad - pw.kw + 1
    elif mode == MODE_DOWN2:
        oh, ow = (h + 2 * pad - pw.kh) // 2 + 1, (w + 2 * pad - pw.kw) // 2 + 1
    else:
        oh, ow = 2 * h + 1, 2 * w + 1
    lib = _lib.get_lib()
    if mode == MODE_UP2T and planar:
        y = L.new((4, nb, pw.o, h + 1, w + 1))
    else:
        y = L.new((nb, pw.o, oh, ow))
    noise, nmode = _noise_args(L, noise, nb)
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    residual = L.req(residual, 'residual')
    in_scale, out_scale, bias = L.req(in_scale, 'in_scale'), L.req(out_scale, 'out_scale'), L.req(bias, 'bias')
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('conv2d: residual shape mismatch')
    if (WINO and mode == MODE_SAME and pad == 1 and pw.kh == 3 and pw.kw == 3 and pw.groups == 1 and h >= WINO_MIN and w >= WINO_MIN
            and w % 4 == 0 and x.data_ptr() % 16 == 0 and i <= 1024 and (pw.wu is not None or pw._w is not None)):
        direct = 2.0 * nb * pw.o * i * 9 * oh * ow                                # direct-form (algorithmic) flops
        if WINO4 and h >= WINO4_MIN and lib.shg_conv2d_wino4_supported(nb, i, pw.o, h, w):
            wu = pw.wino4()
            executed = 2.0 * nb * pw.o * i * 36.0 * ((h + 3) // 4) * ((w + 3) // 4)
            ws_bytes = int(lib.shg_conv2d_wino4_workspace_bytes(nb, i, pw.o, pw.op, h, w)) if WINO_SPLIT else 0
            ws = L.new((ws_bytes // 4,)) if ws_bytes else None
            with _timed(L, 'conv_wino4', direct, executed):
                check(lib.shg_conv2d_wino4_ws_f32(
                    _ptr(x), _ptr(wu), _ptr(y), nb, i, pw.o, pw.op, h, w, _ptr(in_scale), _ptr(out_scale), _ptr(bias), _ptr(noise),
                    nmode, float(noise_strength), a, al, g, cl, _ptr(residual), _ptr(ws), ws_bytes, L.stream()), 'conv2d_wino4')
            return y
        wu = pw.wino()
        ws_bytes = int(lib.shg_conv2d_wino_workspace_bytes(nb, i, pw.o, pw.op, h, w)) if WINO_SPLIT else 0       # (small grids: split along the input channels)
        ws = L.new((ws_bytes // 4,)) if ws_bytes else None
        with _timed(L, 'conv_wino', direct, direct * 16.0 / 36.0):
            check(lib.shg_conv2d_wino_ws_f32(
                _ptr(x), _ptr(wu), _ptr(y), nb, i, pw.o, pw.op, h, w, _ptr(in_scale), _ptr(out_scale), _ptr(bias), _ptr(noise), nmode,
                float(noise_strength), a, al, g, cl, _ptr(residual), _ptr(ws), ws_bytes, L.stream()), 'conv2d_wino')
        return y
    if (UP_POLY and mode == MODE_UP2T and planar and pw.groups == 1 and pw._w is not None and out_scale is None and bias is None
            and noise is None and residual is None and not act and gain == 1.0 and x.data_ptr() % 16 == 0
            and lib.shg_conv2d_up_poly_supported(nb, i, pw.o, h, w)):
        wa, wb = pw.up_poly()
        nba, nbb = ((h + 1 + 2) // 3) * ((w + 1 + 2) // 3), (h // 2) * (w // 2)
        executed = 2.0 * nb * pw.o * i * (16.0 * (nba + nbb) + h + w)
        ws_bytes = int(lib.shg_conv2d_up_poly_workspace_bytes(nb, i, pw.o, pw.op, h, w)) if WINO_SPLIT else 0
        ws = L.new((ws_bytes // 4,)) if ws_bytes else None
        with _timed(L, 'conv_poly_up', 2.0 * nb * pw.o * i * 9 * h * w, executed):
            check(lib.shg_conv2d_up_poly_ws_f32(_ptr(x), _ptr(pw.wt), _ptr(wa), _ptr(wb), _ptr(y), nb, i, pw.o, pw.op, h, w,
                                                _ptr(in_scale), _ptr(ws), ws_bytes, L.stream()), 'conv2d_up_poly')
        return y
    ws, ws_bytes = None, 0
    if mode != MODE_UP2T or planar:      # split-K of the transposed conv is wired for the planar output only
        ws_bytes = int(lib.shg_conv2d_workspace_bytes(nb, i, pw.o, h, w, pw.kh, pw.kw, mode, pad, pw.groups))
        if ws_bytes:
            ws = L.new((ws_bytes // 4,))
    # algorithmic MACs: every (input pixel, tap) pair of the reference convolution, x2 flops
    pix = (h * w) if mode == MODE_UP2T else (oh * ow)
    with _timed(L, _CONV_CLASS[mode] if pw.kh == 3 else 'conv_mfma_1x1', 2.0 * nb * pw.o * i * pw.kh * pw.kw * pix):
        check(lib.shg_conv2d_f32(
            _ptr(x), _ptr(pw.wt), _ptr(y), nb, i, pw.o, pw.op, h, w, pw.kh, pw.kw, mode, pad, pw.groups, pw.wt.shape[1],
            _ptr(in_scale), _ptr(out_scale), _ptr(bias), _ptr(noise), nmode, float(noise_strength), a, al, g, cl, _ptr(residual),
            1 if planar else 0, _ptr(ws), ws_bytes, L.stream()), 'conv2d')
    return y


DOWN_POLY_MIN_I, DOWN_POLY_MIN_OUT = 128, 32     # below: the per-tile fixed cost of the 16-position kernels outweighs the saved
                                                  # multiplies (measured, tools/conv_bench_down.py), the direct stride-2 kernel is used


def down_poly_supported(x, pw, force=False):
    """True when the FIR + stride-2 3x3 convolution of x [N,I,H,W] runs in the polyphase-Winograd form."""
    n, i, h, w = x.shape
    if not force and (i < DOWN_POLY_MIN_I or min(h, w) // 2 < DOWN_POLY_MIN_OUT):
        return False
    return bool(DOWN_POLY and pw.groups == 1 and pw.kh == 3 and pw.kw == 3 and pw._w is not None and h % 2 == 0 and w % 2 == 0
                and x.is_cuda and _lib.get_lib().shg_conv2d_down_poly_supported(n, i, pw.o, h // 2, w // 2))


def fir_conv_down2(x, f, pw, bias=None, act=False, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0, residual=None, flip_filter=False):
    """y = act(conv3x3_stride2(upfirdn2d(x, f, padding=2)) + bias) (conv2d_resample.py:116-120 + the conv2d_layer tail):
    the 4x4 FIR writes the polyphase planes of its result, the convolution runs as two polyphase-Winograd launches."""
    L = _Launch()
    x, f = L.req(x, 'x'), L.req(f, 'f')
    if tuple(f.shape) != (4, 4):
        raise _lib.ShgError('fir_conv_down2: f must be 4x4')
    n, i, h, w = x.shape
    oh, ow = h // 2, w // 2
    lib = _lib.get_lib()
    # plane pitch: whole 128-byte lines per row for the wide planes of the marching FIR (its stores want that), else 16 bytes
    pp4 = (ow + 1 + 3) // 4 * 4
    taps = sep_taps(f) if lib.shg_fir_pad2_sep_supported(h, w, pp4) else None
    pp = (ow + 1 + 31) // 32 * 32 if (taps is not None and w % 256 == 0) else pp4
    wa, wb = pw.down_poly()
    L.view(wa, 'weights')
    xp = L.new((4, n, i, oh + 1, pp))
    y = L.new((n, pw.o, oh, ow))
    bias, residual = L.req(bias, 'bias'), L.req(residual, 'residual')
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('fir_conv_down2: residual shape mismatch')
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    with _timed(L, 'upfirdn2d', 4.0 * (x.numel() + n * i * (h + 1) * (w + 1))):
        if taps is not None:
            check(lib.shg_fir_pad2_sep_f32(_ptr(x), taps, _ptr(xp), n, i, h, w, pp, int(bool(flip_filter)), 1.0, L.stream()), 'fir_pad2_sep')
        else:
            check(lib.shg_fir_down_planar_f32(_ptr(x), _ptr(f), _ptr(xp), n, i, h, w, pp, int(bool(flip_filter)), 1.0, L.stream()),
                  'fir_down_planar')
    nba, nbb = ((oh + 2) // 3) * ((ow + 2) // 3), (oh // 2) * (ow // 2)
    with _timed(L, 'conv_poly_down', 2.0 * n * pw.o * i * 9 * oh * ow, 2.0 * n * pw.o * i * 16.0 * (nba + nbb)):
        check(lib.shg_conv2d_down_poly_f32(_ptr(xp), _ptr(wa), _ptr(wb), _ptr(y), n, i, pw.o, pw.op, oh, ow, pp, None, _ptr(bias),
                                           a, al, g, cl, _ptr(residual), L.stream()), 'conv2d_down_poly')
    return y


def upfir_planar(mid, f, scale=None, bias=None, noise=None, noise_strength=1.0, residual=None, act=False, gain=1.0,
                 alpha=0.2, act_gain=SQRT2, clamp=256.0, fir_gain=4.0, flip=False):
    """mid [4,N,C,H+1,W+1] (phase planes of the transposed conv) -> y [N,C,2H,2W]: 4x4 FIR (pad 1) + fused layer tail."""
    L = _Launch()
    mid = L.req(mid, 'mid')
    f = L.req(f, 'f')
    if mid.ndim != 5 or mid.shape[0] != 4 or tuple(f.shape) != (4, 4):
        raise _lib.ShgError('upfir_planar: mid must be [4,N,C,H+1,W+1] and f 4x4')
    _, n, c, hp, wp = mid.shape
    h, w = hp - 1, wp - 1
    y = L.new((n, c, 2 * h, 2 * w))
    noise, nmode = _noise_args(L, noise, n)
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    residual = L.req(residual, 'residual')
    scale, bias = L.req(scale, 'scale'), L.req(bias, 'bias')
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.ShgError('upfir_planar: residual shape mismatch')
    with _timed(L, 'fir_up_planar', 4.0 * (mid.numel() + y.numel() * (2 if residual is not None else 1))):
        lib = _lib.get_lib()
        taps = sep_taps(f) if lib.shg_upfir_planar_sep_supported(h, w) else None
        if taps is not None and all(t is None or t.data_ptr() % 16 == 0 for t in (y, noise, residual)):
            check(lib.shg_upfir_planar_sep_f32(_ptr(mid), taps, _ptr(y), n, c, h, w, int(bool(flip)), float(fir_gain),
                                               _ptr(scale), _ptr(bias), _ptr(noise), nmode, float(noise_strength), a, al, g, cl,
                                               _ptr(residual), L.stream()), 'upfir_planar_sep')
        else:
            check(lib.shg_upfir_planar_f32(_ptr(mid), _ptr(f), _ptr(y), n, c, h, w, int(bool(flip)), float(fir_gain),
                                           _ptr(scale), _ptr(bias), _ptr(noise), nmode, float(noise_strength), a, al, g, cl,
                                           _ptr(residual), L.stream()), 'upfir_planar')
    return y


def conv1x1_thin_in(x, w, bias=None, wgain=1.0, act=True, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0):
    L = _Launch()
    x = L.req(x, 'x')
    w = L.req(w, 'w')
    bias = L.req(bias, 'bias')
    n, i, h, wd = x.shape
    o = w.shape[0]
    y = L.new((n, o, h, wd))
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    with _timed(L, 'fromrgb', 4.0 * (x.numel() + y.numel())):
        check(_lib.get_lib().shg_conv1x1_thin_in_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, i, o, h * wd, float(wgain), a, al, g,
                                                     cl, L.stream()), 'conv1x1_thin_in')
    return y


def torgb(x, w, styles=None, bias=None, base_up=None, f=None):
    L = _Launch()
    x = L.req(x, 'x')
    w = L.req(w, 'w')
    styles, bias, base_up, f = L.req(styles, 'styles'), L.req(bias, 'bias'), L.req(base_up, 'base_up'), L.req(f, 'f')
    n, i, h, wd = x.shape
    o = w.shape[0]
    y = L.new((n, o, h, wd))
    with _timed(L, 'torgb', 4.0 * (x.numel() + y.numel() + (base_up.numel() if base_up is not None else 0))):
        check(_lib.get_lib().shg_torgb_f32(_ptr(x), _ptr(w), _ptr(styles), _ptr(bias), _ptr(base_up), _ptr(f), _ptr(y), n, i, o, h,
                                           wd, L.stream()), 'torgb')
    return y


# ------------------------------------------------------------------------------------------------
# dense / style
# ------------------------------------------------------------------------------------------------

def dense(x, w, b=None, wgain=1.0, bgain=1.0, act=False, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0, out=None):
    L = _Launch()
    x = L.req(x, 'x')
    w = L.req(w, 'w')
    b = L.req(b, 'b')
    n, k = x.shape
    o = w.shape[0]
    if w.shape[1] != k:
        raise _lib.ShgError(f'dense: x has {k} features, weight expects {w.shape[1]}')
    y = L.rows(out, 'out') if out is not None else L.new((n, o))
    a, al, g, cl = _act_args(act, gain, alpha, act_gain, clamp)
    with _timed(L, 'dense', 4.0 * (w.numel() + x.numel() + n * o)):
        check(_lib.get_lib().shg_dense_f32(_ptr(x), _ptr(w), _ptr(b), _ptr(y), n, k, o, x.stride(0), y.stride(0), float(wgain),
                                           float(bgain), a, al, g, cl, L.stream()), 'dense')
    return y


def demod_weight(w, prenorm=False):
    """-> (wn, wsq [O,I], sfac [O]): the weight side of modulated_conv2d (csrc/dense.hip: demod_weight_kernel)."""
    L = _Launch()
    w = L.req(w, 'w')
    o, i = w.shape[0], w.shape[1]
    k = w.numel() // (o * i)
    wn, wsq, sfac = torch.empty_like(w), L.new((o, i)), L.new((o,))
    with L:
        check(_lib.get_lib().shg_demod_weight_f32(_ptr(w), _ptr(wn), _ptr(wsq), _ptr(sfac), o, i, k, int(bool(prenorm)), L.stream()), 'demod_weight')
    return wn, wsq, sfac


def demod_weight_backward(wn, sfac, gwn, gwsq):
    L = _Launch()
    wn, sfac, gwn, gwsq = L.req(wn, 'wn'), L.req(sfac, 'sfac'), L.req(gwn, 'gwn'), L.req(gwsq, 'gwsq')
    o, i = wn.shape[0], wn.shape[1]
    k = wn.numel() // (o * i)
    gw = torch.empty_like(wn)
    with L:
        check(_lib.get_lib().shg_demod_weight_backward_f32(_ptr(wn), _ptr(sfac), _ptr(gwn), _ptr(gwsq), _ptr(gw), o, i, k, L.stream()),
              'demod_weight_backward')
    return gw


def style_factors_supported(styles, wsq):
    n, i = styles.shape
    return (styles.is_cuda and styles.dtype == torch.float32 and styles.ndim == 2 and wsq is not None and wsq.ndim == 2 and wsq.shape[1] == i
            and n <= 32 and n * i <= 8192 and i <= 1024)


def style_factors(styles, wsq, prenorm=False):
    """-> (sn [N,I], dcoefs [N,O], aux [N+1]): the style side of modulated_conv2d (csrc/dense.hip: style_factors_kernel)."""
    L = _Launch()
    styles, wsq = L.req(styles, 'styles'), L.req(wsq, 'wsq')
    n, i = styles.shape
    o = wsq.shape[0]
    sn, d, aux = L.new((n, i)), L.new((n, o)), L.new((n + 1,))
    with L:
        check(_lib.get_lib().shg_style_factors_f32(_ptr(styles), _ptr(wsq), _ptr(sn), _ptr(d), _ptr(aux), n, i, o, int(bool(prenorm)), L.stream()),
              'style_factors')
    return sn, d, aux


def style_factors_backward(sn, d, wsq, aux, gsn, gd, prenorm=False, want_wsq=True):
    """-> (g_styles [N,I], g_wsq [O,I] | None)."""
    L = _Launch()
    sn, d, wsq, aux, gsn, gd = L.req(sn, 'sn'), L.req(d, 'd'), L.req(wsq, 'wsq'), L.req(aux, 'aux'), L.req(gsn, 'gsn'), L.req(gd, 'gd')
    n, i = sn.shape
    o = wsq.shape[0]
    gs, gw, part = L.new((n, i)), (L.new((o, i)) if want_wsq else None), L.new(((o + 63) // 64, n, i))
    with L:
        check(_lib.get_lib().shg_style_factors_backward_f32(_ptr(sn), _ptr(d), _ptr(wsq), _ptr(aux), _ptr(gsn), _ptr(gd), _ptr(gs), _ptr(gw), _ptr(part),
                                                            n, i, o, int(bool(prenorm)), L.stream()), 'style_factors_backward')
    return gs, gw


def matmul_nn(a, b, scale=1.0):
    """scale * a[N,M] @ b[M,K] (csrc/dense.hip: the input gradient of a dense layer)."""
    L = _Launch()
    a, b = L.req(a, 'a'), L.req(b, 'b')
    n, m = a.shape
    if b.shape[0] != m:
        raise _lib.ShgError(f'matmul_nn: a is {tuple(a.shape)}, b is {tuple(b.shape)}')
    k = b.shape[1]
    out = L.new((n, k))
    with _timed(L, 'dense', 4.0 * (a.numel() + b.numel() + out.numel())):
        check(_lib.get_lib().shg_matmul_nn_f32(_ptr(a), _ptr(b), _ptr(out), n, m, k, a.stride(0), out.stride(0), float(scale), L.stream()), 'matmul_nn')
    return out


def matmul_tn(a, b, scale=1.0, colsum_scale=None):
    """-> (scale * a[N,M]^T @ b[N,K], colsum_scale * a.sum(0) or None): weight and bias gradient of a dense layer in one pass."""
    L = _Launch()
    a, b = L.req(a, 'a'), L.req(b, 'b')
    n, m = a.shape
    if b.shape[0] != n:
        raise _lib.ShgError(f'matmul_tn: a is {tuple(a.shape)}, b is {tuple(b.shape)}')
    k = b.shape[1]
    out = L.new((m, k))
    col = L.new((m,)) if colsum_scale is not None else None
    with _timed(L, 'dense', 4.0 * (a.numel() + b.numel() + out.numel())):
        check(_lib.get_lib().shg_matmul_tn_f32(_ptr(a), _ptr(b), _ptr(out), _ptr(col), n, m, k, a.stride(0), b.stride(0), float(scale),
                                               float(colsum_scale or 0.0), L.stream()), 'matmul_tn')
    return out, col


def normalize_2nd_moment(x, eps=1e-8):
    L = _Launch()
    x = L.req(x, 'x')
    y = torch.empty_like(x)
    with L:
        check(_lib.get_lib().shg_normalize_2nd_moment_f32(_ptr(x), _ptr(y), x.shape[0], x.shape[1], float(eps), L.stream()),
              'normalize_2nd_moment')
    return y


def modconv_style_prep(styles, pw=None, demod=True, pre_gain=1.0):
    """-> (s [N,I], dcoef [N,O] or None)."""
    L = _Launch()
    styles = L.req(styles, 'styles')
    n, i = styles.shape
    s = L.new((n, i))
    d = None
    o = op = 0
    wsq = None
    if demod:
        o, op, wsq = pw.o, pw.op, L.view(pw.wsq, 'wsq')
        d = L.new((n, o))
    with L:
        check(_lib.get_lib().shg_modconv_style_prep_f32(_ptr(styles), styles.stride(0), _ptr(wsq), _ptr(s), _ptr(d), n, i, o, op,
                                                        int(bool(demod)), float(pre_gain), L.stream()), 'modconv_style_prep')
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
        L = _Launch()
        nbytes = 0.0
        for g, it in zip(arr, chunk):
            x1, x2, w, y = L.rows(it['x1'], 'x1'), L.rows(it.get('x2'), 'x2'), L.req(it['w'], 'w'), L.rows(it['y'], 'y')
            b = L.req(it.get('b'), 'b')
            k1 = x1.shape[1]
            k2 = 0 if x2 is None else x2.shape[1]
            if x1.shape[0] != n or y.shape[0] != n or w.shape[1] != k1 + k2 or y.shape[1] != w.shape[0]:
                raise _lib.ShgError('dense_grouped: inconsistent shapes')
            g.x1, g.x2, g.w, g.b, g.y = _addr(x1), _addr(x2), _addr(w), _addr(b), _addr(y)
            g.ld1, g.ld2, g.K1, g.K2, g.O, g.ldy = x1.stride(0), (x2.stride(0) if x2 is not None else 0), k1, k2, w.shape[0], y.stride(0)
            g.wgain, g.bgain = float(it.get('wgain', 1.0)), float(it.get('bgain', 1.0))
            nbytes += 4.0 * w.numel()
        with _timed(L, 'dense_grouped', nbytes):
            check(_lib.get_lib().shg_dense_grouped_f32(arr, len(chunk), n, L.stream()), 'dense_grouped')


def modconv_style_prep_grouped(items):
    """items: list of dicts(styles [N,I], pw|None, demod, pre_gain, s [N,I] out, d [N,O] out|None) -- one launch per 32."""
    if not items:
        return
    n = items[0]['styles'].shape[0]
    for lo in range(0, len(items), MAX_GROUPS):
        chunk = items[lo:lo + MAX_GROUPS]
        arr = (_lib.StyleGroup * len(chunk))()
        L = _Launch()
        for g, it in zip(arr, chunk):
            st, s_out = L.rows(it['styles'], 'styles'), L.req(it['s'], 's')
            demod = bool(it.get('demod', True))
            pw = it.get('pw')
            g.styles, g.s_out = _addr(st), _addr(s_out)
            g.wsq = _addr(L.view(pw.wsq, 'wsq')) if demod else None
            g.dcoef = _addr(L.req(it['d'], 'd')) if demod else None
            g.ld, g.I = st.stride(0), st.shape[1]
            g.O, g.OP = (pw.o, pw.op) if demod else (0, 0)
            g.demod, g.pre_gain = int(demod), float(it.get('pre_gain', 1.0))
        with L:
            check(_lib.get_lib().shg_modconv_style_prep_grouped_f32(arr, len(chunk), n, L.stream()), 'modconv_style_prep_grouped')


# ------------------------------------------------------------------------------------------------
# SHU
# ------------------------------------------------------------------------------------------------

def shu_rfft2_shift(x):
    """x: [N,C,64,64] view whose channel planes are contiguous (a channel slice of an NCHW tensor is fine)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and tuple(x.shape[2:]) == (64, 64)):
        raise _lib.ShgError('shu_rfft2_shift: x must be a float32 HIP tensor [N,C,64,64]')
    if x.stride(3) != 1 or x.stride(2) != 64 or x.stride(1) != 4096:
        x = x.contiguous()
    L = _Launch()
    L.view(x, 'x')
    n, c = x.shape[:2]
    t = L.new((n, 2 * c, 64, 33))
    with _timed(L, 'shu_rfft2', 4.0 * (x.numel() + t.numel())):
        check(_lib.get_lib().shg_shu_rfft2_shift_f32(_ptr(x), x.stride(0), _ptr(t), n, c, L.stream()), 'shu_rfft2_shift')
    return t


def shu_split_adjoint(grads, gauss, n, c):
    """Transpose of ``shu_split_irfft2`` (bands == 1): grads = 5 tensors [N,C,r,r] (r = 4..64; None = no gradient) -> [N,2C,64,33]."""
    L = _Launch()
    g_arr = (ctypes.c_void_p * 5)()
    s_arr = (ctypes.c_long * 5)()
    t_arr = (ctypes.c_void_p * 5)()
    for l in range(5):
        r = 4 << l
        t_arr[l] = L.req(gauss[l], 'gauss').data_ptr()
        g = grads[l]
        if g is None:
            g_arr[l] = None
            continue
        g = L.req(g, 'grad')
        if tuple(g.shape) != (n, c, r, r):
            raise _lib.ShgError(f'shu_split_adjoint: grads[{l}] must be [N,{c},{r},{r}]')
        g_arr[l] = g.data_ptr()
        s_arr[l] = g.stride(0)
    out = L.new((n, 2 * c, 64, 33))
    with _timed(L, 'shu_split_adjoint', 4.0 * out.numel()):
        check(_lib.get_lib().shg_shu_split_adjoint_f32(g_arr, s_arr, t_arr, _ptr(out), n, c, L.stream()), 'shu_split_adjoint')
    return out


WGRAD_WINO = True        # stride-1 3x3 'same' layers: weight gradient in the Winograd domain (csrc/conv_wgrad_wino.hip); False = the direct kernel


def conv2d_wgrad(x, g, kh, kw, stride=1, pad=0):
    """Weight gradient of y = conv2d(x, w, stride, pad): x [NB,I,H,W], g = dL/dy [NB,O,OH,OW] -> dw [O,I,kh,kw]
    (shg_conv2d_wgrad_f32, or shg_conv2d_wgrad_wino_f32 where it applies; for conv_transpose2d call it with (dL/dy, x) and read the
    result as [Cin,Cout,kh,kw])."""
    L = _Launch()
    x, g = L.req(x, 'x'), L.req(g, 'g')
    nb, i, h, w = x.shape
    nb2, o, oh, ow = g.shape
    if nb != nb2:
        raise _lib.ShgError('conv2d_wgrad: batch sizes differ')
    lib = _lib.get_lib()
    dw = L.new((o, i, kh, kw))
    flops = 2.0 * nb * o * i * kh * kw * oh * ow
    if (WGRAD_WINO and lib.shg_conv2d_wgrad_wino_supported(h, w, oh, ow, kh, kw, int(stride), int(pad)) and x.data_ptr() % 16 == 0
            and g.data_ptr() % 16 == 0 and max(i, o) * h * w * 4 < 2 ** 31):
        ws_bytes = int(lib.shg_conv2d_wgrad_wino_workspace_bytes(nb, i, o, h, w))
        ws = L.new((ws_bytes // 4,)) if ws_bytes else None
        with _timed(L, 'conv_wgrad_wino', flops, executed=flops / 4):
            check(lib.shg_conv2d_wgrad_wino_f32(_ptr(x), _ptr(g), _ptr(dw), nb, i, o, h, w, _ptr(ws), ws_bytes, L.stream()), 'conv2d_wgrad_wino')
        return dw
    ws_bytes = int(lib.shg_conv2d_wgrad_workspace_bytes(nb, i, o, oh, ow, kh, kw))
    ws = L.new((ws_bytes // 4,)) if ws_bytes else None
    with _timed(L, 'conv_wgrad', flops):
        check(lib.shg_conv2d_wgrad_f32(_ptr(x), _ptr(g), _ptr(dw), nb, i, o, h, w, oh, ow, kh, kw, int(stride), int(pad), _ptr(ws), ws_bytes,
                                       L.stream()), 'conv2d_wgrad')
    return dw


def mfma_pack_rows(w):
    """w [..., 64, 64] (rows = output channels) -> [..., 32, 2, 64]: the A operands of v_mfma_f32_32x32x2_f32 per k-step,
    element [ks][mo][l] = w[mo*32 + (l & 31)][2*ks + (l >> 5)] (layout of shg_shu_spectral_f32)."""
    l = torch.arange(64, device=w.device)
    rows = torch.arange(2, device=w.device)[:, None] * 32 + (l & 31)[None, :]            # [2, 64]
    cols = 2 * torch.arange(32, device=w.device)[:, None] + (l >> 5)[None, :]             # [32, 64]
    return w[..., rows[None, :, :], cols[:, None, :]].contiguous()


def shu_spectral(t, w0p, b0, w1p, cw):
    """t [N,64,64,33] -> S [N,64,64,33]: conv0 + ReLU + heterogeneous filter + band sum in one launch (shg_shu_spectral_f32)."""
    L = _Launch()
    t, w0p, b0, w1p, cw = L.req(t, 't'), L.req(w0p, 'w0p'), L.req(b0, 'b0'), L.req(w1p, 'w1p'), L.req(cw, 'cw')
    n, c2, h, w = t.shape
    bands = cw.shape[0]
    if tuple(w0p.shape) != (32, 2, 64) or tuple(w1p.shape) != (bands * 32, 2, 64) or cw.numel() != bands * h * w:
        raise _lib.ShgError('shu_spectral: packed weight / cw shapes do not match')
    out = L.new((n, c2, h, w))
    # matrix work: conv0 (c2 x c2) + the band filter (bands c2 x c2) per spectral pixel, fp32 MFMA
    with _timed(L, 'shu_spectral', 2.0 * n * h * w * c2 * c2 * (1 + bands)):
        check(_lib.get_lib().shg_shu_spectral_f32(_ptr(t), _ptr(w0p), _ptr(b0), _ptr(w1p), _ptr(cw), _ptr(out), n, c2, h * w, bands,
                                                  L.stream()), 'shu_spectral')
    return out


def shu_split_irfft2(y, cw, gauss, outs, accumulate):
    """y: [N, 2C*B, 64, 33]; cw: [B,64,33] or None; gauss: list of 5 tables (r=4..64);
    outs: list of 5 tensors/views [N,C,r,r] with contiguous channel planes (or None to skip)."""
    L = _Launch()
    y = L.req(y, 'y')
    cw = L.req(cw, 'cw')
    bands = cw.shape[0] if cw is not None else 1
    n = y.shape[0]
    c = y.shape[1] // (2 * bands)
    g_arr = (ctypes.c_void_p * 5)()
    o_arr = (ctypes.c_void_p * 5)()
    s_arr = (ctypes.c_long * 5)()
    nbytes = 4.0 * y.numel()
    for l in range(5):
        r = 4 << l
        g = L.req(gauss[l], 'gauss')
        g_arr[l] = g.data_ptr()
        o = outs[l]
        if o is None:
            o_arr[l] = None
            continue
        if not (o.is_cuda and o.dtype == torch.float32 and tuple(o.shape) == (n, c, r, r)
                and o.stride(3) == 1 and o.stride(2) == r and o.stride(1) == r * r):
            raise _lib.ShgError(f'shu_split_irfft2: out[{l}] must be a float32 [N,{c},{r},{r}] view with contiguous planes')
        L.view(o, 'out')
        o_arr[l] = o.data_ptr()
        s_arr[l] = o.stride(0)
        nbytes += 4.0 * o.numel() * (2 if accumulate else 1)
    with _timed(L, 'shu_irfft2', nbytes):
        check(_lib.get_lib().shg_shu_split_irfft2_f32(_ptr(y), _ptr(cw), g_arr, o_arr, s_arr, n, c, bands, int(bool(accumulate)),
                                                      L.stream()), 'shu_split_irfft2')
    return outs
