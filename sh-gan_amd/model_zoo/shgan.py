"""SH-GAN encoder with the Spectral Hint Unit, host side.  Mirrors the reference's
lib/model_zoo/shgan.py: ``one_hot_2d`` :18, ``make_cweight`` :70, ``heterogeneous_filter`` :123,
``gaussian_heatmap_2d`` :162, ``SHU`` :252, ``Encoder`` :338 (registry name 'shgan_encoder').

MI355X design: the constant tables (band weights ``cw`` and the Gaussian-split maps) are built once
on the host and kept resident on the device as non-persistent buffers -- the reference rebuilds
``cw`` lazily and re-uploads the Gaussian maps on every forward (shgan.py:145-155,329).  The forward
is four kernels: rFFT2+shift, two single-tap MFMA convolutions (64->64 +bias+ReLU, 64->384), and one
fused band-sum + Gaussian split + unshift + irFFT2 x5 that adds the hints straight into the encoder
feature maps (shgan.py:378-382)."""
import numpy as np
import torch
import torch.nn as nn

from .. import kernels
from .comodgan import Encoder as Encoder_base
from .common.get_model import register
from .stylegan import _cache_of
from .stylegan_utils import grad_ops

version = '0'
symbol = 'shgan'


class one_hot_2d(object):
    """[bs,h,w] integer ids -> [bs,max_dim,h,w] uint8 one-hot (shgan.py:18-68)."""

    def __init__(self, max_dim=None, ignore_label=None, **kwargs):
        self.max_dim = max_dim
        self.ignore_label = [ignore_label] if isinstance(ignore_label, int) else list(ignore_label or [])

    def __call__(self, x, mask=None):
        x = np.asarray(x)
        if mask is not None:
            x = x * (mask == 1)
        present = [i for i, cnt in enumerate(np.bincount(x.flatten())) if cnt > 0 and i not in self.ignore_label]
        max_dim = self.max_dim if self.max_dim is not None else present[-1] + 1
        oh = np.zeros((x.shape[0], max_dim) + x.shape[1:], dtype=np.uint8)
        for cid in present:
            if cid < max_dim:
                oh[:, cid] = x == cid
        if mask is not None:
            oh[:, 0] *= (mask == 1).astype(np.uint8)
        return oh


def _interp_matrix(pos, size, kind):
    """[len(pos), size] weights of `F.grid_sample(..., padding_mode='border', align_corners=True)` along one axis for pixel
    coordinates `pos` (already clipped to [0, size-1]): the bilinear hat, or the cubic convolution kernel (A = -0.75) on the four
    taps floor(p)-1 .. floor(p)+2 with out-of-range taps clamped to the border."""
    m = np.zeros((len(pos), size), dtype=np.float64)
    if kind == 'piecewise_linear':
        for j in range(size):
            m[:, j] = np.clip(1 - np.abs(pos - j), 0, 1)
        return m
    A = -0.75
    i0 = np.floor(pos).astype(np.int64)
    t = pos - i0

    def c1(x):                     # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):                     # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    coef = [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]
    for k in range(4):
        idx = np.clip(i0 - 1 + k, 0, size - 1)
        np.add.at(m, (np.arange(len(pos)), idx), coef[k])
    return m


def make_cweight(half_size, half_sample, type='piecewise_linear', oddeven_aligned=True, device='cpu'):
    """Band-weight table cw[h0*w0, hs, ws] (shgan.py:70-121): every cell of an h0 x w0 control grid
    is a one-hot image, mirrored along w so it spans [-1,1], and sampled bilinearly or bicubically
    (align_corners, border clamp) at rows -1+2(i+1)/hs (even hs, 'oddeven aligned') and columns
    i/(ws-1).  Evaluated here in closed form with numpy (both interpolations are separable) -- the result is a
    partition of unity."""
    if type not in ('piecewise_linear', 'bicubic'):
        raise NotImplementedError("heterogeneous filter types: 'piecewise_linear', 'bicubic' (shgan.py:113-120)")
    h0, w0 = half_size
    hs, ws = half_sample
    wpad = 2 * w0 - 1                                    # reflect pad on the left by w0-1
    if oddeven_aligned and hs % 2 == 0:
        gy = np.array([-1 + (i + 1) / hs * 2 for i in range(hs)], dtype=np.float64)
    else:
        gy = np.array([-1 + i / (hs - 1) * 2 for i in range(hs)], dtype=np.float64)
    gx = np.array([i / (ws - 1) for i in range(ws)], dtype=np.float64)
    # align_corners=True: pixel coordinate = (g+1)/2 * (size-1)
    py = np.clip((gy.astype(np.float32).astype(np.float64) + 1) / 2 * (h0 - 1), 0, h0 - 1)
    px = np.clip((gx.astype(np.float32).astype(np.float64) + 1) / 2 * (wpad - 1), 0, wpad - 1)
    my, mx = _interp_matrix(py, h0, type), _interp_matrix(px, wpad, type)      # [hs, h0], [ws, wpad]
    cw = np.zeros((h0 * w0, hs, ws), dtype=np.float64)
    for a in range(h0):
        for b in range(w0):
            # padded columns: index j in [0, wpad) maps to original column |j - (w0-1)|
            wx = np.zeros(ws)
            for j in range(wpad):
                if abs(j - (w0 - 1)) == b:
                    wx += mx[:, j]
            cw[a * w0 + b] = my[:, a][:, None] * wx[None, :]
    return torch.tensor(cw, dtype=torch.float32, device=device)


class heterogeneous_filter(nn.Module):
    """1x1 conv C -> C*fh*fw whose outputs are blended per spectral position by ``cw``
    (shgan.py:123-160).  Holds ``weight`` [in, out*fh*fw]; standalone forward runs the two HIP
    kernels (MFMA single-tap conv + band sum)."""

    def __init__(self, in_channels, out_channels, freedom, type, init='ones'):
        super().__init__()
        self.in_channels, self.out_channels, self.freedom, self.type = in_channels, out_channels, freedom, type
        if type not in ('piecewise_linear', 'bicubic'):       # (shgan.py:133-137)
            raise NotImplementedError
        fh, fw = freedom
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels * fh * fw), requires_grad=True)
        if init == 'ones':
            nn.init.ones_(self.weight)

    def prepped(self):
        # weight is [I][O*B]; as a single-tap convolution its [O*B, I, 1, 1] form goes through the usual weight prep
        def build():
            i, ob = self.weight.shape
            return kernels.conv_weight_prep(self.weight.detach().t().contiguous().reshape(ob, i, 1, 1))
        return _cache_of(self).get('w', [self.weight], build)

    def band_conv(self, x):
        n, c, h, w = x.shape
        if (h * w) % 32 != 0:
            raise NotImplementedError('heterogeneous_filter: H*W must be a multiple of 32')
        y = kernels.conv2d(x.reshape(n, c, (h * w) // 32, 32), self.prepped(), mode=kernels.MODE_SAME, pad=0)
        return y.reshape(n, -1, h, w)

    def cweight(self, h, w, device):
        """Band-weight table for an [h, w] half spectrum, built once per geometry (shgan.py:145-155)."""
        key = (h, w, str(device))
        if getattr(self, '_cw_key', None) != key:
            self.__dict__['_cw_val'] = make_cweight(self.freedom, (h, w), self.type, device=device)
            self.__dict__['_cw_key'] = key
        return self.__dict__['_cw_val']

    def forward(self, x):
        """Standalone form (shgan.py:143-160): y = sum_k conv1x1(x, W)[:, :, k] * cw[k].  SHU.forward does not come through
        here -- it fuses this band sum with the Gaussian split and the inverse FFTs in one kernel."""
        n, c, h, w = x.shape
        if grad_ops.wants_grad(x, self.weight):
            # training rows (shgan.py:143-160 verbatim in structure): 1x1 convolution to O*B channels on the HIP conv (forward
            # and backward), view [O, B], weighted sum over the bands
            from .stylegan_utils import conv2d_gradfix
            y = conv2d_gradfix.conv2d(x, self.weight.t().reshape(-1, c, 1, 1)).view(n, self.out_channels, -1, h, w)
            return (y * self.cweight(h, w, x.device)[None, None]).sum(2)
        y = self.band_conv(x).view(n, self.out_channels, -1, h, w)
        cw = self.cweight(h, w, x.device)
        out = None
        for k in range(cw.shape[0]):     # plain fma launches: a cold path kept for drop-in completeness
            term = y[:, :, k].contiguous()
            wk = cw[k].expand_as(term).contiguous()
            out = kernels.fma(term, wk, out if out is not None else torch.zeros_like(term))
        return out


class gaussian_heatmap_2d(object):
    """Sum/max of 2-D Gaussians rendered on an [h,w] grid, each evaluated only inside a window of
    +-int(3*sigma_max+1) around its integer centre (shgan.py:162-250)."""

    def __init__(self, size, merge_type='max'):
        self.size = size
        self.merge_type = merge_type
        self.speedup = True

    def __call__(self, c, v):
        if c.shape[0] != v.shape[0]:
            raise ValueError
        h, w = self.size
        out = np.zeros((h, w), dtype=float)
        for ci, vi in zip(c, v):
            try:
                radius = int(3 * np.sqrt(np.max(np.linalg.svd(vi, compute_uv=False))) + 1)
                vinv = np.linalg.inv(vi)
            except np.linalg.LinAlgError:
                continue
            ch, cw_ = int(ci[0]), int(ci[1])
            h0, h1 = (min(max(t, 0), h) for t in (ch - radius, ch + radius))
            w0, w1 = (min(max(t, 0), w) for t in (cw_ - radius, cw_ + radius))
            if not self.speedup:
                h0, h1, w0, w1 = 0, h, 0, w
            if h1 <= h0 or w1 <= w0:
                continue
            dy = (np.arange(h0, h1) - ci[0])[:, None]
            dx = (np.arange(w0, w1) - ci[1])[None, :]
            q = vinv[0, 0] * dy * dy + (vinv[0, 1] + vinv[1, 0]) * dy * dx + vinv[1, 1] * dx * dx
            val = np.exp(-0.5 * q)
            if self.merge_type == 'max':
                out[h0:h1, w0:w1] = np.maximum(out[h0:h1, w0:w1], val)
            elif self.merge_type == 'add':
                out[h0:h1, w0:w1] += val
            else:
                raise ValueError
        return out


def _adjoint_table(device):
    """(1/c_k) / 4096 on [64,33], c_0 = c_32 = 1, else 2: with it the 64 x 64 level of ``shu_split_irfft2`` is the transpose of
    ``shu_rfft2_shift`` (rfft2 with norm='forward' keeps half of a Hermitian spectrum; irfft2 counts the interior columns twice)."""
    t = _ADJ_TABLE.get(str(device))
    if t is None:
        w = torch.full((64, 33), 0.5 / 4096.0)
        w[:, 0] = w[:, 32] = 1.0 / 4096.0
        t = _ADJ_TABLE[str(device)] = w.to(device)
    return t


_ADJ_TABLE = {}
_LEVELS = (4, 8, 16, 32, 64)


class _ShuSpectrum(torch.autograd.Function):
    """x [N,C,64,64] -> [N,2C,64,33] (rfft2, norm='forward', rows shifted: shgan.py:313-319); backward = its transpose."""
    @staticmethod
    def forward(ctx, x):
        return kernels.shu_rfft2_shift(x.detach())

    @staticmethod
    def backward(ctx, g):
        return _ShuSpectrumT.apply(g.contiguous())


class _ShuSpectrumT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g):
        g = g.detach()
        n, c = g.shape[0], g.shape[1] // 2
        out = torch.empty((n, c, 64, 64), device=g.device, dtype=torch.float32)
        tab = _adjoint_table(g.device)
        kernels.shu_split_irfft2(g, None, [tab] * 5, [None, None, None, None, out], accumulate=False)
        return out

    @staticmethod
    def backward(ctx, gx):
        return _ShuSpectrum.apply(gx.contiguous())


class _ShuSplit(torch.autograd.Function):
    """S [N,2C,64,33] -> the five hints [N,C,r,r] (crop, Gaussian split, un-shift, irfft2: shgan.py:326-336); backward = its transpose."""
    @staticmethod
    def forward(ctx, s, *gauss):
        s = s.detach()
        n, c = s.shape[0], s.shape[1] // 2
        ctx.gauss = gauss
        outs = [torch.empty((n, c, r, r), device=s.device, dtype=torch.float32) for r in _LEVELS]
        kernels.shu_split_irfft2(s, None, list(gauss), outs, accumulate=False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        return (_ShuSplitT.apply(*[None if g is None else g.contiguous() for g in grads], *ctx.gauss),) + (None,) * 5


class _ShuSplitT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g4, g8, g16, g32, g64, *gauss):
        grads = [None if g is None else g.detach() for g in (g4, g8, g16, g32, g64)]
        ctx.gauss = gauss
        ctx.present = [g is not None for g in grads]
        ref = next(g for g in grads if g is not None)
        return kernels.shu_split_adjoint(grads, list(gauss), ref.shape[0], ref.shape[1])

    @staticmethod
    def backward(ctx, gs):
        outs = _ShuSplit.apply(gs.contiguous(), *ctx.gauss)
        return tuple(o if pr else None for o, pr in zip(outs, ctx.present)) + (None,) * 5


class SHU(nn.Module):
    """Spectral Hint Unit (shgan.py:252-336): rFFT2 -> 1x1 conv + ReLU -> heterogeneous filter ->
    Gaussian split into a pyramid of bands -> irFFT2 per band.  x [N,C,64,64] -> {r: [N,C,r,r]}."""

    def __init__(self, in_channels, out_channels, dfilter_freedom=[3, 2], dfilter_type='piecewise_linear', input_res=256,
                 lowest_res=4, tail_sigma_mult=3, gaussian_at_input_res=False):
        super().__init__()
        from .stylegan import conv2d
        if input_res != 64 or lowest_res != 4 or in_channels != out_channels:
            raise NotImplementedError('the HIP SHU kernels are built for the shipped geometry: 64x64 input, levels 4..64')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.input_res, self.lowest_res = input_res, lowest_res
        self.conv0 = conv2d(in_channels * 2, in_channels * 2, 1, 1, 0)
        self.df1 = heterogeneous_filter(in_channels * 2, out_channels * 2, freedom=dfilter_freedom, type=dfilter_type)
        nn.init.normal_(self.df1.weight, mean=1 / (out_channels * 2), std=0.1 / (out_channels * 2))
        self.tail_sigma_mult = tail_sigma_mult
        self.gaussian_at_input_res = gaussian_at_input_res
        self.reslist = [2 ** i for i in range(int(np.log2(lowest_res)), int(np.log2(input_res)) + 1)]

        # Gaussian-split pyramid, built top-down: each level's raw Gaussian is carved out of the
        # centre block of the level above (shgan.py:284-306); the top level starts from ones.
        maps = {}
        prev = None
        for res in self.reslist[::-1]:
            if prev is None and not gaussian_at_input_res:
                maps[res] = np.ones((res, res // 2 + 1), dtype=np.float32).astype(float)
            else:
                sigma = (res // 2) / tail_sigma_mult
                g = gaussian_heatmap_2d(size=[res, res // 2 + 1])(
                    c=np.array([[res // 2 - 1, 0]], dtype=float), v=np.array([[[sigma ** 2, 0], [0, sigma ** 2]]], dtype=float))
                maps[res] = g
                if prev is not None:
                    maps[prev][prev // 2 - res // 2: prev // 2 + res // 2, 0: res // 2 + 1] -= g
            prev = res
        self.gaussian_weight_map = {r: torch.tensor(maps[r], dtype=torch.float64).float() for r in self.reslist}
        for r in self.reslist:   # resident on the device with the module, not part of the state dict
            self.register_buffer(f'_gauss{r}', self.gaussian_weight_map[r], persistent=False)
        self.register_buffer('_cw', make_cweight(dfilter_freedom, (input_res, input_res // 2 + 1), dfilter_type),
                             persistent=False)

    FUSED_SPECTRAL = True      # conv0 + ReLU + heterogeneous filter + band sum in one launch (False: two 1x1 convolutions)

    def _packed(self):
        """MFMA-ordered copies of the conv0 / heterogeneous-filter weights (per parameter version)."""
        def build():
            w0 = self.conv0.weight.detach()[:, :, 0, 0] * self.conv0.weight_gain                    # [out, in]
            i, ob = self.df1.weight.shape
            w1 = self.df1.weight.detach().t().reshape(ob // self._cw.shape[0], self._cw.shape[0], i).permute(1, 0, 2)   # [B][o][i]
            b0 = self.conv0.bias.detach() * self.conv0.bias_gain if self.conv0.bias is not None else \
                torch.zeros(w0.shape[0], device=w0.device)
            return kernels.mfma_pack_rows(w0), b0.contiguous(), kernels.mfma_pack_rows(w1).reshape(-1, 2, 64)
        return _cache_of(self).get('spec', [q for q in (self.conv0.weight, self.conv0.bias, self.df1.weight) if q is not None], build)

    def _spectral(self, x):
        """-> (y, cw): y [N,2C*B,64,33] with its band table, or the band-summed [N,2C,64,33] and None."""
        t = kernels.shu_rfft2_shift(x)                 # [N,2C,64,33]: Re | Im, DC on row 31
        if self.FUSED_SPECTRAL and t.shape[1] == 64 and self._cw.shape[0] <= 8:
            w0p, b0, w1p = self._packed()
            return kernels.shu_spectral(t, w0p, b0, w1p, self._cw), None
        t = self.conv0(t, relu=True)                   # 1x1 conv + bias + ReLU on the MFMA kernel
        return self.df1.band_conv(t), self._cw         # [N,2C*6,64,33]

    def _forward_train(self, x):
        """Training rows: shgan.py:312-336 on differentiable operators that are all HIP kernels -- the two transform stages through
        ``_ShuSpectrum`` / ``_ShuSplit`` (the inference kernels forward, their transposes backward: both stages are linear maps),
        the two 1x1 convolutions on the HIP conv kernels (forward / backward)."""
        gauss = [getattr(self, f'_gauss{r}') for r in self.reslist]
        t = _ShuSpectrum.apply(x.contiguous())                                   # [N,2C,64,33]: Re | Im, DC on row 31, norm='forward'
        t = self.conv0(t, relu=True)
        t = self.df1(t)                                                          # [N,2C,64,33], bands summed
        outs = _ShuSplit.apply(t.contiguous(), *gauss)
        return dict(zip(self.reslist, outs))

    def forward(self, x):
        if grad_ops.wants_grad(x, *self.parameters()):
            return self._forward_train(x)
        y, cw = self._spectral(x)
        n, c = x.shape[0], self.out_channels
        outs = [torch.empty((n, c, r, r), device=x.device, dtype=torch.float32) for r in self.reslist]
        kernels.shu_split_irfft2(y, cw, [getattr(self, f'_gauss{r}') for r in self.reslist], outs, accumulate=False)
        return dict(zip(self.reslist, outs))

    def forward_accumulate(self, x, feats):
        """Fused form used by the encoder: feats[r][:, -C:] += hint_r, in place."""
        y, cw = self._spectral(x)
        c = self.out_channels
        outs = [feats[r][:, feats[r].shape[1] - c:] for r in self.reslist]
        kernels.shu_split_irfft2(y, cw, [getattr(self, f'_gauss{r}') for r in self.reslist], outs, accumulate=True)
        return feats


@register('shgan_encoder', version)
class Encoder(Encoder_base):
    """Co-modulation encoder + SHU: the last ``shu_channels`` channels of the 64x64 feature feed the SHU
    and its band outputs are added to the same channels of the 4..64 skip features (shgan.py:338-383)."""

    def __init__(self, *args, **kwargs):
        self.shu_input_res = kwargs.pop('shu_input_res')
        self.shu_lowest_res = kwargs.pop('shu_lowest_res')
        self.shu_channels = kwargs.pop('shu_channels')
        self.shu_df_freedom = kwargs.pop('shu_df_freedom')
        self.shu_df_type = kwargs.pop('shu_df_type')
        self.shu_tail_sigma_mult = kwargs.pop('shu_tail_sigma_mult')
        self.shu_gaussian_at_input_res = kwargs.pop('shu_gaussian_at_input_res')
        super().__init__(*args, **kwargs)
        self.shu = SHU(self.shu_channels, self.shu_channels, self.shu_df_freedom, self.shu_df_type,
                       input_res=self.shu_input_res, lowest_res=self.shu_lowest_res,
                       tail_sigma_mult=self.shu_tail_sigma_mult, gaussian_at_input_res=self.shu_gaussian_at_input_res)

    def forward(self, img, c=None):
        x, feats = super().forward(img, c)
        src = feats[self.shu_input_res]
        ch = self.shu_channels
        any_half = any(f.dtype == torch.float16 for f in feats.values())
        if any_half or grad_ops.wants_grad(src, *self.shu.parameters()):
            # training rows (shgan.py:374-382): out-of-place split / add / cat so that autograd sees the hints.  With float16 encoder
            # blocks the SHU itself stays float32 -- its input is cast up (the reference hands a half tensor to torch.fft, which only
            # cuFFT accepts: not reproducible on the CPU reference, stated in DESIGN.md) -- and `fb + v`, `cat` promote as torch does.
            for r, v in self.shu(src[:, src.shape[1] - ch:].to(torch.float32)).items():
                fa, fb = torch.split(feats[r], [feats[r].size(1) - ch, ch], dim=1)
                feats[r] = torch.cat([fa, fb + v], dim=1)
            return x, feats
        self.shu.forward_accumulate(src[:, src.shape[1] - ch:], feats)
        return x, feats
