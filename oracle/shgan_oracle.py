"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the SH-GAN generator forward path.

This file is a CPU restatement (plain PyTorch fp32 CPU ops + numpy) of the
reference algorithm for the hot path of SURVEY.md section 8(a).  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``sh-gan_amd/``) never imports it and has no CPU fallback.

Pinning: the reference ships no tests and no golden vectors,
so this oracle is pinned against outputs of the reference itself, generated in
the build container by ``tools/gen_golden.py`` (which imports /root/reference)
and committed under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those fixtures.

Style: functional, operating on a flat ``state_dict`` (the key schema of
SURVEY.md appendix E) rather than on module objects.  Every function cites the
reference file:line whose semantics it restates (paths relative to
/root/reference/lib/model_zoo unless noted).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)

# ----------------------------------------------------------------------------
# FIR resampling  (stylegan_utils/upfirdn2d.py)
# ----------------------------------------------------------------------------


def setup_filter(taps=(1, 3, 3, 1), normalize=True, flip_filter=False, gain=1.0, separable=None):
    """upfirdn2d.py:66-92 -- 1-D taps with < 8 entries become an outer product."""
    f = torch.as_tensor(taps, dtype=torch.float32).clone()
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(tuple(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def _two(v):
    if isinstance(v, int):
        return v, v
    a, b = v
    return int(a), int(b)


def _four(p):
    if isinstance(p, int):
        return p, p, p, p
    p = [int(v) for v in p]
    if len(p) == 2:
        return p[0], p[0], p[1], p[1]
    return tuple(p)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:98-138 (``_upfirdn2d_ref``): zero-insert, pad/crop, FIR, decimate.

    ``f`` is a 2-D filter ([fh, fw]), a 1-D separable filter, or None (identity).
    The FIR is a true convolution unless ``flip_filter``.
    """
    upx, upy = _two(up)
    dnx, dny = _two(down)
    px0, px1, py0, py1 = _four(padding)
    n, c, h, w = x.shape
    if f is None:
        f = torch.ones(1, 1, dtype=torch.float32)
    z = x.new_zeros(n, c, h * upy, w * upx)
    z[:, :, ::upy, ::upx] = x
    z = F.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(tuple(range(k.ndim)))
    if k.ndim == 2:
        z = F.conv2d(z, k[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        z = F.conv2d(z, k[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        z = F.conv2d(z, k[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return z[:, :, ::dny, ::dnx]


def _fsize(f):
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def filter2d(x, f, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:245-277."""
    px0, px1, py0, py1 = _four(padding)
    fw, fh = _fsize(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:279-314."""
    upx, upy = _two(up)
    px0, px1, py0, py1 = _four(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2,
         py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1.0):
    """upfirdn2d.py:316-351."""
    dnx, dny = _two(down)
    px0, px1, py0, py1 = _four(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2,
         py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


# ----------------------------------------------------------------------------
# conv2d_resample  (stylegan_utils/conv2d_resample.py)
# ----------------------------------------------------------------------------


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d_resample.py:26-51: F.conv2d is a correlation; flip_weight=False flips."""
    if not flip_weight:
        w = w.flip([2, 3])
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """conv2d_resample.py:57-154 (same branch order; the branch decides the arithmetic)."""
    oc, icg, kh, kw = w.shape
    fw, fh = _fsize(f)
    px0, px1, py0, py1 = _four(padding)
    if up > 1:   # :93-97
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:  # :98-102
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2
    pad = [px0, px1, py0, py1]

    if kw == 1 and kh == 1 and down > 1 and up == 1:  # :104-108
        x = upfirdn2d(x, f, down=down, padding=pad, flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:  # :110-114
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d(x, f, up=up, padding=pad, gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:  # :116-120
        x = upfirdn2d(x, f, padding=pad, flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:  # :122-142
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * icg, oc // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                  flip_weight=(not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                      flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:  # :145-147
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d(x, None, padding=pad, flip_filter=flip_filter)  # :150-153 (up == 1 here)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    return x


# ----------------------------------------------------------------------------
# activation / dense / modulated conv  (common/utils.py, stylegan.py)
# ----------------------------------------------------------------------------


def lrelu_agc(x, gain=1.0, alpha=0.2, act_gain=SQRT2, clamp=256.0):
    """common/utils.py:135-143: leaky-relu -> *(act_gain*gain) -> clamp(+-clamp*gain)."""
    x = F.leaky_relu(x, negative_slope=alpha)
    g = act_gain * gain
    if g != 1:
        x = x * g
    if clamp is not None:
        x = x.clamp(-clamp * gain, clamp * gain)
    return x


def fma(a, b, c):
    """stylegan_utils/fma.py:15 -- a*b+c via addcmul."""
    return torch.addcmul(c, a, b)


def dense(x, weight, bias=None, lr_multi=1.0, act=False):
    """stylegan.py:87-98: addmm(b*lr, x, (W*lr/sqrt(in)).T) then optional lrelu_agc."""
    w = weight * (lr_multi / math.sqrt(weight.shape[1]))
    if bias is not None:
        b = bias * lr_multi if lr_multi != 1 else bias
        y = torch.addmm(b.unsqueeze(0), x, w.t())
    else:
        y = x @ w.t()
    return lrelu_agc(y) if act else y


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True):
    """stylegan.py:103-193 (fp32 branches only; the fp16 pre-normalisation :136-138 is unused)."""
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    w = None
    dcoefs = None
    if demodulate:  # :145-147
        weight = weight * weight.square().mean([1, 2, 3], keepdim=True).rsqrt()
        styles = styles * styles.square().mean().rsqrt()
    if demodulate or fused_modconv:  # :149-151
        w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:  # :155
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if demodulate and fused_modconv:  # :168-169
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    if not fused_modconv:  # :172-181
        x = x * styles.reshape(n, -1, 1, 1)
        x = conv2d_resample(x, weight, f=resample_filter, up=up, down=down, padding=padding,
                            flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma(x, dcoefs.reshape(n, -1, 1, 1), noise)
        elif demodulate:
            x = x * dcoefs.reshape(n, -1, 1, 1)
        elif noise is not None:
            x = x + noise
        return x
    x = x.reshape(1, -1, *x.shape[2:])  # :187-192
    w = w.reshape(-1, ic, kh, kw)
    x = conv2d_resample(x, w, f=resample_filter, up=up, down=down, padding=padding, groups=n,
                        flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


# ----------------------------------------------------------------------------
# layers on a flat state dict
# ----------------------------------------------------------------------------


def conv2d_layer(sd, p, x, k, up=1, down=1, use_filter=False, act=True, gain=1.0):
    """stylegan.py:226-238 (conv2d_layer.forward).  ``p`` = key prefix incl. trailing dot."""
    w = sd[p + 'weight']
    w = w * (1.0 / math.sqrt(w.shape[1] * k * k))
    f = sd[p + 'resample_filter'] if use_filter else None
    x = conv2d_resample(x, w, f=f, up=up, down=down, padding=k // 2, flip_weight=(up == 1))
    if (p + 'bias') in sd:
        x = x + sd[p + 'bias'].view(1, -1, 1, 1)
    return lrelu_agc(x, gain=gain) if act else x * gain


def _noise(sd, p, n, res, noise_mode, noise_in):
    """stylegan.py:281-285.  'random' draws are supplied by the caller via ``noise_in``
    (dict prefix -> [N,1,r,r] standard normal) so that runs are reproducible."""
    if noise_mode == 'none':
        return None
    if noise_mode == 'const':
        return sd[p + 'noise_const'] * sd[p + 'noise_strength']
    if noise_in is not None and p in noise_in:
        z = noise_in[p]
    else:
        z = torch.randn(n, 1, res, res)
    return z * sd[p + 'noise_strength']


def synthesis_layer(sd, p, x, wlong, res, up=1, gain=1.0, noise_mode='const', noise_in=None,
                    fused_modconv=True):
    """stylegan.py:276-304."""
    styles = dense(wlong, sd[p + 'affine.weight'], sd[p + 'affine.bias'])
    noise = _noise(sd, p, x.shape[0], res, noise_mode, noise_in)
    f = sd[p + 'resample_filter'] if up > 1 else None
    x = modulated_conv2d(x, sd[p + 'weight'], styles, noise=noise, up=up, padding=1,
                         resample_filter=f, flip_weight=(up == 1), fused_modconv=fused_modconv)
    x = x + sd[p + 'bias'].view(1, -1, 1, 1)
    return lrelu_agc(x, gain=gain)


def torgb_layer(sd, p, x, wlong, fused_modconv=True):
    """stylegan.py:325-337: styles scaled by 1/sqrt(Cin), no demodulation, no activation."""
    w = sd[p + 'weight']
    styles = dense(wlong, sd[p + 'affine.weight'], sd[p + 'affine.bias']) * (1.0 / math.sqrt(w.shape[1]))
    x = modulated_conv2d(x, w, styles, demodulate=False, fused_modconv=fused_modconv)
    return x + sd[p + 'bias'].view(1, -1, 1, 1)


def mapping(sd, z, num_ws, num_layers=8, lr_multi=0.01, p='mapping.'):
    """stylegan.py:394-430 with truncation_psi == 1, eval mode, c_dim == 0."""
    x = z.to(torch.float32)
    x = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()   # :343-344
    for i in range(num_layers):
        x = dense(x, sd[f'{p}fc{i}.weight'], sd[f'{p}fc{i}.bias'], lr_multi=lr_multi, act=True)
    return x.unsqueeze(1).repeat(1, num_ws, 1)


# ----------------------------------------------------------------------------
# Spectral Hint Unit  (shgan.py)
# ----------------------------------------------------------------------------


def make_cweight_closed_form(hs=64, ws=33):
    """shgan.py:70-121 for freedom [2,3], 'piecewise_linear' -- closed form of the
    bilinear grid_sample of the reflect-padded one-hot grid:
    cw[3a+b,h,w] = rho_a(h) * kappa_b(w)."""
    h = torch.arange(hs, dtype=torch.float64)
    w = torch.arange(ws, dtype=torch.float64)
    t = (h + 1) / hs
    rho = torch.stack([1 - t, t])                       # [2,hs]
    u = w / ((ws - 1) / 2)
    kap = torch.stack([(1 - u).clamp(0, 1), (1 - (u - 1).abs()).clamp(0, 1), (u - 1).clamp(0, 1)])
    cw = (rho[:, None, :, None] * kap[None, :, None, :]).reshape(6, hs, ws)
    return cw.to(torch.float32)


def make_cweight_grid_sample(freedom=(2, 3), half_sample=(64, 33)):
    """shgan.py:70-121 restated literally (one-hot grid -> reflect pad -> grid_sample)."""
    h0, w0 = freedom
    hs, ws = half_sample
    oh = torch.eye(h0 * w0, dtype=torch.float32).reshape(1, h0 * w0, h0, w0)
    oh = F.pad(oh, pad=(w0 - 1, 0, 0, 0), mode='reflect')
    if hs % 2 == 0:
        hg = np.array([-1 + i / hs * 2 for i in range(hs + 1)])[1:]
    else:
        hg = np.array([-1 + i / (hs - 1) * 2 for i in range(hs)])
    wg = np.array([i / (ws - 1) for i in range(ws)])
    wg, hg = np.meshgrid(wg, hg)
    grid = torch.tensor(np.stack([wg, hg], axis=-1), dtype=torch.float32).unsqueeze(0)
    cw = F.grid_sample(oh, grid, mode='bilinear', padding_mode='border', align_corners=True)
    return cw.squeeze(0)


def gaussian_split_tables(input_res=64, lowest_res=4, tail_sigma_mult=3.0):
    """shgan.py:281-310 (+ gaussian_heatmap_2d :162-250): per-resolution spectral weights.

    g_r(h,w) = exp(-((h-(r/2-1))^2 + w^2) / (2 sigma^2)), sigma=(r/2)/mult, on [r, r/2+1];
    the 'speedup' window of :206-229 is applied exactly as in the reference.  Then, walking
    resolutions downward, each smaller raw g_r is subtracted from the centre block of the
    next larger table; the top table starts as ones."""
    res_desc = []
    r = input_res
    while r >= lowest_res:
        res_desc.append(r)
        r //= 2
    tabs = {}
    for idx, r in enumerate(res_desc):
        if idx == 0:
            tabs[r] = np.ones((r, r // 2 + 1), dtype=np.float32).astype(np.float64)
            continue
        hh, ww = r, r // 2 + 1
        ch, cw_ = float(r // 2 - 1), 0.0
        sigma = (r // 2) / tail_sigma_mult
        g = np.zeros((hh, ww), dtype=np.float64)
        sr = int(3 * sigma + 1)
        h0, h1 = [max(min(v, hh), 0) for v in (int(ch) - sr, int(ch) + sr)]
        w0, w1 = [max(min(v, ww), 0) for v in (int(cw_) - sr, int(cw_) + sr)]
        if h1 > h0 and w1 > w0:
            yy = np.arange(h0, h1)[:, None] - ch
            xx = np.arange(w0, w1)[None, :] - cw_
            val = np.exp(-0.5 * (yy * yy + xx * xx) / (sigma * sigma))
            g[h0:h1, w0:w1] = np.maximum(g[h0:h1, w0:w1], val)
        tabs[r] = g
        rp = res_desc[idx - 1]
        tabs[rp][rp // 2 - r // 2: rp // 2 + r // 2, 0: r // 2 + 1] -= g
    return {r: torch.tensor(t, dtype=torch.float64).to(torch.float32) for r, t in tabs.items()}


def heterogeneous_filter(x, df_weight, cw):
    """shgan.py:143-160: 1x1 conv C->C*6 then band-weighted sum; flat out index = o*6+k."""
    n, c, h, w = x.shape
    y = F.conv2d(x, df_weight.t()[:, :, None, None]).view(n, c, -1, h, w)
    return (y * cw[None, None]).sum(2)


def shu_forward(sd, x, p='encoder.shu.', input_res=64, lowest_res=4, tail_sigma_mult=3.0):
    """shgan.py:312-336.  x = [N,32,64,64] -> {r: [N,32,r,r]}."""
    ch = x.shape[1]
    sp = torch.fft.rfftn(x, dim=(2, 3), norm='forward')
    hh = sp.shape[2]
    sp = torch.cat([sp[:, :, hh // 2 + 1:], sp[:, :, :hh // 2 + 1]], dim=2)      # :315-317
    t = torch.cat([sp.real, sp.imag], dim=1)
    t = F.relu(F.conv2d(t, sd[p + 'conv0.weight'], sd[p + 'conv0.bias']))        # :320-321
    cw = make_cweight_closed_form(t.shape[2], t.shape[3])
    t = heterogeneous_filter(t, sd[p + 'df1.weight'], cw)
    sp = torch.complex(t[:, :ch], t[:, ch:])
    tabs = gaussian_split_tables(input_res, lowest_res, tail_sigma_mult)
    out = {}
    r = lowest_res
    while r <= input_res:
        s = sp[:, :, input_res // 2 - r // 2: input_res // 2 + r // 2, 0: r // 2 + 1].clone()
        s = s * tabs[r][None, None]
        k = r - r // 2 - 1
        s = torch.cat([s[:, :, k:], s[:, :, :k]], dim=2)                          # :331-333
        out[r] = torch.fft.irfftn(s, dim=(2, 3), norm='forward')
        r *= 2
    return out


# ----------------------------------------------------------------------------
# Encoder / Synthesis / Generator  (comodgan.py, shgan.py)
# ----------------------------------------------------------------------------


def _ch(res, ch_base=32768, ch_max=512):
    return min(ch_base // res, ch_max)


def encoder(sd, img, resolution, p='encoder.', shu=True, shu_channels=32, shu_input_res=64):
    """shgan.py:361-383 / comodgan.py:191-205,38-64,98-113 (reslink off, mbstd off,
    dropout identity in eval, no extra final layer)."""
    feats = {}
    x = None
    r = resolution
    first = True
    while r > 4:
        b = f'{p}b{r}.'
        if first:
            x = conv2d_layer(sd, b + 'fromrgb.', img, k=1)
            first = False
        feat = conv2d_layer(sd, b + 'conv0.', x, k=3)
        x = conv2d_layer(sd, b + 'conv1.', feat, k=3, down=2, use_filter=True)
        feats[r] = feat
        r //= 2
    feat = conv2d_layer(sd, p + 'b4.conv.', x, k=3)
    xg = dense(feat.flatten(1), sd[p + 'b4.fc.weight'], sd[p + 'b4.fc.bias'], act=True)
    feats[4] = feat
    if shu:
        hints = shu_forward(sd, feats[shu_input_res][:, -shu_channels:], p=p + 'shu.',
                            input_res=shu_input_res)
        for r, v in hints.items():
            fa, fb = torch.split(feats[r], [feats[r].shape[1] - shu_channels, shu_channels], dim=1)
            feats[r] = torch.cat([fa, fb + v], dim=1)
    return xg, feats


def synthesis(sd, xg, feats, ws, resolution, noise_mode='const', noise_in=None, p='synthesis.',
              fused_modconv=True):
    """comodgan.py:396-433, 237-262, 304-340."""
    ws = ws.to(torch.float32)
    n = xg.shape[0]
    b = p + 'b4.'
    x = dense(xg, sd[b + 'fc.weight'], sd[b + 'fc.bias'], act=True).view(n, -1, 4, 4) + feats[4]
    x = synthesis_layer(sd, b + 'conv.', x, torch.cat([ws[:, 0], xg], 1), 4, noise_mode=noise_mode,
                        noise_in=noise_in, fused_modconv=fused_modconv)
    img = torgb_layer(sd, b + 'torgb.', x, torch.cat([ws[:, 1], xg], 1), fused_modconv=True)
    widx = 1
    r = 8
    while r <= resolution:
        b = f'{p}b{r}.'
        x = synthesis_layer(sd, b + 'conv0.', x, torch.cat([ws[:, widx], xg], 1), r, up=2,
                            noise_mode=noise_mode, noise_in=noise_in, fused_modconv=fused_modconv)
        x = x + feats[r]
        x = synthesis_layer(sd, b + 'conv1.', x, torch.cat([ws[:, widx + 1], xg], 1), r,
                            noise_mode=noise_mode, noise_in=noise_in, fused_modconv=fused_modconv)
        img = upsample2d(img, sd[b + 'resample_filter'])
        img = img + torgb_layer(sd, b + 'torgb.', x, torch.cat([ws[:, widx + 2], xg], 1),
                                fused_modconv=fused_modconv)
        widx += 2
        r *= 2
    return img


NUM_WS = {256: 14, 512: 16, 1024: 18}   # comodgan.py:367-372


def generator_forward(sd, x, z, resolution, noise_mode='const', noise_in=None, shu=True,
                      return_intermediates=False, fused_modconv=True):
    """comodgan.py:449-481 (c_dim == 0, truncation_psi == 1)."""
    num_ws = NUM_WS.get(resolution, 2 * int(math.log2(resolution)) - 2)
    ws = mapping(sd, z, num_ws)
    xg, feats = encoder(sd, x, resolution, shu=shu)
    img = synthesis(sd, xg, feats, ws, resolution, noise_mode=noise_mode, noise_in=noise_in,
                    fused_modconv=fused_modconv)
    if return_intermediates:
        return img, dict(ws=ws, xg=xg, feats=feats)
    return img


def run_generator(sd, x, z, resolution, noise_mode='const', noise_in=None):
    """lib/experiments/shgan_default.py:257-262: composite + uint8 by truncation."""
    m = x[:, 0:1] + 0.5
    img = generator_forward(sd, x, z, resolution, noise_mode=noise_mode, noise_in=noise_in)
    comb = x[:, 1:4] * m + img * (1 - m)
    return (comb * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)


def composite_u8(x, img):
    """The arithmetic tail of run_generator, given an already generated image."""
    m = x[:, 0:1] + 0.5
    comb = x[:, 1:4] * m + img * (1 - m)
    return (comb * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------
# random-init state dict of the reference's initialisers
# ----------------------------------------------------------------------------


def seeded_fill_(module, seed, bias_std=0.1):
    """Fill every parameter of ``module`` (the reference's or this package's -- same key schema) in ``named_parameters()`` order from
    numpy's legacy ``RandomState(seed)``: weights ~ N(0,1) as the reference initialises them (stylegan.py:80,219), biases ~
    N(0, bias_std) instead of 0 so the bias paths carry signal.  Lets a fixture for a 29 M-parameter discriminator record a seed
    instead of the weights (tests / tools only)."""
    g = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = torch.from_numpy(g.standard_normal(tuple(p.shape)).astype(np.float32))
            p.copy_(v * bias_std if name.endswith('bias') else v)
    return module


def init_state_dict(resolution, seed=0, ch_base=32768, ch_max=512, w_dim=512, z_dim=512,
                    w0_dim=1024, shu_channels=32, noise_strength=0.0, bias_std=0.0):
    """Random-init weights with the reference's initialisers and key schema (appendix E).

    conv/dense weights ~ N(0,1) (stylegan.py:80,219; mapping weights divided by lr_multi),
    biases 0 (affine bias 1, :266), noise_const ~ N(0,1), noise_strength 0 (:270-271),
    SHU conv0 ~ N(0, 1/sqrt(64)) (:40-49), df1 ~ N(1/64, 0.1/64) (shgan.py:275).
    ``noise_strength`` / ``bias_std`` optionally perturb the zero-initialised entries so
    parity tests exercise those data paths.  Draws come from numpy's legacy
    ``RandomState(seed)`` (bit-stable across machines and numpy versions), in this
    function's own order -- fixtures therefore only need to record the seed; the golden
    generator loads this dict into the reference modules with ``strict=True``."""
    g = np.random.RandomState(seed)

    def rn(*s):
        return torch.from_numpy(g.standard_normal(s).astype(np.float32))

    def bias(nn_):
        return rn(nn_) * bias_std if bias_std else torch.zeros(nn_)

    sd = {}
    sd['mapping.w_avg'] = torch.zeros(w_dim)
    for i in range(8):
        sd[f'mapping.fc{i}.weight'] = rn(w_dim, z_dim if i == 0 else w_dim) / 0.01
        sd[f'mapping.fc{i}.bias'] = bias(w_dim)
    filt = setup_filter([1, 3, 3, 1])
    r = resolution
    first = True
    while r > 4:
        c, c2 = _ch(r, ch_base, ch_max), _ch(r // 2, ch_base, ch_max)
        b = f'encoder.b{r}.'
        sd[b + 'resample_filter'] = filt.clone()
        if first:
            sd[b + 'fromrgb.weight'] = rn(c, 4, 1, 1)
            sd[b + 'fromrgb.bias'] = bias(c)
            first = False
        sd[b + 'conv0.weight'] = rn(c, c, 3, 3)
        sd[b + 'conv0.bias'] = bias(c)
        sd[b + 'conv1.weight'] = rn(c2, c, 3, 3)
        sd[b + 'conv1.bias'] = bias(c2)
        sd[b + 'conv1.resample_filter'] = filt.clone()
        r //= 2
    c4 = _ch(4, ch_base, ch_max)
    sd['encoder.b4.conv.weight'] = rn(c4, c4, 3, 3)
    sd['encoder.b4.conv.bias'] = bias(c4)
    sd['encoder.b4.fc.weight'] = rn(w0_dim, c4 * 16)
    sd['encoder.b4.fc.bias'] = bias(w0_dim)
    sc = 2 * shu_channels
    sd['encoder.shu.conv0.weight'] = rn(sc, sc, 1, 1) / math.sqrt(sc)
    sd['encoder.shu.conv0.bias'] = bias(sc)
    sd['encoder.shu.df1.weight'] = 1.0 / sc + rn(sc, sc * 6) * (0.1 / sc)
    wl = w_dim + w0_dim
    sd['synthesis.b4.fc.weight'] = rn(c4 * 16, w0_dim)
    sd['synthesis.b4.fc.bias'] = bias(c4 * 16)

    def syn_layer(pfx, ci, co, res, up):
        sd[pfx + 'weight'] = rn(co, ci, 3, 3)
        sd[pfx + 'bias'] = bias(co)
        sd[pfx + 'noise_strength'] = torch.tensor(float(noise_strength))
        if up:
            sd[pfx + 'resample_filter'] = filt.clone()
        sd[pfx + 'noise_const'] = rn(res, res)
        sd[pfx + 'affine.weight'] = rn(ci, wl)
        sd[pfx + 'affine.bias'] = torch.ones(ci)

    def rgb_layer(pfx, ci):
        sd[pfx + 'weight'] = rn(3, ci, 1, 1)
        sd[pfx + 'bias'] = bias(3)
        sd[pfx + 'affine.weight'] = rn(ci, wl)
        sd[pfx + 'affine.bias'] = torch.ones(ci)

    syn_layer('synthesis.b4.conv.', c4, c4, 4, True)   # b4.conv carries a resample_filter buffer
    rgb_layer('synthesis.b4.torgb.', c4)
    r = 8
    while r <= resolution:
        ci, co = _ch(r // 2, ch_base, ch_max), _ch(r, ch_base, ch_max)
        b = f'synthesis.b{r}.'
        sd[b + 'resample_filter'] = filt.clone()
        syn_layer(b + 'conv0.', ci, co, r, True)
        syn_layer(b + 'conv1.', co, co, r, False)
        rgb_layer(b + 'torgb.', co)
        r *= 2
    return sd


# ----------------------------------------------------------------------------
# integer paths: masks, sampler, re-interleave
# ----------------------------------------------------------------------------


def random_brush(max_tries, s, min_num_vertex=4, max_num_vertex=18, mean_angle=2 * math.pi / 5,
                 angle_range=2 * math.pi / 15, min_width=12, max_width=48):
    """lib/data_factory/ds_ffhq.py:145-197 (numpy global RNG call order preserved; the two
    ``mask.transpose`` calls whose results the reference discards still consume RNG draws)."""
    from PIL import Image, ImageDraw
    avg_r = math.sqrt(2 * s * s) / 8
    canvas = Image.new('L', (s, s), 0)
    for _ in range(np.random.randint(max_tries)):
        nv = np.random.randint(min_num_vertex, max_num_vertex)
        amin = mean_angle - np.random.uniform(0, angle_range)
        amax = mean_angle + np.random.uniform(0, angle_range)
        angles = []
        for i in range(nv):
            a = np.random.uniform(amin, amax)
            angles.append(2 * math.pi - a if i % 2 == 0 else a)
        hh, ww = canvas.size
        pts = [(int(np.random.randint(0, ww)), int(np.random.randint(0, hh)))]
        for i in range(nv):
            rad = np.clip(np.random.normal(loc=avg_r, scale=avg_r // 2), 0, 2 * avg_r)
            nx = np.clip(pts[-1][0] + rad * math.cos(angles[i]), 0, ww)
            ny = np.clip(pts[-1][1] + rad * math.sin(angles[i]), 0, hh)
            pts.append((int(nx), int(ny)))
        pen = ImageDraw.Draw(canvas)
        width = int(np.random.uniform(min_width, max_width))
        pen.line(pts, fill=1, width=width)
        for v in pts:
            pen.ellipse((v[0] - width // 2, v[1] - width // 2, v[0] + width // 2, v[1] + width // 2), fill=1)
        np.random.random()   # flip decision whose result is discarded (:188-189)
        np.random.random()   # (:190-191)
    arr = np.asarray(canvas, np.uint8)
    if np.random.random() > 0.5:
        arr = np.flip(arr, 0)
    if np.random.random() > 0.5:
        arr = np.flip(arr, 1)
    return arr


def random_mask(s, hole_range=(0, 1)):
    """lib/data_factory/ds_ffhq.py:199-217 -> float32 [1,s,s] in {0,1}."""
    coef = min(hole_range[0] + hole_range[1], 1.0)
    while True:
        m = np.ones((s, s), np.uint8)

        def fill(max_size):
            w, h = np.random.randint(max_size), np.random.randint(max_size)
            ww, hh = w // 2, h // 2
            x, y = np.random.randint(-ww, s - w + ww), np.random.randint(-hh, s - h + hh)
            m[max(y, 0): min(y + h, s), max(x, 0): min(x + w, s)] = 0

        for _ in range(np.random.randint(int(10 * coef))):
            fill(s // 2)
        for _ in range(np.random.randint(int(5 * coef))):
            fill(s)
        m = np.logical_and(m, 1 - random_brush(int(20 * coef), s))
        ratio = 1 - np.mean(m)
        if hole_range is not None and (ratio <= hole_range[0] or ratio >= hole_range[1]):
            continue
        return m[np.newaxis, ...].astype(np.float32)


def sampler_indices(n_items, world, rank, extend=True):
    """lib/data_factory/common/ds_sampler.py:43-68 with shuffle=False."""
    per = n_items // world
    if extend and n_items != per * world:
        per += 1
    total = per * world
    idx = list(range(n_items))
    if extend:
        idx = idx + idx[0: total - len(idx)]
    else:
        idx = idx[0:total]
    return idx[rank: len(idx): world]


def zipzap_arrange(per_rank):
    """lib/evaluator/eva_base.py:196-212 (list branch): [[0,2,4],[1,3,5]] -> [0,1,2,3,4,5]."""
    out = []
    maxlen = max(len(v) for v in per_rank)
    total = sum(len(v) for v in per_rank)
    for i in range(maxlen):
        for v in per_rank:
            if i < len(v) and len(out) < total:
                out.append(v[i])
    return out


def synthetic_batch(n, resolution, z_dim=512, seed=0):
    """Synthetic masked inputs of SURVEY.md 8(d) / shgan_default.py:267-276:
    real ~ U{0..255}/127.5-1, freeform masks from ``random_mask`` (numpy global RNG seeded
    with ``seed``), x = cat([mask-0.5, real*mask]), z ~ N(0,1).  Same recipe (and RNG
    order) as ``tools/gen_golden.py:synth_inputs`` which uses the reference's RandomMask."""
    g = np.random.RandomState(seed)
    real_u8 = g.randint(0, 256, size=(n, 3, resolution, resolution)).astype(np.uint8)
    np.random.seed(seed)
    mask = np.stack([random_mask(resolution, [0, 1]) for _ in range(n)]).astype(np.uint8)
    z = torch.from_numpy(g.standard_normal((n, z_dim)).astype(np.float32))
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    m = torch.from_numpy(mask.astype(np.float32))
    x = torch.cat([m - 0.5, real * m], dim=1)
    return x, z, real_u8, mask
