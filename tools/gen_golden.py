#!/usr/bin/env python3
"""
Golden-vector generator: runs the *reference itself* (imported from /root/reference,
CPU PyTorch path) and writes small input/output fixtures to tests/golden/*.npz.

Runs ONLY in the build container (where /root/reference exists); the fixtures it writes
are data (inputs + the reference's outputs) and are committed.  Nothing here is imported
by the product or by tests at run time.

Weights come from ``oracle.shgan_oracle.init_state_dict(seed)`` (numpy RandomState, bit
stable) and are loaded into the reference modules with ``load_state_dict(strict=True)``,
so fixtures record only the seed -- and the strict load pins the state_dict key schema
(SURVEY.md appendix E).

Usage:  python tools/gen_golden.py [--only NAME ...]
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('SHGAN_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')

# --- import shim: stub modules the reference imports but never uses here
for _name in ['torchvision', 'torchvision.models', 'torchvision.transforms', 'pyspng', 'cv2']:
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from lib.model_zoo import comodgan, shgan, stylegan  # noqa: E402,F401
from lib.model_zoo.common import utils as ref_utils  # noqa: E402
from lib.model_zoo.stylegan_utils import conv2d_resample as ref_c2r  # noqa: E402
from lib.model_zoo.stylegan_utils import upfirdn2d as ref_ufd  # noqa: E402
from oracle import shgan_oracle as orc  # noqa: E402  (only for init_state_dict / input synthesis)

ACT = 'lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)'


def rs(seed):
    return np.random.RandomState(seed)


def f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def tn(a):
    return torch.from_numpy(f32(a))


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)')


# ---------------------------------------------------------------------------------------------


def gen_upfirdn2d():
    """A10: the three generator configurations + the edge cases the CUDA op accepts."""
    g = rs(100)
    f4 = ref_ufd.setup_filter([1, 3, 3, 1])
    fasym = tn(g.standard_normal((3, 5)))       # asymmetric, non-square: catches flips/transposes
    f1d = ref_ufd.setup_filter([1, 2, 3, 4, 5, 4, 3, 2, 1])   # >= 8 taps -> separable 1-D
    cases = [
        # name, shape, filter, up, down, padding, flip, gain
        ('enc_down_prefilter', (2, 3, 16, 16), f4, 1, 1, [2, 2, 2, 2], False, 1.0),
        ('syn_up_postfilter', (1, 6, 17, 17), f4, 1, 1, [1, 1, 1, 1], False, 4.0),
        ('rgb_upsample', (2, 3, 8, 8), f4, 2, 1, [2, 1, 2, 1], False, 4.0),
        ('down2', (2, 2, 16, 16), f4, 1, 2, [1, 1, 1, 1], False, 1.0),
        ('asym_flip', (1, 2, 9, 11), fasym, 1, 1, [2, 1, 0, 3], True, 0.5),
        ('asym_noflip', (1, 2, 9, 11), fasym, 1, 1, [2, 1, 0, 3], False, 0.5),
        ('asym_up2_down3', (1, 2, 7, 6), fasym, [2, 3], [3, 2], [3, 2, 4, 1], False, 2.0),
        ('negative_pad_crop', (1, 2, 12, 12), f4, 1, 1, [-1, 0, -2, 1], False, 1.0),
        ('separable_1d', (1, 2, 20, 20), f1d, 2, 1, [5, 4, 5, 4], False, 4.0),
        ('identity_none', (1, 2, 5, 5), None, 1, 1, [0, 0, 0, 0], False, 1.0),
        ('odd_513', (1, 1, 33, 33), f4, 1, 1, [1, 1, 1, 1], False, 4.0),
    ]
    out = {}
    names = []
    for name, shape, f, up, down, pad, flip, gain in cases:
        x = tn(g.standard_normal(shape))
        y = ref_ufd.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        names.append(name)
        out[name + '__x'] = x.numpy()
        out[name + '__f'] = f.numpy() if f is not None else np.zeros((0,), np.float32)
        up2 = [up, up] if isinstance(up, int) else list(up)          # [x, y]
        dn2 = [down, down] if isinstance(down, int) else list(down)
        out[name + '__cfg'] = np.array(up2 + dn2 + list(pad) + [int(flip)], dtype=np.int64)
        out[name + '__gain'] = np.float64(gain)
        out[name + '__y'] = y.numpy()
    # helpers: upsample2d / downsample2d / filter2d / setup_filter
    x = tn(g.standard_normal((2, 3, 10, 10)))
    out['helpers__x'] = x.numpy()
    out['helpers__f'] = f4.numpy()
    out['helpers__up'] = ref_ufd.upsample2d(x, f4).numpy()
    out['helpers__down'] = ref_ufd.downsample2d(x, f4).numpy()
    out['helpers__filt'] = ref_ufd.filter2d(x, f4).numpy()
    out['setup_filter_1331'] = f4.numpy()
    out['setup_filter_sep'] = f1d.numpy()
    out['setup_filter_gain_flip'] = ref_ufd.setup_filter([[1, 2], [3, 4]], flip_filter=True, gain=4).numpy()
    out['names'] = np.array(names)
    save('upfirdn2d', **out)


def gen_conv2d_resample():
    """A8/A9: every branch the generator (and D's skip path) can take, groups 1 and N."""
    g = rs(200)
    f4 = ref_ufd.setup_filter([1, 3, 3, 1])
    cases = [
        # name, x shape, w shape, up, down, padding, groups, flip_weight
        ('plain_3x3', (2, 5, 12, 12), (7, 5, 3, 3), 1, 1, 1, 1, True),
        ('plain_3x3_noflip', (2, 5, 12, 12), (7, 5, 3, 3), 1, 1, 1, 1, False),
        ('plain_1x1', (2, 4, 9, 9), (6, 4, 1, 1), 1, 1, 0, 1, True),
        ('down2_3x3', (2, 5, 16, 16), (6, 5, 3, 3), 1, 2, 1, 1, True),
        ('up2_3x3', (2, 5, 8, 8), (6, 5, 3, 3), 2, 1, 1, 1, False),
        ('up2_3x3_grouped', (1, 10, 8, 8), (12, 5, 3, 3), 2, 1, 1, 2, False),
        ('plain_3x3_grouped', (1, 10, 8, 8), (12, 5, 3, 3), 1, 1, 1, 2, True),
        ('down2_1x1', (2, 4, 16, 16), (6, 4, 1, 1), 1, 2, 0, 1, True),
        ('up2_1x1', (2, 4, 8, 8), (6, 4, 1, 1), 2, 1, 0, 1, True),
        ('up2_4x4_small', (1, 3, 4, 4), (2, 3, 3, 3), 2, 1, 1, 1, False),
    ]
    out = {}
    names = []
    for name, xs, ws, up, down, pad, groups, flipw in cases:
        x = tn(g.standard_normal(xs))
        w = tn(g.standard_normal(ws))
        f = f4 if (up > 1 or down > 1) else None
        y = ref_c2r.conv2d_resample(x=x, w=w, f=f, up=up, down=down, padding=pad, groups=groups,
                                    flip_weight=flipw)
        names.append(name)
        out[name + '__x'] = x.numpy()
        out[name + '__w'] = w.numpy()
        out[name + '__cfg'] = np.array([up, down, pad, groups, int(flipw), int(f is not None)], dtype=np.int64)
        out[name + '__y'] = y.numpy()
    out['f'] = f4.numpy()
    out['names'] = np.array(names)
    save('conv2d_resample', **out)


def gen_modconv():
    """A4: fused / non-fused x up{1,2} x demod{T,F}, with and without noise; 1x1 torgb form."""
    g = rs(300)
    f4 = ref_ufd.setup_filter([1, 3, 3, 1])
    out = {}
    names = []
    for up in (1, 2):
        for demod in (True, False):
            for fused in (True, False):
                for k in (3, 1):
                    if k == 1 and up == 2:
                        continue
                    name = f'up{up}_demod{int(demod)}_fused{int(fused)}_k{k}'
                    n, ci, co, r = 3, 6, 5, 8
                    x = tn(g.standard_normal((n, ci, r, r)))
                    w = tn(g.standard_normal((co, ci, k, k)))
                    s = tn(g.standard_normal((n, ci)) + 1.0)
                    ro = r * up
                    noise = tn(g.standard_normal((n, 1, ro, ro)) * 0.3)
                    y = stylegan.modulated_conv2d(
                        x=x.clone(), weight=w, styles=s, noise=noise.clone(), up=up, padding=k // 2,
                        resample_filter=(f4 if up > 1 else None), demodulate=demod,
                        flip_weight=(up == 1), fused_modconv=fused)
                    names.append(name)
                    out[name + '__x'] = x.numpy()
                    out[name + '__w'] = w.numpy()
                    out[name + '__s'] = s.numpy()
                    out[name + '__noise'] = noise.numpy()
                    out[name + '__cfg'] = np.array([up, int(demod), int(fused), k], dtype=np.int64)
                    out[name + '__y'] = y.numpy()
    out['f'] = f4.numpy()
    out['names'] = np.array(names)
    save('modulated_conv2d', **out)


def gen_small_ops():
    """A3 dense, A11 lrelu_agc, A12 get_unit, fma."""
    g = rs(400)
    out = {}
    x = tn(np.concatenate([g.standard_normal(2000) * 3, [0.0, -0.0, 1e30, -1e30, 200.0, -1500.0]]))
    act = ref_utils.get_unit()(ACT)()
    out['lrelu__x'] = x.numpy()
    out['lrelu__y_gain1'] = act(x.clone(), gain=1).numpy()
    out['lrelu__y_gain_sqrt_half'] = act(x.clone(), gain=np.sqrt(0.5)).numpy()
    act2 = ref_utils.get_unit()('lrelu_agc(alpha=0.1, gain=1)')()
    out['lrelu__y_noclamp'] = act2(x.clone()).numpy()
    for tag, (n, i, o, lr, bias_init, use_act) in {
            'mapping': (4, 64, 48, 0.01, 0.0, True),
            'affine': (3, 96, 40, 1.0, 1.0, False),
            'fc': (2, 512, 33, 1.0, 0.0, True)}.items():
        torch.manual_seed(7)
        m = stylegan.dense(i, o, bias=True, bias_init=bias_init, activation=(ACT if use_act else None), lr_multi=lr)
        with torch.no_grad():
            m.weight.copy_(tn(g.standard_normal((o, i))) / lr)
            m.bias.copy_(tn(g.standard_normal(o)) + bias_init)
        xx = tn(g.standard_normal((n, i)))
        with torch.no_grad():
            yy = m(xx)
        out[f'dense_{tag}__x'] = xx.numpy()
        out[f'dense_{tag}__w'] = m.weight.detach().numpy()
        out[f'dense_{tag}__b'] = m.bias.detach().numpy()
        out[f'dense_{tag}__cfg'] = np.array([lr, float(use_act)], dtype=np.float64)
        out[f'dense_{tag}__y'] = yy.numpy()
    from lib.model_zoo.stylegan_utils import fma as ref_fma
    a, b, c = (tn(g.standard_normal((2, 3, 4, 4))), tn(g.standard_normal((2, 3, 1, 1))),
               tn(g.standard_normal((2, 1, 4, 4))))
    out['fma__a'], out['fma__b'], out['fma__c'] = a.numpy(), b.numpy(), c.numpy()
    out['fma__y'] = ref_fma.fma(a, b, c).numpy()
    save('small_ops', **out)


def gen_shu():
    """A16-A19: constant tables + SHU end to end (N=2) + heterogeneous filter alone."""
    g = rs(500)
    out = {}
    out['cweight_2x3_64x33'] = shgan.make_cweight([2, 3], (64, 33)).numpy()
    torch.manual_seed(0)
    shu = shgan.SHU(32, 32, [2, 3], 'piecewise_linear', input_res=64, lowest_res=4,
                    tail_sigma_mult=3, gaussian_at_input_res=False).eval()
    for r, t in shu.gaussian_weight_map.items():
        out[f'gauss_{r}'] = t.numpy()
    sd = orc.init_state_dict(256, seed=11, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128,
                             bias_std=0.2)
    sub = {k[len('encoder.shu.'):]: v for k, v in sd.items() if k.startswith('encoder.shu.')}
    shu.load_state_dict(sub, strict=True)
    x = tn(g.standard_normal((2, 32, 64, 64)))
    with torch.no_grad():
        y = shu(x)
    out['shu__x'] = x.numpy()
    out['shu__seed'] = np.int64(11)
    for r, t in y.items():
        out[f'shu__y{r}'] = t.numpy()
    # heterogeneous filter alone
    hf = shu.df1
    t = tn(g.standard_normal((1, 64, 64, 33)))
    with torch.no_grad():
        out['hf__y'] = hf(t).numpy()
    out['hf__x'] = t.numpy()
    # the bicubic band weights (shgan.py:116-118) and a SHU built on them
    out['cweight_bicubic_2x3_64x33'] = shgan.make_cweight([2, 3], (64, 33), type='bicubic').numpy()
    out['cweight_bicubic_3x2_16x9'] = shgan.make_cweight([3, 2], (16, 9), type='bicubic').numpy()
    shu_b = shgan.SHU(32, 32, [2, 3], 'bicubic', input_res=64, lowest_res=4, tail_sigma_mult=3, gaussian_at_input_res=False).eval()
    shu_b.load_state_dict(sub, strict=True)
    with torch.no_grad():
        yb = shu_b(x)
    for r, t in yb.items():
        out[f'shu_bicubic__y{r}'] = t.numpy()
    save('shu', **out)


def build_reference_generator(resolution, ch_base, ch_max, w_dim, z_dim, w0_dim):
    num_ws = {256: 14, 512: 16, 1024: 18}[resolution]
    mp = comodgan.Mapping(z_dim=z_dim, c_dim=0, w_dim=w_dim, num_ws=num_ws, num_layers=8,
                          embed_features=None, layer_features=None, activation=ACT,
                          lr_multiplier=0.01, w_avg_beta=0.995)
    enc = shgan.Encoder(resolution=resolution, ic_n=4, oc_n=w0_dim, ch_base=ch_base, ch_max=ch_max,
                        use_fp16_before_res=None, resample_filter=[1, 3, 3, 1], activation=ACT,
                        mbstd_group_size=0, mbstd_c_n=0, c_dim=None, cmap_dim=None, use_dropout=True,
                        has_extra_final_layer=False, shu_channels=32, shu_df_freedom=[2, 3],
                        shu_df_type='piecewise_linear', shu_input_res=64, shu_lowest_res=4,
                        shu_tail_sigma_mult=3, shu_gaussian_at_input_res=False)
    syn = comodgan.Synthesis(w_dim=w_dim, w0_dim=w0_dim, resolution=resolution, rgb_n=3, ch_base=ch_base,
                             ch_max=ch_max, use_fp16_after_res=None, resample_filter=[1, 3, 3, 1],
                             activation=ACT)
    return comodgan.Generator(mp, enc, syn).eval().requires_grad_(False)


def synth_inputs(n, resolution, z_dim, seed):
    """Synthetic masked inputs: real ~ U(-1,1) quantised to u8 levels, freeform
    mask from the reference's RandomMask, x = cat([mask-0.5, real*mask]), z ~ N(0,1)."""
    from lib.data_factory.ds_ffhq import RandomMask
    g = rs(seed)
    real_u8 = g.randint(0, 256, size=(n, 3, resolution, resolution)).astype(np.uint8)
    np.random.seed(seed)
    mask = np.stack([RandomMask(resolution, [0, 1]) for _ in range(n)]).astype(np.uint8)  # [n,1,R,R]
    z = f32(g.standard_normal((n, z_dim)))
    return real_u8, mask, z


def assemble_x(real_u8, mask):
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    m = torch.from_numpy(mask.astype(np.float32))
    return torch.cat([m - 0.5, real * m], dim=1)


def gen_generator_small():
    """A1,A2,A13-A15,A20-A24: full generator at reduced width (R=256, <=32 ch), N=2,
    noise_mode const/none, with intermediates + the uint8 eval composite."""
    cfg = dict(resolution=256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = build_reference_generator(**cfg)
    sd = orc.init_state_dict(256, seed=21, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128,
                             noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    real_u8, mask, z = synth_inputs(2, 256, 64, seed=22)
    x = assemble_x(real_u8, mask)
    zt = torch.from_numpy(z)
    c = torch.zeros(2, 0)
    out = dict(real_u8=real_u8, mask_bits=np.packbits(mask), z=z, seed=np.int64(21),
               cfg=np.array([256, 2048, 32, 64, 64, 128], dtype=np.int64))
    with torch.no_grad():
        ws = G.mapping(zt, c)
        xg, feats = G.encoder(x)
        img = G(x=x, z=zt, c=c, noise_mode='const')
        img_none = G(x=x, z=zt, c=c, noise_mode='none')
    out['ws'] = ws.numpy()
    out['xg'] = xg.numpy()
    for r in (4, 8, 16, 32, 64):
        out[f'feat{r}'] = feats[r].numpy()
    for r in (128, 256):
        out[f'feat{r}_stats'] = np.array([feats[r].mean().item(), feats[r].std().item(),
                                          feats[r].min().item(), feats[r].max().item()])
    out['img_const'] = img.numpy()
    out['img_none_ds'] = img_none[:, :, ::4, ::4].numpy()
    m = x[:, 0:1] + 0.5
    comb = x[:, 1:4] * m + img * (1 - m)
    out['comb_u8'] = (comb * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).numpy()
    # 'random' noise mode with externally supplied noise: emulate by patching torch.randn order
    torch.manual_seed(1234)
    with torch.no_grad():
        img_rand = G(x=x, z=zt, c=c, noise_mode='random')
    out['img_random_seed1234_ds'] = img_rand[:, :, ::4, ::4].numpy()
    save('generator_small', **out)


def gen_generator_small1024():
    """The shipped 1024 configuration (configs/model/shgan.yaml:94-124: num_ws 18, two more encoder / synthesis blocks) at reduced width,
    N=1, noise_mode const: ws, x_global, the low-resolution skip features, the image on a stride-4 grid + sampled pixels, the uint8
    composite's known region as a hash."""
    cfg = dict(resolution=1024, ch_base=4096, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = build_reference_generator(**cfg)
    sd = orc.init_state_dict(1024, seed=71, ch_base=4096, ch_max=32, w_dim=64, z_dim=64, w0_dim=128, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    real_u8, mask, z = synth_inputs(1, 1024, 64, seed=72)
    x = assemble_x(real_u8, mask)
    zt = torch.from_numpy(z)
    c = torch.zeros(1, 0)
    with torch.no_grad():
        ws = G.mapping(zt, c)
        xg, feats = G.encoder(x)
        img = G(x=x, z=zt, c=c, noise_mode='const')
    out = dict(mask_bits=np.packbits(mask), z=z, seeds=np.array([71, 72], dtype=np.int64),
               cfg=np.array([1024, 4096, 32, 64, 64, 128], dtype=np.int64), ws=ws.numpy(), xg=xg.numpy())
    for r in (4, 16, 64):
        out[f'feat{r}'] = feats[r].numpy()
    for r in (256, 512, 1024):
        out[f'feat{r}_stats'] = np.array([feats[r].mean().item(), feats[r].std().item(), feats[r].min().item(), feats[r].max().item()])
    out['img_ds'] = img[:, :, ::4, ::4].numpy()
    idx = rs(73).randint(0, img.numel(), size=512)
    out['sample_idx'], out['sample_val'] = idx, img.flatten()[idx].numpy()
    out['img_stats'] = np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()])
    m = x[:, 0:1] + 0.5
    comb_u8 = ((x[:, 1:4] * m + img * (1 - m)) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)
    out['known_sha256'] = np.array(hashlib.sha256((comb_u8 * torch.from_numpy(mask)).numpy().tobytes()).hexdigest())
    out['state_dict_keys'] = np.array(sorted(G.state_dict().keys()))
    save('generator_small1024', **out)


def gen_generator_full_stats():
    """Full-width G (79.2 M params) at 256x256, N=2 (BASELINE config 1): statistics + sampled
    pixels + a strided slice; weights are re-creatable from the seed."""
    G = build_reference_generator(256, 32768, 512, 512, 512, 1024)
    sd = orc.init_state_dict(256, seed=31)
    G.load_state_dict(sd, strict=True)
    nparam = sum(p.numel() for p in G.parameters())
    real_u8, mask, z = synth_inputs(2, 256, 512, seed=32)
    x = assemble_x(real_u8, mask)
    with torch.no_grad():
        img = G(x=x, z=torch.from_numpy(z), c=torch.zeros(2, 0), noise_mode='const')
    g = rs(33)
    idx = g.randint(0, img.numel(), size=256)
    m = x[:, 0:1] + 0.5
    comb_u8 = ((x[:, 1:4] * m + img * (1 - m)) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)
    save('generator_full256_stats', seed=np.int64(31), input_seed=np.int64(32), nparam=np.int64(nparam),
         stats=np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()]),
         sample_idx=idx, sample_val=img.flatten()[idx].numpy(), img_ds=img[:, :, ::8, ::8].numpy(),
         known_sha256=np.array(hashlib.sha256(
             (comb_u8 * torch.from_numpy(mask)).numpy().tobytes()).hexdigest()),
         state_dict_keys=np.array(sorted(G.state_dict().keys())),
         state_dict_shapes=np.array([str(tuple(G.state_dict()[k].shape)) for k in sorted(G.state_dict().keys())]))


def gen_masks():
    """A25-A27: integer paths -- golden freeform masks (bit-packed), sampler indices, zipzap."""
    from lib.data_factory.ds_ffhq import RandomMask
    from lib.data_factory.common.ds_sampler import DistributedSampler
    from lib.evaluator.eva_base import base_evaluator
    out = {}
    for s in (64, 256, 512):
        np.random.seed(0)
        ms = [RandomMask(s, [0, 1]) for _ in range(4)]
        out[f'mask{s}_bits'] = np.stack([np.packbits(m.astype(np.uint8)) for m in ms])
        out[f'mask{s}_sha'] = np.array([hashlib.sha256(np.packbits(m.astype(np.uint8)).tobytes()).hexdigest()[:16]
                                        for m in ms])
    import lib.data_factory.common.ds_sampler as dss
    dss.print_log = lambda *a, **k: None
    rows = []
    for n_items, world in [(10, 4), (16, 8), (7, 2), (1000, 8), (3, 8)]:
        for rank in range(world):
            smp = DistributedSampler(list(range(n_items)), num_replicas=world, rank=rank, shuffle=False, extend=True)
            rows.append(np.array([n_items, world, rank] + list(iter(smp)), dtype=np.int64))
    out['sampler_rows'] = np.array(rows, dtype=object)
    ev = base_evaluator.__new__(base_evaluator)
    out['zipzap_in'] = np.array([[0, 2, 4, 6], [1, 3, 5, 7]])
    out['zipzap_out'] = np.array(ev.zipzap_arrange([[0, 2, 4, 6], [1, 3, 5, 7]]))
    out['zipzap_out_ragged'] = np.array(ev.zipzap_arrange([[0, 3, 6], [1, 4, 7], [2, 5]]))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'integer_paths.npz')
    np.savez_compressed(path, **out, allow_pickle=True)
    print('  wrote', path)


def gen_generator_full512_stats():
    """Full-width G (79.8 M params) at 512x512, N=1 (BASELINE config 3's model): statistics, sampled pixels, a strided
    slice and the b512 intermediate statistics -- pins the 512 blocks against the reference itself."""
    G = build_reference_generator(512, 32768, 512, 512, 512, 1024)
    sd = orc.init_state_dict(512, seed=35, noise_strength=0.05)
    G.load_state_dict(sd, strict=True)
    nparam = sum(p.numel() for p in G.parameters())
    real_u8, mask, z = synth_inputs(1, 512, 512, seed=36)
    x = assemble_x(real_u8, mask)
    with torch.no_grad():
        xg, feats = G.encoder(x)
        img = G(x=x, z=torch.from_numpy(z), c=torch.zeros(1, 0), noise_mode='const')
    g = rs(37)
    idx = g.randint(0, img.numel(), size=512)
    m = x[:, 0:1] + 0.5
    comb_u8 = ((x[:, 1:4] * m + img * (1 - m)) * 127.5 + 127.5).clamp(0, 255).to(torch.uint8)
    save('generator_full512_stats', seed=np.int64(35), input_seed=np.int64(36), nparam=np.int64(nparam),
         stats=np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()]),
         sample_idx=idx, sample_val=img.flatten()[idx].numpy(), img_ds=img[:, :, ::16, ::16].numpy(),
         feat512_ds=feats[512][:, ::8, ::32, ::32].numpy(), feat256_ds=feats[256][:, ::16, ::16, ::16].numpy(),
         xg=xg.numpy(),
         known_sha256=np.array(hashlib.sha256((comb_u8 * torch.from_numpy(mask)).numpy().tobytes()).hexdigest()),
         comb_u8_ds=comb_u8[:, :, ::16, ::16].numpy(),
         num_keys=np.int64(len(G.state_dict())))


def _sd_arrays(module, prefix='sd__'):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def gen_stylegan2_plain():
    """stylegan.py:436-606 (`stylegan2_mapping` / `stylegan2_synthesis` / `stylegan2_generator`, const-input blocks) and
    the `res_link` branch of `synthesis_block` -- the plain StyleGAN2 modules the co-modulated ones derive from.  Small
    widths; the reference's own initial weights (torch.manual_seed) travel inside the fixture."""
    torch.manual_seed(2024)
    mp = stylegan.Mapping(z_dim=32, c_dim=0, w_dim=32, num_ws=8, num_layers=3, activation=ACT, lr_multiplier=0.01,
                          w_avg_beta=0.995)
    syn = stylegan.Synthesis(w_dim=32, resolution=32, rgb_n=3, ch_base=256, ch_max=16, use_fp16_after_res=32,
                             resample_filter=[1, 3, 3, 1], activation=ACT)
    G = stylegan.Generator(mp, syn).eval().requires_grad_(False)
    for n_, p_ in G.named_parameters():
        if n_.endswith('noise_strength'):
            p_.fill_(0.1)
        if n_.endswith('.bias') and 'affine' not in n_:
            p_.copy_(torch.randn_like(p_) * 0.1)
    z = torch.randn(3, 32)
    out = dict(z=z.numpy(), num_ws=np.int64(G.num_ws))
    out.update(_sd_arrays(G))
    out['img_const'] = G(z, torch.zeros(3, 0), noise_mode='const').numpy()
    out['img_none'] = G(z, torch.zeros(3, 0), noise_mode='none').numpy()
    out['img_trunc'] = G(z, torch.zeros(3, 0), truncation_psi=0.6, truncation_cutoff=4, noise_mode='const').numpy()
    # res_link block (stylegan.py:481-483,500-507): 1x1 up-skip added to the conv branch
    torch.manual_seed(2025)
    blk = stylegan.synthesis_block(8, 12, w_dim=16, resolution=16, rgb_n=3, activation=ACT, res_link=True).eval().requires_grad_(False)
    blk.conv0.noise_strength.fill_(0.2)
    blk.conv1.noise_strength.fill_(0.3)
    x = torch.randn(2, 8, 8, 8)
    img = torch.randn(2, 3, 8, 8)
    ws = torch.randn(2, 3, 16)
    xo, io = blk(x, img, ws, noise_mode='const')
    out.update({'rl__x': x.numpy(), 'rl__img': img.numpy(), 'rl__ws': ws.numpy(), 'rl__x_out': xo.numpy(), 'rl__img_out': io.numpy()})
    out.update(_sd_arrays(blk, 'rlsd__'))
    save('stylegan2_plain', **out)


def gen_discriminator():
    """stylegan.py:624-838: `stylegan2_discriminator` forward (reslink blocks with the down-2 1x1 skip of
    conv2d_resample.py:104-108, `minibatch_std_layer` :686-704, epilogue) at reduced width, N=6 (one full group of 4 + a
    ragged rest is rejected by the reference's reshape, so N is a multiple of the group size... N=8) and N=2 (< group)."""
    torch.manual_seed(2026)
    D = stylegan.Discriminator(resolution=32, ic_n=4, ch_base=256, ch_max=16, use_fp16_before_res=None,
                               resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=4, mbstd_c_n=1,
                               c_dim=None, cmap_dim=None).eval().requires_grad_(False)
    for n_, p_ in D.named_parameters():
        if n_.endswith('.bias'):
            p_.copy_(torch.randn_like(p_) * 0.1)
    out = _sd_arrays(D)
    for n in (8, 2):
        img = torch.randn(n, 4, 32, 32)
        out[f'img{n}'] = img.numpy()
        out[f'logits{n}'] = D(img, None).numpy()
    # the std layer on its own, incl. several feature groups
    x = torch.randn(8, 6, 4, 4)
    out['mb__x'] = x.numpy()
    out['mb__y_g4_f1'] = stylegan.minibatch_std_layer(4, 1)(x).numpy()
    out['mb__y_g2_f3'] = stylegan.minibatch_std_layer(2, 3)(x).numpy()
    out['mb__y_gN_f2'] = stylegan.minibatch_std_layer(None, 2)(x).numpy()
    save('discriminator', **out)


def gen_discriminator_conditional():
    """stylegan.py:707-755: the epilogue's conditional projection `(x * cmap).sum(1) / sqrt(cmap_dim)` (the label-conditioned
    critic itself cannot be built in the reference: Mapping(c_dim > 0) passes an unknown keyword to dense, stylegan.py:377)."""
    torch.manual_seed(2031)
    ep = stylegan.discrim_epilogue(16, resolution=4, cmap_dim=8, rgb_n=3, mbstd_group_size=2, mbstd_c_n=1,
                                   activation=ACT).eval().requires_grad_(False)
    for n_, p_ in ep.named_parameters():
        if n_.endswith('.bias'):
            p_.copy_(torch.randn_like(p_) * 0.1)
    out = _sd_arrays(ep)
    x4, img4, cmap = torch.randn(4, 16, 4, 4), torch.randn(4, 3, 4, 4), torch.randn(4, 8)
    out['x4'], out['img4'], out['cmap'] = x4.numpy(), img4.numpy(), cmap.numpy()
    out['proj'] = ep(x4, img4, cmap).numpy()
    save('discriminator_conditional', **out)


def gen_generator_grads():
    """Training row N3: gradients of a scalar functional of the generator output (sum(img * r) / N, noise_mode='const', dropout
    off) with respect to EVERY generator parameter and to z, through the reference's own modules under autograd at reduced width
    (R=256, <=32 ch).  Parameters larger than 4096 elements are stored as their first 2048 + last 2048 entries plus sum and
    L2 norm."""
    cfg = dict(resolution=256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = build_reference_generator(**cfg)
    sd = orc.init_state_dict(256, seed=41, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    G = G.eval().requires_grad_(True)
    real_u8, mask, z = synth_inputs(2, 256, 64, seed=42)
    x = assemble_x(real_u8, mask)
    rs = np.random.RandomState(43)
    r = rs.standard_normal((2, 3, 256, 256)).astype(np.float32)
    with torch.enable_grad():
        zt = torch.from_numpy(z).requires_grad_(True)
        img = G(x=x, z=zt, c=torch.zeros(2, 0), noise_mode='const')
        loss = (img * torch.from_numpy(r)).sum() / 2
        loss.backward()
    out = dict(real_u8=real_u8, mask_bits=np.packbits(mask), z=z, r_seed=np.int64(43), seed=np.int64(41),
               cfg=np.array([256, 2048, 32, 64, 64, 128], dtype=np.int64), loss=np.float64(loss.item()),
               img_ds=img.detach()[:, :, ::4, ::4].numpy(), grad__z=zt.grad.numpy())
    for n_, p_ in G.named_parameters():
        g = p_.grad.reshape(-1).double()
        out['gsum__' + n_] = np.array([g.sum().item(), g.norm().item()])
        gn = p_.grad.reshape(-1).numpy()
        out['grad__' + n_] = gn if gn.size <= 4096 else np.concatenate([gn[:2048], gn[-2048:]])
    # Gpl (stylegan_default_loss.py:76-91): path-length penalty = second-order path through mapping output -> synthesis
    G.zero_grad()
    pl_noise = rs.standard_normal((2, 3, 256, 256)).astype(np.float32) / np.sqrt(256 * 256)
    with torch.enable_grad():
        ws = G.mapping(torch.from_numpy(z), torch.zeros(2, 0))
        xg, feats = G.encoder(x)
        img = G.synthesis(xg, feats, ws, noise_mode='const')
        pl_grads = torch.autograd.grad(outputs=[(img * torch.from_numpy(pl_noise)).sum()], inputs=[ws], create_graph=True, only_inputs=True)[0]
        pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
        ((img[:, 0, 0, 0] * 0 + pl_lengths.square() * 2.0).mean()).backward()
    out['pl_lengths'] = pl_lengths.detach().numpy()
    for n_, p_ in G.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.reshape(-1).double()
        out['plsum__' + n_] = np.array([g.sum().item(), g.norm().item()])
        gn = p_.grad.reshape(-1).numpy()
        out['plgrad__' + n_] = gn if gn.size <= 4096 else np.concatenate([gn[:2048], gn[-2048:]])
    save('generator_grads', **out)


def gen_discriminator_grads():
    """Training row N3: one discriminator loss evaluation of stylegan_default_loss.py:96-117 (Dmain: softplus(D(fake)) and
    softplus(-D(real)), means, gain 1) through the reference's own modules under autograd -- logits, loss, the gradient of
    every parameter and of the fake image (what the generator step receives)."""
    torch.manual_seed(2027)
    with torch.enable_grad():
        D = stylegan.Discriminator(resolution=32, ic_n=4, ch_base=256, ch_max=16, use_fp16_before_res=None,
                                   resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=4, mbstd_c_n=1,
                                   c_dim=None, cmap_dim=None).train().requires_grad_(True)
        with torch.no_grad():
            for n_, p_ in D.named_parameters():
                if n_.endswith('.bias'):
                    p_.copy_(torch.randn_like(p_) * 0.1)
        out = _sd_arrays(D)
        fake = torch.randn(8, 4, 32, 32).requires_grad_(True)
        real = torch.randn(8, 4, 32, 32)
        lf, lr = D(fake, None), D(real, None)
        loss = torch.nn.functional.softplus(lf).mean() + torch.nn.functional.softplus(-lr).mean()
        loss.backward()
    out['fake'], out['real'] = fake.detach().numpy(), real.numpy()
    out['logits_fake'], out['logits_real'], out['loss'] = lf.detach().numpy(), lr.detach().numpy(), np.float64(loss.item())
    out['grad__fake'] = fake.grad.numpy()
    for n_, p_ in D.named_parameters():
        out['grad__' + n_] = p_.grad.numpy()
    # Dreal + Dr1 (stylegan_default_loss.py:104-127): the R1 penalty differentiates the logits with respect to the real images
    # and is then differentiated with respect to the parameters -- a second-order path through every operator
    D.zero_grad()
    with torch.enable_grad():
        real_tmp = real.detach().requires_grad_(True)
        real_logits = D(real_tmp, None)
        r1_grads = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp], create_graph=True, only_inputs=True)[0]
        r1_penalty = r1_grads.square().sum([1, 2, 3])
        (real_logits * 0 + torch.nn.functional.softplus(-real_logits) + r1_penalty.reshape(-1, 1) * (10.0 / 2)).mean().backward()
    out['r1_penalty'] = r1_penalty.detach().numpy()
    for n_, p_ in D.named_parameters():
        out['r1grad__' + n_] = p_.grad.numpy()
    save('discriminator_grads', **out)


def _sampled(gn, k=512):
    gn = gn.reshape(-1)
    return gn if gn.size <= 2 * k else np.concatenate([gn[:k], gn[-k:]])


def gen_generator_grads64():
    """The first functional of ``gen_generator_grads`` (sum(img * r) / N) once more with the reference's modules redirected to FLOAT64 (the
    ``torch.float32`` casts hard-wired in stylegan.py:403,486,517,567 / comodgan.py:43,238,305,337,398 read ``torch.float64``): the
    yardstick for the round-off of the float32 golden itself.  Stored like the float32 gradients (first / last 2048 entries)."""
    import copy
    cfg = dict(resolution=256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = build_reference_generator(**cfg)
    sd = orc.init_state_dict(256, seed=41, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    real_u8, mask, z = synth_inputs(2, 256, 64, seed=42)
    x = assemble_x(real_u8, mask)
    r = np.random.RandomState(43).standard_normal((2, 3, 256, 256)).astype(np.float32)

    class _Torch64:
        float32 = torch.float64

        def __getattr__(self, k):
            return getattr(torch, k)

    mods = [stylegan, comodgan, shgan, ref_utils, ref_ufd, ref_c2r]
    saved = [m_.torch for m_ in mods]
    out = {}
    try:
        for m_ in mods:
            m_.torch = _Torch64()
        G64 = copy.deepcopy(G).double().eval().requires_grad_(True)
        with torch.enable_grad():
            zt = torch.from_numpy(z).double().requires_grad_(True)
            img = G64(x=x.double(), z=zt, c=torch.zeros(2, 0, dtype=torch.float64), noise_mode='const')
            assert img.dtype == torch.float64
            ((img * torch.from_numpy(r).double()).sum() / 2).backward()
        out['grad__z'] = zt.grad.numpy()
        for n_, p_ in G64.named_parameters():
            gn = p_.grad.reshape(-1).numpy()
            out['grad__' + n_] = gn if gn.size <= 4096 else np.concatenate([gn[:2048], gn[-2048:]])
        # the path-length functional of gen_generator_grads (second order), same draws
        G64.zero_grad()
        rs_ = np.random.RandomState(43)
        rs_.standard_normal((2, 3, 256, 256))
        pl_noise = torch.from_numpy(rs_.standard_normal((2, 3, 256, 256)).astype(np.float32) / np.sqrt(256 * 256)).double()
        with torch.enable_grad():
            ws = G64.mapping(torch.from_numpy(z).double(), torch.zeros(2, 0, dtype=torch.float64))
            xg, feats = G64.encoder(x.double())
            img = G64.synthesis(xg, feats, ws, noise_mode='const')
            pl_grads = torch.autograd.grad(outputs=[(img * pl_noise).sum()], inputs=[ws], create_graph=True, only_inputs=True)[0]
            pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
            ((img[:, 0, 0, 0] * 0 + pl_lengths.square() * 2.0).mean()).backward()
        out['pl_lengths'] = pl_lengths.detach().numpy()
        for n_, p_ in G64.named_parameters():
            if p_.grad is None:
                continue
            gn = p_.grad.reshape(-1).numpy()
            out['plgrad__' + n_] = gn if gn.size <= 4096 else np.concatenate([gn[:2048], gn[-2048:]])
    finally:
        for m_, t_ in zip(mods, saved):
            m_.torch = t_
    save('generator_grads64', **out)


def gen_config5_step():
    """BASELINE config 5 at FULL width (FFHQ-512: ch_base 32768, ch_max 512, w/z 512, w0 1024; discriminator 512, ic_n 4), batch 2:
    the four phases of stylegan_default_loss.py:53-128 -- Gmain :56-66, Dmain :94-127 (Dgen + Dreal), Dreg (R1, gamma 10) :118-124,
    Greg (path length on the shrunk batch, pl_weight 2) :69-91 -- evaluated with the REFERENCE's own modules under torch autograd on CPU.
    The reference's loss class itself is not importable (undefined misc / training_stats / dnnlib), so the formulae are
    applied here to its modules; the co-modulated generator takes x = cat([mask-.5, real*mask]) and the discriminator
    cat([mask-.5, image]) (ic_n = 4).  train() mode (non-fused modulated convolutions, stylegan.py:172-181, comodgan.py:307-309) with the
    encoder's dropout switched off and noise_mode='const' so that the run is deterministic.  Weights: generator
    ``init_state_dict(512, seed)``, discriminator ``seeded_fill_(D, seed)``; parameter gradients are stored as first / last 512 entries
    + (sum, L2 norm)."""
    R, N = 512, 2
    G = build_reference_generator(R, 32768, 512, 512, 512, 1024)
    sd = orc.init_state_dict(R, seed=51, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    G = G.train()
    for m in G.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    D = stylegan.Discriminator(resolution=R, ic_n=4, ch_base=32768, ch_max=512, use_fp16_before_res=None,
                               resample_filter=[1, 3, 3, 1], activation=ACT, mbstd_group_size=4, mbstd_c_n=1,
                               c_dim=None, cmap_dim=None).train()
    orc.seeded_fill_(D, seed=52, bias_std=0.1)
    real_u8, mask, z = synth_inputs(N, R, 512, seed=53)
    x = assemble_x(real_u8, mask)
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    m05 = x[:, 0:1]
    cnd = torch.zeros(N, 0)
    zt = torch.from_numpy(z)
    sp = torch.nn.functional.softplus
    out = dict(mask_bits=np.packbits(mask), z=z, seeds=np.array([51, 52, 53, 54], dtype=np.int64), cfg=np.array([R, N], dtype=np.int64))

    def store(prefix, module):
        for n_, p_ in module.named_parameters():
            if p_.grad is None:
                continue
            g = p_.grad.reshape(-1).double()
            out[prefix + 'sum__' + n_] = np.array([g.sum().item(), g.norm().item()])
            out[prefix + '__' + n_] = _sampled(p_.grad.numpy()).astype(np.float32)

    import time
    t0 = time.time()
    # ---- Gmain (:56-66)
    G.requires_grad_(True); D.requires_grad_(False)
    with torch.enable_grad():
        img = G(x=x, z=zt, c=cnd, noise_mode='const')
        logits = D(torch.cat([m05, img], 1), None)
        loss = sp(-logits).mean()
        loss.backward()
    out['img_ds'] = img.detach()[:, :, ::8, ::8].numpy()
    out['img_stats'] = np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()])
    out['gmain_logits'], out['gmain_loss'] = logits.detach().numpy(), np.float64(loss.item())
    store('gmain', G)
    fake = img.detach()
    print(f'    Gmain done {time.time() - t0:.0f}s  loss {loss.item():.6f}')
    # ---- Dmain (:94-127)
    G.zero_grad(); G.requires_grad_(False); D.requires_grad_(True)
    with torch.enable_grad():
        lf = D(torch.cat([m05, fake], 1), None)
        sp(lf).mean().backward()
        lr = D(torch.cat([m05, real], 1), None)
        sp(-lr).mean().backward()
    out['dmain_logits_fake'], out['dmain_logits_real'] = lf.detach().numpy(), lr.detach().numpy()
    out['dmain_loss'] = np.float64((sp(lf) + sp(-lr)).mean().item())
    store('dmain', D)
    print(f'    Dmain done {time.time() - t0:.0f}s')
    # ---- Dreg (:104-127, r1_gamma = 10)
    D.zero_grad()
    with torch.enable_grad():
        rt = torch.cat([m05, real], 1).detach().requires_grad_(True)
        lr = D(rt, None)
        r1g = torch.autograd.grad(outputs=[lr.sum()], inputs=[rt], create_graph=True, only_inputs=True)[0]
        r1 = r1g.square().sum([1, 2, 3])
        (lr * 0 + (r1 * 5.0).reshape(-1, 1)).mean().backward()
    out['r1_penalty'] = r1.detach().numpy()
    store('dreg', D)
    print(f'    Dreg done {time.time() - t0:.0f}s  r1 {r1.detach().numpy()}')
    # ---- Greg (:69-91, batch // pl_batch_shrink = 1, pl_weight = 2, pl_mean = 0 -> lerp(mean, 0.01))
    D.zero_grad(); D.requires_grad_(False); G.requires_grad_(True)
    pl_noise = torch.from_numpy(rs(54).standard_normal((1, 3, R, R)).astype(np.float32)) / np.sqrt(R * R)
    with torch.enable_grad():
        ws = G.mapping(zt[:1], cnd[:1])
        xg, feats = G.encoder(x[:1])
        img1 = G.synthesis(xg, feats, ws, noise_mode='const')
        plg = torch.autograd.grad(outputs=[(img1 * pl_noise).sum()], inputs=[ws], create_graph=True, only_inputs=True)[0]
        pll = plg.square().sum(2).mean(1).sqrt()
        plm = torch.zeros([]).lerp(pll.mean(), 0.01)
        (img1[:, 0, 0, 0] * 0 + (pll - plm).square() * 2.0).mean().backward()          # (pl_mean carries its graph, :84-86)
    out['pl_lengths'], out['pl_mean'] = pll.detach().numpy(), np.float64(plm.item())
    store('greg', G)
    print(f'    Greg done {time.time() - t0:.0f}s  pl {pll.detach().numpy()}')
    # ---- the same Gmain and Greg in FLOAT64 (the reference's modules with their hard-wired ``torch.float32`` casts -- stylegan.py:403,
    # 486,517,567, comodgan.py:43,238,305,337,398 -- redirected to float64): the yardstick for the fp32 round-off of ~1e5-term
    # sums in the generator's gradients.  Tests hold the HIP path's distance to float64 against the reference's OWN fp32 distance.
    import copy

    class _Torch64:
        float32 = torch.float64

        def __getattr__(self, k):
            return getattr(torch, k)

    mods = [stylegan, comodgan, shgan, ref_utils, ref_ufd, ref_c2r]
    saved = [m_.torch for m_ in mods]
    try:
        for m_ in mods:
            m_.torch = _Torch64()
        G64, D64 = copy.deepcopy(G).double(), copy.deepcopy(D).double()
        x64, z64, m64 = x.double(), zt.double(), m05.double()
        G64.zero_grad(); G64.requires_grad_(True); D64.requires_grad_(False)
        with torch.enable_grad():
            img = G64(x=x64, z=z64, c=cnd.double(), noise_mode='const')
            assert img.dtype == torch.float64
            logits = D64(torch.cat([m64, img], 1), None)
            sp(-logits).mean().backward()
        out['gmain64_logits'] = logits.detach().numpy()
        out['img64_ds'] = img.detach()[:, :, ::8, ::8].numpy()
        store('gmain64', G64)
        print(f'    Gmain fp64 done {time.time() - t0:.0f}s')
        G64.zero_grad()
        with torch.enable_grad():
            ws = G64.mapping(z64[:1], cnd[:1].double())
            xg, feats = G64.encoder(x64[:1])
            img1 = G64.synthesis(xg, feats, ws, noise_mode='const')
            plg = torch.autograd.grad(outputs=[(img1 * pl_noise.double()).sum()], inputs=[ws], create_graph=True, only_inputs=True)[0]
            pll = plg.square().sum(2).mean(1).sqrt()
            plm = torch.zeros([], dtype=torch.float64).lerp(pll.mean(), 0.01)
            (img1[:, 0, 0, 0] * 0 + (pll - plm).square() * 2.0).mean().backward()
        out['pl_lengths64'] = pll.detach().numpy()
        store('greg64', G64)
        print(f'    Greg fp64 done {time.time() - t0:.0f}s  pl {pll.detach().numpy()}')
    finally:
        for m_, t_ in zip(mods, saved):
            m_.torch = t_
    save('config5_step512', **out)


def gen_fp16():
    """The reference's ``use_fp16`` branches, run by the reference itself on CPU (torch 2.10 has half convolutions / FIR on CPU):
    operator level -- conv2d_resample branches, upfirdn2d, the non-fused modulated convolution with the fp16 pre-normalisation
    (stylegan.py:136-138,172-181), lrelu_agc -- on half tensors (each case also in float32 on the SAME half-rounded inputs: the
    fp32-accumulate answer the MFMA path should round to); module level -- the discriminator with ``use_fp16_before_res`` (logits,
    every parameter gradient, the gradient of the input image), the co-modulated SH-GAN generator with fp16 encoder / synthesis blocks
    (image, parameter gradients).  Encoder blocks at the SHU's input resolution stay float32: torch.fft rejects half on CPU
    ('Unsupported dtype Half'), so ``use_fp16_before_res`` >= 64 is the most the CPU reference can run."""
    g = rs(700)
    f4 = ref_ufd.setup_filter([1, 3, 3, 1])
    out, names = {'f': f4.numpy()}, []

    def h(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32)).half()

    cases = [
        # name, x shape, w shape, up, down, padding, flip_weight
        ('plain_3x3', (2, 32, 20, 24), (48, 32, 3, 3), 1, 1, 1, True),
        ('plain_3x3_wide', (1, 64, 17, 24), (160, 64, 3, 3), 1, 1, 1, True),
        ('plain_3x3_noflip', (2, 16, 12, 12), (24, 16, 3, 3), 1, 1, 1, False),
        ('plain_1x1', (2, 48, 9, 17), (40, 48, 1, 1), 1, 1, 0, True),
        ('fromrgb_1x1', (2, 4, 16, 16), (32, 4, 1, 1), 1, 1, 0, True),
        ('torgb_1x1', (2, 32, 16, 16), (3, 32, 1, 1), 1, 1, 0, True),
        ('down2_3x3', (2, 32, 32, 32), (48, 32, 3, 3), 1, 2, 1, True),
        ('down2_3x3_odd', (1, 16, 18, 22), (16, 16, 3, 3), 1, 2, 1, True),
        ('up2_3x3', (2, 32, 16, 16), (24, 32, 3, 3), 2, 1, 1, False),
        ('up2_3x3_rect', (1, 48, 9, 20), (64, 48, 3, 3), 2, 1, 1, False),
        ('down2_1x1', (2, 32, 16, 16), (40, 32, 1, 1), 1, 2, 0, True),
        ('up2_1x1', (2, 32, 8, 8), (16, 32, 1, 1), 2, 1, 0, True),
    ]
    for name, xs, ws, up, down, pad, flipw in cases:
        x = h(g.standard_normal(xs))
        w = h(g.standard_normal(ws) / np.sqrt(ws[1] * ws[2] * ws[3]))
        f = f4 if (up > 1 or down > 1) else None
        y16 = ref_c2r.conv2d_resample(x=x, w=w, f=f, up=up, down=down, padding=pad, flip_weight=flipw)
        y32 = ref_c2r.conv2d_resample(x=x.float(), w=w.float(), f=f, up=up, down=down, padding=pad, flip_weight=flipw)
        assert y16.dtype == torch.float16
        names.append(name)
        out['c2r__' + name + '__x'], out['c2r__' + name + '__w'] = x.numpy(), w.numpy()
        out['c2r__' + name + '__cfg'] = np.array([up, down, pad, int(flipw), int(f is not None)], dtype=np.int64)
        out['c2r__' + name + '__y16'], out['c2r__' + name + '__y32'] = y16.numpy(), y32.numpy()
    out['c2r_names'] = np.array(names)
    # upfirdn2d on halves
    ufd = [('pad2', (2, 16, 12, 14), 1, 1, [2, 2, 2, 2], 1.0), ('post_up', (1, 8, 17, 17), 1, 1, [1, 1, 1, 1], 4.0),
           ('up2', (2, 8, 8, 8), 2, 1, [2, 1, 2, 1], 4.0), ('down2', (2, 24, 16, 16), 1, 2, [1, 1, 1, 1], 1.0)]
    for name, xs, up, down, pad, gain in ufd:
        x = h(g.standard_normal(xs))
        out['ufd__' + name + '__x'] = x.numpy()
        out['ufd__' + name + '__cfg'] = np.array([up, down] + pad, dtype=np.int64)
        out['ufd__' + name + '__gain'] = np.float32(gain)
        out['ufd__' + name + '__y16'] = ref_ufd.upfirdn2d(x, f4, up=up, down=down, padding=pad, gain=gain).numpy()
        out['ufd__' + name + '__y32'] = ref_ufd.upfirdn2d(x.float(), f4, up=up, down=down, padding=pad, gain=gain).numpy()
    out['ufd_names'] = np.array([u[0] for u in ufd])
    # non-fused modulated convolution on halves incl. the pre-normalisation; weights / styles are float32 parameters as in the layers
    mc = [('same', (3, 32, 16, 16), (48, 32, 3, 3), 1, True, True), ('up', (2, 32, 8, 8), (24, 32, 3, 3), 2, True, True),
          ('torgb', (2, 32, 16, 16), (3, 32, 1, 1), 1, False, False)]
    for name, xs, ws, up, demod, with_noise in mc:
        x = h(g.standard_normal(xs) * 3.0)
        w = tn(g.standard_normal(ws))
        st = tn(g.standard_normal((xs[0], xs[1])) + 1.0)
        oh = xs[2] * up
        noise = tn(g.standard_normal((oh, oh)) * 0.1) if with_noise else None
        k = ws[2]
        y16 = stylegan.modulated_conv2d(x=x, weight=w, styles=st, noise=noise, up=up, padding=k // 2, resample_filter=f4 if up > 1 else None,
                                        demodulate=demod, flip_weight=(up == 1), fused_modconv=False)
        y32 = stylegan.modulated_conv2d(x=x.float(), weight=w, styles=st, noise=noise, up=up, padding=k // 2, resample_filter=f4 if up > 1 else None,
                                        demodulate=demod, flip_weight=(up == 1), fused_modconv=False)
        for key, val in (('x', x), ('w', w), ('styles', st), ('y16', y16), ('y32', y32)):
            out['mc__' + name + '__' + key] = val.numpy()
        if noise is not None:
            out['mc__' + name + '__noise'] = noise.numpy()
        out['mc__' + name + '__cfg'] = np.array([up, int(demod)], dtype=np.int64)
    out['mc_names'] = np.array([m[0] for m in mc])
    act = ref_utils.get_unit()(ACT)()
    xa = h(g.standard_normal((2, 16, 8, 8)) * 200.0)
    out['act__x'], out['act__y'] = xa.numpy(), act(xa.clone(), gain=np.sqrt(0.5)).numpy()

    # ---- discriminator with fp16 blocks (stylegan.py:660-667,788): R=64, blocks 64 and 32 in half
    with torch.enable_grad():
        D = stylegan.Discriminator(resolution=64, ic_n=4, ch_base=1024, ch_max=32, use_fp16_before_res=16, resample_filter=[1, 3, 3, 1],
                                   activation=ACT, mbstd_group_size=4, mbstd_c_n=1, c_dim=None, cmap_dim=None).train().requires_grad_(True)
        orc.seeded_fill_(D, seed=71, bias_std=0.1)
        img = torch.from_numpy(rs(72).standard_normal((4, 4, 64, 64)).astype(np.float32)).requires_grad_(True)
        logits = D(img, None)
        loss = torch.nn.functional.softplus(logits).mean()
        loss.backward()
    out['D__logits'], out['D__loss'], out['D__grad_img'] = logits.detach().numpy(), np.float64(loss.item()), img.grad.numpy()[:, :, ::2, ::2]      # (img = rs(72) draw)
    for n_, p_ in D.named_parameters():
        out['D__grad__' + n_] = p_.grad.numpy()
    D32 = stylegan.Discriminator(resolution=64, ic_n=4, ch_base=1024, ch_max=32, use_fp16_before_res=None, resample_filter=[1, 3, 3, 1],
                                 activation=ACT, mbstd_group_size=4, mbstd_c_n=1, c_dim=None, cmap_dim=None).train()
    D32.load_state_dict(D.state_dict())
    out['D__logits_fp32'] = D32(img.detach(), None).detach().numpy()

    # ---- SH-GAN generator with fp16 blocks: encoder 256 / 128 (use_fp16_before_res = 64), synthesis 64 / 128 / 256 (use_fp16_after_res = 32)
    cfg = dict(resolution=256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G32 = build_reference_generator(**cfg)
    num_ws = 14
    mp = comodgan.Mapping(z_dim=64, c_dim=0, w_dim=64, num_ws=num_ws, num_layers=8, embed_features=None, layer_features=None, activation=ACT,
                          lr_multiplier=0.01, w_avg_beta=0.995)
    enc = shgan.Encoder(resolution=256, ic_n=4, oc_n=128, ch_base=2048, ch_max=32, use_fp16_before_res=64, resample_filter=[1, 3, 3, 1],
                        activation=ACT, mbstd_group_size=0, mbstd_c_n=0, c_dim=None, cmap_dim=None, use_dropout=True, has_extra_final_layer=False,
                        shu_channels=32, shu_df_freedom=[2, 3], shu_df_type='piecewise_linear', shu_input_res=64, shu_lowest_res=4,
                        shu_tail_sigma_mult=3, shu_gaussian_at_input_res=False)
    syn = comodgan.Synthesis(w_dim=64, w0_dim=128, resolution=256, rgb_n=3, ch_base=2048, ch_max=32, use_fp16_after_res=32,
                             resample_filter=[1, 3, 3, 1], activation=ACT)
    G = comodgan.Generator(mp, enc, syn)
    sd = orc.init_state_dict(256, seed=73, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    G32.load_state_dict(sd, strict=True)
    real_u8, mask, z = synth_inputs(2, 256, 64, seed=74)
    x = assemble_x(real_u8, mask)
    out['G__mask_bits'], out['G__z'] = np.packbits(mask), z               # (real_u8 = the first draw of rs(74), synth_inputs)
    out['G__seed'] = np.int64(73)
    for mode in ('eval', 'train'):
        G = G.eval() if mode == 'eval' else G.train()
        for m in G.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        G.zero_grad()
        G.requires_grad_(mode == 'train')
        with torch.set_grad_enabled(mode == 'train'):
            img = G(x=x, z=torch.from_numpy(z), c=torch.zeros(2, 0), noise_mode='const')
            if mode == 'train':
                r = torch.from_numpy(rs(75).standard_normal((2, 3, 256, 256)).astype(np.float32))
                ((img * r).sum() / 2).backward()
        out[f'G__img_{mode}'] = img.detach().numpy()[:, :, ::4, ::4]
        out[f'G__img_{mode}_stats'] = np.array([img.mean().item(), img.std().item(), img.abs().max().item()])
    for n_, p_ in G.named_parameters():
        gn = p_.grad.reshape(-1)
        out['G__gnorm__' + n_] = np.array([gn.double().norm().item()])
        out['G__grad__' + n_] = _sampled(gn.numpy(), 256)
    out['G__img_fp32'] = G32.eval()(x=x, z=torch.from_numpy(z), c=torch.zeros(2, 0), noise_mode='const').numpy()[:, :, ::4, ::4]
    # the same gradients from the float32 generator: the yardstick for the half-precision noise of the reference's own fp16 gradients
    # (broadcast reductions and ~1e4-term cancellations carried out in half)
    G32 = G32.train().requires_grad_(True)
    for m in G32.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with torch.enable_grad():
        img = G32(x=x, z=torch.from_numpy(z), c=torch.zeros(2, 0), noise_mode='const')
        ((img * r).sum() / 2).backward()
    for n_, p_ in G32.named_parameters():
        out['G__grad32__' + n_] = _sampled(p_.grad.reshape(-1).numpy(), 256)
    save('fp16', **out)


def gen_fid():
    """N1: the reference's own FID tail -- ``base_evaluator.zipzap_arrange`` (eva_base.py:196-230) applied per batch the way
    ``fid_evaluator.add_batch`` arranges the per-rank features after ``sync`` (eva_fid.py:217-237), then ``compute_fid``
    (eva_fid.py:239-263) incl. its ``[0:sample_n]`` truncation of the padded tail.  The detector (a TorchScript download,
    eva_fid.py:145-158) and the broadcasts of ``sync_`` are bypassed: the per-rank feature arrays go straight into the list
    ``sync`` would have returned.  Rank shards come from the reference's own ``DistributedSampler(extend=True)``."""
    from lib.evaluator import eva_fid
    from lib.data_factory.common import ds_sampler as ref_sampler
    cases = {}
    # name: (D, n_items, world, batch per rank, sample_fake_n, sample_real_n)
    spec = dict(w1_d64=(64, 300, 1, 32, None, None),              # one rank, ragged last batch
                w2_d64_padded=(64, 301, 2, 16, None, None),       # odd dataset on two ranks: one padded duplicate
                w4_d2048_padded=(2048, 66, 4, 6, None, None),    # the real feature width, two padded duplicates, n << D
                w2_d64_trunc=(64, 300, 2, 25, 200, 260))          # explicit sample_fake_n / sample_real_n
    for ci, (name, (D, n, world, bs, sfn, srn)) in enumerate(spec.items()):
        g = rs(4100 + ci)
        mix = g.standard_normal((D, D)) / np.sqrt(D)
        real = f32(np.abs(g.standard_normal((n, D)) @ mix + 0.3 * g.standard_normal(D)))
        fake = f32(np.abs(g.standard_normal((n, D)) @ (mix + 0.15 * g.standard_normal((D, D)) / np.sqrt(D)) + 0.3 * g.standard_normal(D) + 0.05))
        shards = [list(ref_sampler.DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=False, extend=True))
                  for r in range(world)]
        ev = object.__new__(eva_fid.fid_evaluator)          # (no __init__: it downloads the detector)
        ev.final, ev.sample_n, ev.sample_fake_n, ev.sample_real_n = {}, n, sfn, srn
        ev.dsstat_use_cache, ev.dsstat_cache_file = False, None
        ev.data_fake_feat, ev.data_real_feat = [], []
        per_rank = len(shards[0])
        for k0 in range(0, per_rank, bs):
            ids = [sh[k0:k0 + bs] for sh in shards]
            ff = tuple(fake[i].astype(float) for i in ids)   # what zip(*sync([fake_feat, real_feat, fn])) yields (eva_fid.py:217-219)
            rf = tuple(real[i].astype(float) for i in ids)
            ev.data_fake_feat.append(ev.zipzap_arrange(ff))
            ev.data_real_feat.append(ev.zipzap_arrange(rf))
        fid = float(ev.compute_fid())
        cases[name + '__real'], cases[name + '__fake'] = real, fake
        cases[name + '__meta'] = np.asarray([D, n, world, bs, -1 if sfn is None else sfn, -1 if srn is None else srn], dtype=np.int64)
        cases[name + '__fid'] = np.asarray(fid, dtype=np.float64)
        cases[name + '__shards'] = np.asarray(shards, dtype=np.int64)
        print(f'    {name}: fid = {fid:.10f}')
    save('fid', names=np.asarray(list(spec.keys())), **cases)


def gen_fp16_fused():
    """The reference's FUSED modulated convolution on half tensors (stylegan.py:149-170,183-193: per-sample weights w * s * d rounded to
    half once, grouped convolution with groups = N), the form its blocks take in eval when the batch is ONE image
    (``fused_modconv = (not training) and (fp32 or N == 1)``, stylegan.py:490, comodgan.py:242,309) -- operator level for N = 1 and N = 2,
    and the SH-GAN generator with fp16 blocks on a batch of one (eval).  Run by the reference itself on CPU."""
    g = rs(900)
    f4 = ref_ufd.setup_filter([1, 3, 3, 1])
    out = {'f': f4.numpy()}

    def h(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32)).half()

    mc = [('same_n1', (1, 32, 16, 16), (48, 32, 3, 3), 1, True, True), ('up_n1', (1, 32, 8, 8), (24, 32, 3, 3), 2, True, True),
          ('torgb_n1', (1, 32, 16, 16), (3, 32, 1, 1), 1, False, False), ('same_n2', (2, 32, 12, 16), (40, 32, 3, 3), 1, True, True),
          ('up_n3', (3, 16, 8, 8), (24, 16, 3, 3), 2, True, True), ('torgb_n2', (2, 32, 16, 16), (3, 32, 1, 1), 1, False, False),
          ('same_n1_wide', (1, 64, 20, 20), (64, 64, 3, 3), 1, True, False)]
    for name, xs, ws, up, demod, with_noise in mc:
        x = h(g.standard_normal(xs) * 3.0)
        w = tn(g.standard_normal(ws))
        st = tn(g.standard_normal((xs[0], xs[1])) + 1.0)
        noise = tn(g.standard_normal((xs[2] * up, xs[3] * up)) * 0.1) if with_noise else None
        k = ws[2]
        kw = dict(weight=w, styles=st, noise=noise, up=up, padding=k // 2, resample_filter=f4 if up > 1 else None, demodulate=demod,
                  flip_weight=(up == 1), fused_modconv=True)
        y16 = stylegan.modulated_conv2d(x=x, **kw)
        y32 = stylegan.modulated_conv2d(x=x.float(), **kw)
        assert y16.dtype == torch.float16
        for key, val in (('x', x), ('w', w), ('styles', st), ('y16', y16), ('y32', y32)):
            out['mc__' + name + '__' + key] = val.numpy()
        if noise is not None:
            out['mc__' + name + '__noise'] = noise.numpy()
        out['mc__' + name + '__cfg'] = np.array([up, int(demod)], dtype=np.int64)
    out['mc_names'] = np.array([m[0] for m in mc])

    # ---- SH-GAN generator with fp16 blocks, ONE image, eval: every half block takes the fused form
    cfg = dict(resolution=256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G32 = build_reference_generator(**cfg)
    mp = comodgan.Mapping(z_dim=64, c_dim=0, w_dim=64, num_ws=14, num_layers=8, embed_features=None, layer_features=None, activation=ACT,
                          lr_multiplier=0.01, w_avg_beta=0.995)
    enc = shgan.Encoder(resolution=256, ic_n=4, oc_n=128, ch_base=2048, ch_max=32, use_fp16_before_res=64, resample_filter=[1, 3, 3, 1],
                        activation=ACT, mbstd_group_size=0, mbstd_c_n=0, c_dim=None, cmap_dim=None, use_dropout=True, has_extra_final_layer=False,
                        shu_channels=32, shu_df_freedom=[2, 3], shu_df_type='piecewise_linear', shu_input_res=64, shu_lowest_res=4,
                        shu_tail_sigma_mult=3, shu_gaussian_at_input_res=False)
    syn = comodgan.Synthesis(w_dim=64, w0_dim=128, resolution=256, rgb_n=3, ch_base=2048, ch_max=32, use_fp16_after_res=32,
                             resample_filter=[1, 3, 3, 1], activation=ACT)
    G = comodgan.Generator(mp, enc, syn).eval()
    sd = orc.init_state_dict(256, seed=83, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128, noise_strength=0.1, bias_std=0.1)
    G.load_state_dict(sd, strict=True)
    G32.load_state_dict(sd, strict=True)
    real_u8, mask, z = synth_inputs(1, 256, 64, seed=84)
    x = assemble_x(real_u8, mask)
    out['G__mask_bits'], out['G__z'], out['G__seed'] = np.packbits(mask), z, np.int64(83)
    # which form every modulated layer call took (the claim this fixture exists for): record the flag the reference passed down
    seen = []
    orig = stylegan.modulated_conv2d

    def spy(*a, **k):
        seen.append((str(k['x'].dtype).replace('torch.', ''), bool(k.get('fused_modconv', True)), int(k['x'].shape[0])))
        return orig(*a, **k)
    stylegan.modulated_conv2d = spy
    try:
        img = G(x=x, z=torch.from_numpy(z), c=torch.zeros(1, 0), noise_mode='const')
    finally:
        stylegan.modulated_conv2d = orig
    assert any(d == 'float16' and fz for d, fz, _ in seen), seen
    assert all(fz for _, fz, _ in seen), seen
    out['G__calls_half'] = np.int64(sum(1 for d, _, _ in seen if d == 'float16'))
    out['G__img_eval'] = img.numpy()[:, :, ::2, ::2]
    out['G__img_eval_stats'] = np.array([img.mean().item(), img.std().item(), img.abs().max().item()])
    out['G__img_fp32'] = G32.eval()(x=x, z=torch.from_numpy(z), c=torch.zeros(1, 0), noise_mode='const').numpy()[:, :, ::2, ::2]
    # the same batch of one through the NON-fused form (what a batch of 2+ takes): how far apart the reference's own two forms are
    img_nf = None
    def spy_nf(*a, **k):
        if k['x'].dtype == torch.float16:
            k['fused_modconv'] = False
        return orig(*a, **k)
    stylegan.modulated_conv2d = spy_nf
    try:
        img_nf = G(x=x, z=torch.from_numpy(z), c=torch.zeros(1, 0), noise_mode='const')
    finally:
        stylegan.modulated_conv2d = orig
    out['G__img_eval_nonfused'] = img_nf.numpy()[:, :, ::2, ::2]
    save('fp16_fused', **out)


GENS = dict(upfirdn2d=gen_upfirdn2d, conv2d_resample=gen_conv2d_resample, modconv=gen_modconv,
            small_ops=gen_small_ops, shu=gen_shu, generator_small=gen_generator_small,
            generator_full_stats=gen_generator_full_stats, masks=gen_masks,
            generator_full512_stats=gen_generator_full512_stats, stylegan2_plain=gen_stylegan2_plain,
            discriminator=gen_discriminator, discriminator_grads=gen_discriminator_grads, generator_grads=gen_generator_grads,
            discriminator_conditional=gen_discriminator_conditional, fid=gen_fid, config5_step=gen_config5_step, fp16=gen_fp16, generator_small1024=gen_generator_small1024,
            generator_grads64=gen_generator_grads64, fp16_fused=gen_fp16_fused)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    for k, fn in GENS.items():
        if a.only and k not in a.only:
            continue
        print('[gen]', k)
        fn()
