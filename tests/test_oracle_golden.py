"""CPU: pin the oracle (oracle/shgan_oracle.py) against golden vectors produced by the
reference itself (tools/gen_golden.py).  Tolerances: the oracle uses the same torch CPU
ops as the reference, so agreement is at fp32 round-off (<= 2e-6 relative); integer paths
are bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import shgan_oracle as orc

TOL = 2e-6


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_upfirdn2d_cases():
    g = load_golden('upfirdn2d')
    for name in g['names']:
        x, f, cfg = t(g[f'{name}__x']), g[f'{name}__f'], g[f'{name}__cfg']
        f = t(f) if f.size else None
        upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in cfg]
        y = orc.upfirdn2d(x, f, up=(upx, upy), down=(dnx, dny), padding=[px0, px1, py0, py1],
                          flip_filter=bool(flip), gain=float(g[f'{name}__gain']))
        ref = g[f'{name}__y']
        assert tuple(y.shape) == ref.shape, name
        assert rel_err(y.numpy(), ref) < TOL, name


def test_upfirdn2d_helpers_and_setup_filter():
    g = load_golden('upfirdn2d')
    x, f = t(g['helpers__x']), t(g['helpers__f'])
    assert rel_err(orc.upsample2d(x, f).numpy(), g['helpers__up']) < TOL
    assert rel_err(orc.downsample2d(x, f).numpy(), g['helpers__down']) < TOL
    assert rel_err(orc.filter2d(x, f).numpy(), g['helpers__filt']) < TOL
    assert np.array_equal(orc.setup_filter([1, 3, 3, 1]).numpy(), g['setup_filter_1331'])
    assert np.allclose(orc.setup_filter([1, 2, 3, 4, 5, 4, 3, 2, 1]).numpy(), g['setup_filter_sep'], rtol=1e-7)
    assert np.allclose(orc.setup_filter([[1, 2], [3, 4]], flip_filter=True, gain=4).numpy(),
                       g['setup_filter_gain_flip'], rtol=1e-7)


def test_conv2d_resample_branches():
    g = load_golden('conv2d_resample')
    f4 = t(g['f'])
    for name in g['names']:
        up, down, pad, groups, flipw, hasf = [int(v) for v in g[f'{name}__cfg']]
        y = orc.conv2d_resample(t(g[f'{name}__x']), t(g[f'{name}__w']), f=(f4 if hasf else None), up=up,
                                down=down, padding=pad, groups=groups, flip_weight=bool(flipw))
        assert tuple(y.shape) == g[f'{name}__y'].shape, name
        assert rel_err(y.numpy(), g[f'{name}__y']) < TOL, name


def test_modulated_conv2d_all_modes():
    g = load_golden('modulated_conv2d')
    f4 = t(g['f'])
    for name in g['names']:
        up, demod, fused, k = [int(v) for v in g[f'{name}__cfg']]
        y = orc.modulated_conv2d(t(g[f'{name}__x']), t(g[f'{name}__w']), t(g[f'{name}__s']),
                                 noise=t(g[f'{name}__noise']), up=up, padding=k // 2,
                                 resample_filter=(f4 if up > 1 else None), demodulate=bool(demod),
                                 flip_weight=(up == 1), fused_modconv=bool(fused))
        assert rel_err(y.numpy(), g[f'{name}__y']) < 5e-6, name


def test_small_ops():
    g = load_golden('small_ops')
    x = t(g['lrelu__x'])
    assert np.array_equal(orc.lrelu_agc(x.clone()).numpy(), g['lrelu__y_gain1'])
    assert np.array_equal(orc.lrelu_agc(x.clone(), gain=np.sqrt(0.5)).numpy(), g['lrelu__y_gain_sqrt_half'])
    assert np.array_equal(orc.lrelu_agc(x.clone(), alpha=0.1, act_gain=1, clamp=None).numpy(),
                          g['lrelu__y_noclamp'])
    for tag in ('mapping', 'affine', 'fc'):
        lr, use_act = g[f'dense_{tag}__cfg']
        y = orc.dense(t(g[f'dense_{tag}__x']), t(g[f'dense_{tag}__w']), t(g[f'dense_{tag}__b']),
                      lr_multi=float(lr), act=bool(use_act))
        assert rel_err(y.numpy(), g[f'dense_{tag}__y']) < TOL, tag
    assert rel_err(orc.fma(t(g['fma__a']), t(g['fma__b']), t(g['fma__c'])).numpy(), g['fma__y']) < TOL


def test_shu_tables():
    g = load_golden('shu')
    cw_ref = g['cweight_2x3_64x33']
    assert np.abs(orc.make_cweight_closed_form(64, 33).numpy() - cw_ref).max() < 1e-6
    assert np.abs(orc.make_cweight_grid_sample((2, 3), (64, 33)).numpy() - cw_ref).max() < 1e-6
    tabs = orc.gaussian_split_tables(64, 4, 3.0)
    for r in (4, 8, 16, 32, 64):
        assert tabs[r].shape == g[f'gauss_{r}'].shape
        assert np.abs(tabs[r].numpy() - g[f'gauss_{r}']).max() < 1e-7, r
    tot = np.zeros((64, 33))
    for r in (4, 8, 16, 32, 64):
        tot[32 - r // 2: 32 + r // 2, : r // 2 + 1] += tabs[r].numpy()
    assert np.abs(tot - 1).max() < 1e-6        # partition of unity


def test_shu_end_to_end():
    g = load_golden('shu')
    sd = orc.init_state_dict(256, seed=int(g['shu__seed']), ch_base=2048, ch_max=32, w_dim=64, z_dim=64,
                             w0_dim=128, bias_std=0.2)
    out = orc.shu_forward(sd, t(g['shu__x']))
    for r in (4, 8, 16, 32, 64):
        assert rel_err(out[r].numpy(), g[f'shu__y{r}']) < 1e-5, r
    cw = orc.make_cweight_closed_form(64, 33)
    y = orc.heterogeneous_filter(t(g['hf__x']), sd['encoder.shu.df1.weight'], cw)
    assert rel_err(y.numpy(), g['hf__y']) < 1e-5


def _small_inputs(g):
    real = torch.from_numpy(g['real_u8'].astype(np.float32)) / 127.5 - 1.0
    n, _, r, _ = real.shape
    mask = torch.from_numpy(np.unpackbits(g['mask_bits'])[: n * r * r].reshape(n, 1, r, r).astype(np.float32))
    return torch.cat([mask - 0.5, real * mask], dim=1), torch.from_numpy(g['z'])


def test_generator_small_against_reference():
    g = load_golden('generator_small')
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seed']), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim,
                             w0_dim=w0_dim, noise_strength=0.1, bias_std=0.1)
    x, z = _small_inputs(g)
    img, mid = orc.generator_forward(sd, x, z, res, noise_mode='const', return_intermediates=True)
    assert rel_err(mid['ws'].numpy(), g['ws']) < TOL
    assert rel_err(mid['xg'].numpy(), g['xg']) < 1e-5
    for r in (4, 8, 16, 32, 64):
        assert rel_err(mid['feats'][r].numpy(), g[f'feat{r}']) < 1e-5, r
    assert rel_err(img.numpy(), g['img_const']) < 2e-5
    img_none = orc.generator_forward(sd, x, z, res, noise_mode='none')
    assert rel_err(img_none[:, :, ::4, ::4].numpy(), g['img_none_ds']) < 2e-5
    # non-fused modconv algebra == fused (what the HIP path uses)
    img_nf = orc.generator_forward(sd, x, z, res, noise_mode='const', fused_modconv=False)
    assert rel_err(img_nf.numpy(), g['img_const']) < 5e-5
    u8 = orc.composite_u8(x, torch.from_numpy(g['img_const']))
    assert np.array_equal(u8.numpy(), g['comb_u8'])
    # known region is bit-exact == the real image
    m = (x[:, 0:1] + 0.5).numpy().astype(bool)
    assert np.array_equal(np.where(m, g['comb_u8'], 0), np.where(m, g['real_u8'], 0))


def test_generator_small1024_against_reference():
    """configs/model/shgan.yaml:94-124 (shgan_g1024: num_ws 18, two more blocks) at reduced width, N=1."""
    g = load_golden('generator_small1024')
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seeds'][0]), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim,
                             noise_strength=0.1, bias_std=0.1)
    assert sorted(sd.keys()) == list(g['state_dict_keys'])
    x, z, real_u8, mask = orc.synthetic_batch(1, res, z_dim, seed=int(g['seeds'][1]))
    assert np.array_equal(np.packbits(mask), g['mask_bits']) and np.array_equal(z.numpy(), g['z'])
    img, mid = orc.generator_forward(sd, x, z, res, noise_mode='const', return_intermediates=True)
    assert mid['ws'].shape[1] == 18
    assert rel_err(mid['ws'].numpy(), g['ws']) < TOL
    assert rel_err(mid['xg'].numpy(), g['xg']) < 1e-5
    for r in (4, 16, 64):
        assert rel_err(mid['feats'][r].numpy(), g[f'feat{r}']) < 1e-5, r
    assert rel_err(img[:, :, ::4, ::4].numpy(), g['img_ds']) < 2e-5
    assert rel_err(img.flatten()[torch.from_numpy(g['sample_idx'])].numpy(), g['sample_val']) < 2e-5


def test_generator_full256_stats():
    """BASELINE config 1: full-width 256x256, batch 2, random-init, CPU."""
    g = load_golden('generator_full256_stats')
    sd = orc.init_state_dict(256, seed=int(g['seed']))
    assert sorted(sd.keys()) == list(g['state_dict_keys'])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == list(g['state_dict_shapes'])
    x, z, _, _ = orc.synthetic_batch(2, 256, 512, seed=int(g['input_seed']))
    img = orc.generator_forward(sd, x, z, 256, noise_mode='const')
    assert rel_err(img[:, :, ::8, ::8].numpy(), g['img_ds']) < 5e-5
    assert rel_err(img.flatten()[torch.from_numpy(g['sample_idx'])].numpy(), g['sample_val']) < 5e-5
    st = np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()])
    assert np.allclose(st, g['stats'], rtol=1e-4, atol=1e-4)


def test_integer_paths_bit_exact():
    g = load_golden('integer_paths')
    for s in (64, 256, 512):
        np.random.seed(0)
        for i in range(4):
            m = orc.random_mask(s, [0, 1])
            assert m.shape == (1, s, s) and m.dtype == np.float32
            bits = np.packbits(m.astype(np.uint8))
            assert np.array_equal(bits, g[f'mask{s}_bits'][i]), (s, i)
            assert hashlib.sha256(bits.tobytes()).hexdigest()[:16] == str(g[f'mask{s}_sha'][i])
    for row in g['sampler_rows']:
        n_items, world, rank = [int(v) for v in row[:3]]
        assert orc.sampler_indices(n_items, world, rank, extend=True) == [int(v) for v in row[3:]]
    assert orc.zipzap_arrange([[0, 2, 4, 6], [1, 3, 5, 7]]) == list(g['zipzap_out'])
    assert orc.zipzap_arrange([[0, 3, 6], [1, 4, 7], [2, 5]]) == list(g['zipzap_out_ragged'])


def test_oracle_full_width_512_matches_reference():
    """The 512 blocks of the oracle against the reference-generated fixture (round 1 pinned them only through their symmetry
    with the 256 blocks): full-width 512x512, batch 1."""
    g = load_golden('generator_full512_stats')
    sd = orc.init_state_dict(512, seed=int(g['seed']), noise_strength=0.05)
    x, z, _, mask = orc.synthetic_batch(1, 512, 512, seed=int(g['input_seed']))
    with torch.no_grad():
        img, inter = orc.generator_forward(sd, x, z, 512, noise_mode='const', return_intermediates=True)
    assert rel_err(inter['xg'].numpy(), g['xg']) < 2e-6
    assert rel_err(inter['feats'][512].numpy()[:, ::8, ::32, ::32], g['feat512_ds']) < 2e-6
    assert rel_err(img.numpy()[:, :, ::16, ::16], g['img_ds']) < 1e-5
    assert rel_err(img.numpy().flatten()[g['sample_idx']], g['sample_val']) < 1e-5
    u8 = orc.composite_u8(x, img).numpy()
    import hashlib
    assert hashlib.sha256((u8 * mask).tobytes()).hexdigest() == str(g['known_sha256'])
