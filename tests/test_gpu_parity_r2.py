"""GPU parity, round 2 additions: direct oracle tests of the kernels that round 1 covered only end to end (torgb with
and without the fused RGB up-sampling, the thin 1x1 convolution, bias_act in every operand combination, scale_channels,
the generic upfirdn2d epilogue), reference-generated fixtures for the 512 model, the plain StyleGAN2 modules and the
discriminator, full-batch 512x16 against the oracle, an element-wise relative check, and the launch-side guards."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def mods():
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, eval_harness, kernels
    from shgan_amd.model_zoo import shgan, stylegan
    from shgan_amd.model_zoo.stylegan_utils import upfirdn2d
    from oracle import shgan_oracle as orc
    return dict(kernels=kernels, stylegan=stylegan, shgan=shgan, orc=orc, configs=configs, harness=eval_harness, ufd=upfirdn2d)


def c(a):
    return a.detach().cpu().numpy()


def rnd(rs, *shape):
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


def elementwise_rel(a, b, floor):
    """max over elements of |a-b| / max(|b|, floor): the element-wise counterpart of conftest.rel_err."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


@pytest.mark.parametrize('n,ci,h,w', [(2, 64, 64, 64), (3, 37, 8, 8), (1, 512, 4, 4), (2, 128, 32, 64)])
@pytest.mark.parametrize('with_base', [False, True])
def test_torgb_kernel_vs_oracle(mods, n, ci, h, w, with_base):
    """torgb_layer (stylegan.py:325-337) + `upsample2d(img) + y` (comodgan.py:331-338) in one kernel."""
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(11)
    x, wt, styles, bias = rnd(rs, n, ci, h, w), rnd(rs, 3, ci), rnd(rs, n, ci), rnd(rs, 3)
    f = orc.setup_filter([1, 3, 3, 1])
    ref = orc.modulated_conv2d(x, wt.view(3, ci, 1, 1), styles, demodulate=False) + bias.view(1, -1, 1, 1)
    base = None
    if with_base:
        base = rnd(rs, n, 3, h // 2, w // 2)
        ref = ref + orc.upsample2d(base, f)
    y = k.torgb(x.to(DEV), wt.to(DEV), styles.to(DEV), bias.to(DEV), base_up=None if base is None else base.to(DEV),
                f=f.to(DEV) if with_base else None)
    assert rel_err(c(y), ref.numpy()) < 1e-5


@pytest.mark.parametrize('act,alpha,act_gain,clamp,gain', [(True, 0.2, 2 ** 0.5, 256.0, 1.0), (True, 0.1, 1.0, None, 0.5),
                                                          (True, 0.3, 1.7, 0.4, 1.0), (False, 0.2, 1.0, None, 0.25)])
def test_thin_conv1x1_and_dense_activation_arguments(mods, act, alpha, act_gain, clamp, gain):
    """fromrgb (stylegan.py:640-642) and dense (stylegan.py:87-98) with non-default lrelu_agc arguments."""
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(12)
    x, wt, b = rnd(rs, 3, 4, 16, 24), rnd(rs, 20, 4), rnd(rs, 20)
    ref = torch.einsum('oi,nihw->nohw', wt * 0.5, x) + b.view(1, -1, 1, 1)
    ref = orc.lrelu_agc(ref, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp) if act else ref * gain
    y = k.conv1x1_thin_in(x.to(DEV), wt.to(DEV), b.to(DEV), wgain=0.5, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
    assert rel_err(c(y), ref.numpy()) < 1e-5
    xd, wd, bd = rnd(rs, 5, 96), rnd(rs, 33, 96), rnd(rs, 33)
    refd = xd @ (wd * 0.1).t() + bd * 2.0
    refd = orc.lrelu_agc(refd, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp) if act else refd * gain
    yd = k.dense(xd.to(DEV), wd.to(DEV), bd.to(DEV), wgain=0.1, bgain=2.0, act=act, gain=gain, alpha=alpha, act_gain=act_gain, clamp=clamp)
    assert rel_err(c(yd), refd.numpy()) < 1e-5


@pytest.mark.parametrize('n,i,o,h,w', [(8, 3, 512, 4, 4), (2, 3, 512, 8, 8), (3, 3, 70, 2, 6), (8, 3, 512, 64, 64), (1, 4, 64, 34, 50), (2, 8, 33, 16, 16),
                                       (5, 1, 7, 2, 2), (2, 3, 256, 128, 128), (70, 3, 16, 4, 4)])
def test_thin_conv1x1_every_grid_shape(mods, n, i, o, h, w):
    """``conv1x1_thin_in`` (fromRGB and the toRGB input gradient: 3 -> C at EVERY resolution of the synthesis network) spreads its output
    channels over the grid when the image is small (workgroup = pixel groups x channel slices): every (pixels, channels, batch) regime
    against a float64 einsum, bias and activation on."""
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(n * 31 + o)
    x, wt, b = rnd(rs, n, i, h, w), rnd(rs, o, i), rnd(rs, o)
    ref = torch.einsum('oi,nihw->nohw', wt.double() * 0.7, x.double()) + b.double().view(1, -1, 1, 1)
    y = k.conv1x1_thin_in(x.to(DEV), wt.to(DEV), b.to(DEV), wgain=0.7, act=False, gain=1.0)
    assert tuple(y.shape) == (n, o, h, w) and rel_err(c(y), ref.numpy()) < 2e-6
    ya = k.conv1x1_thin_in(x.to(DEV), wt.to(DEV), b.to(DEV), wgain=0.7, act=True, gain=0.5)
    assert rel_err(c(ya), orc.lrelu_agc(ref, gain=0.5).numpy()) < 2e-6


def test_bias_act_all_operand_combinations(mods):
    """y = act((x*scale) + noise*strength + bias) + residual with every subset of the optional operands
    (stylegan.py:175-180,232-238; common/utils.py:135-143)."""
    import itertools
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(13)
    n, ch, h, w = 3, 5, 6, 10
    x, scale, bias, res = rnd(rs, n, ch, h, w), rnd(rs, n * ch), rnd(rs, ch), rnd(rs, n, ch, h, w)
    noises = {'none': None, 'shared': rnd(rs, h, w), 'per_sample': rnd(rs, n, 1, h, w)}
    for use_scale, use_bias, use_res, nk, act in itertools.product((0, 1), (0, 1), (0, 1), noises, (True, False)):
        ref = x * scale.view(n, ch, 1, 1) if use_scale else x.clone()
        if noises[nk] is not None:
            ref = ref + noises[nk] * 0.7
        if use_bias:
            ref = ref + bias.view(1, -1, 1, 1)
        ref = orc.lrelu_agc(ref, gain=0.6, alpha=0.15, act_gain=1.3, clamp=1.5) if act else ref * 0.6
        if use_res:
            ref = ref + res
        y = k.bias_act(x.to(DEV), bias=bias.to(DEV) if use_bias else None, scale=scale.to(DEV) if use_scale else None,
                       noise=None if noises[nk] is None else noises[nk].to(DEV), noise_strength=0.7,
                       residual=res.to(DEV) if use_res else None, act=act, gain=0.6, alpha=0.15, act_gain=1.3, clamp=1.5)
        assert rel_err(c(y), ref.numpy()) < 1e-6, (use_scale, use_bias, use_res, nk, act)


def test_scale_channels_and_fma(mods):
    k = mods['kernels']
    rs = np.random.RandomState(14)
    x, s = rnd(rs, 4, 7, 9, 11), rnd(rs, 4 * 7)
    assert torch.equal(k.scale_channels(x.to(DEV), s.to(DEV)).cpu(), x * s.view(4, 7, 1, 1))          # one rounding: exact
    a, b, cc = rnd(rs, 2, 3, 5, 5), rnd(rs, 2, 3, 1, 1), rnd(rs, 1, 3, 5, 5)
    assert rel_err(c(k.fma(a.to(DEV), b.to(DEV), cc.to(DEV))), torch.addcmul(cc, a, b).numpy()) < 1e-6


@pytest.mark.parametrize('up,down,pad,fshape', [(2, 1, [2, 1, 2, 1], (4, 4)), (1, 2, [1, 1, 1, 1], (4, 4)), (1, 1, [2, 1, 0, 3], (3, 5)),
                                                ([2, 3], [3, 2], [3, 2, 4, 1], (3, 5))])
def test_generic_upfirdn2d_epilogue_vs_oracle(mods, up, down, pad, fshape):
    """The generic gather (upfirdn2d.cu:29-92 semantics) with the fused layer tail, on geometries the fast paths do not take."""
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(15)
    n, ch = 2, 3
    x, f = rnd(rs, n, ch, 13, 11), rnd(rs, *fshape)
    ref0 = orc.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=1.5)
    scale, bias, res = rnd(rs, n * ch), rnd(rs, ch), rnd(rs, *ref0.shape)
    noise = rnd(rs, *ref0.shape[2:])
    ref = orc.lrelu_agc(ref0 * scale.view(n, ch, 1, 1) + noise * 0.2 + bias.view(1, -1, 1, 1), gain=0.8) + res
    upx, upy = (up, up) if isinstance(up, int) else up
    dnx, dny = (down, down) if isinstance(down, int) else down
    y = k.upfirdn2d(x.to(DEV), f.to(DEV), upx, upy, dnx, dny, pad[0], pad[1], pad[2], pad[3], gain=1.5,
                    epilogue=dict(scale=scale.to(DEV), bias=bias.to(DEV), noise=noise.to(DEV), noise_strength=0.2,
                                  residual=res.to(DEV), act=True, gain=0.8))
    assert rel_err(c(y), ref.numpy()) < 1e-5


def test_heterogeneous_filter_standalone_forward(mods):
    """shgan.py:143-160 called on its own, on the SHU geometry and on another one."""
    orc, shgan = mods['orc'], mods['shgan']
    rs = np.random.RandomState(16)
    for h, w in ((64, 33), (16, 32)):
        hf = shgan.heterogeneous_filter(8, 8, freedom=[2, 3], type='piecewise_linear').to(DEV)
        wt = (1.0 / 8 + rnd(rs, 8, 48) * 0.1)
        with torch.no_grad():
            hf.weight.copy_(wt.to(DEV))
        x = rnd(rs, 2, 8, h, w)
        cw = shgan.make_cweight([2, 3], (h, w))
        ref = orc.heterogeneous_filter(x, wt, cw)
        assert rel_err(c(hf(x.to(DEV))), ref.numpy()) < 1e-5


def test_operands_on_cpu_or_mixed_devices_are_rejected(mods):
    from shgan_amd._lib import ShgError
    k = mods['kernels']
    x = torch.zeros(1, 2, 4, 4, device=DEV)
    with pytest.raises(ShgError):
        k.bias_act(x, bias=torch.zeros(2))                     # CPU operand
    with pytest.raises(ShgError):
        k.scale_channels(torch.zeros(1, 2, 4, 4), torch.zeros(2))
    # non-contiguous operands are copied and kept alive until the launch (two temporaries must not alias)
    rs = np.random.RandomState(17)
    sc, bs = rnd(rs, 6, 2).to(DEV), rnd(rs, 2, 2).to(DEV)
    xx = rnd(rs, 3, 2, 4, 4).to(DEV)
    y = k.bias_act(xx, scale=sc[:, 0], bias=bs[:, 1], act=False)
    ref = xx * sc[:, 0].reshape(3, 2, 1, 1) + bs[:, 1].reshape(1, 2, 1, 1)
    assert rel_err(c(y), c(ref)) < 1e-6


def test_generator_full_width_512_golden(mods):
    """Reference-generated fixture of the full-width 512 model (BASELINE config 3's network, batch 1)."""
    g = load_golden('generator_full512_stats')
    orc, hz, cfgs = mods['orc'], mods['harness'], mods['configs']
    sd = orc.init_state_dict(512, seed=int(g['seed']), noise_strength=0.05)
    G = cfgs.build_generator(512)
    G.load_state_dict(sd, strict=True)
    assert len(G.state_dict()) == int(g['num_keys']) and sum(p.numel() for p in G.parameters()) == int(g['nparam'])
    G = G.eval().requires_grad_(False).to(DEV)
    x, z, _, mask = orc.synthetic_batch(1, 512, 512, seed=int(g['input_seed']))
    xg, feats = G.encoder(x.to(DEV))
    assert rel_err(c(xg), g['xg']) < 1e-4
    assert rel_err(c(feats[512])[:, ::8, ::32, ::32], g['feat512_ds']) < 1e-4
    assert rel_err(c(feats[256])[:, ::16, ::16, ::16], g['feat256_ds']) < 1e-4
    img = G(x=x.to(DEV), z=z.to(DEV), c=torch.zeros(1, 0, device=DEV), noise_mode='const')
    assert rel_err(c(img)[:, :, ::16, ::16], g['img_ds']) < 1e-3
    assert rel_err(c(img).flatten()[g['sample_idx']], g['sample_val']) < 1e-3
    assert elementwise_rel(c(img).flatten()[g['sample_idx']], g['sample_val'], floor=1e-2 * float(np.abs(g['sample_val']).max())) < 1e-3
    st = np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()])
    assert np.allclose(st, g['stats'], rtol=1e-3, atol=1e-3)
    u8 = c(hz.run_generator(G, x.to(DEV), z.to(DEV), noise_mode='const'))
    d = np.abs(u8[:, :, ::16, ::16].astype(np.int32) - g['comb_u8_ds'].astype(np.int32))
    assert d.max() <= 1
    import hashlib
    assert hashlib.sha256((u8 * mask).tobytes()).hexdigest() == str(g['known_sha256'])     # known region: bit-exact


def test_full_batch_512x16_vs_oracle_four_images(mods):
    """BASELINE config 3 at full size: the 16-image batch on the GPU against the CPU oracle on four of its images
    (the oracle runs them as 1-image batches: the batch-global style norm cancels under demodulation),
    with the global-max AND an element-wise relative criterion."""
    orc, hz, cfgs = mods['orc'], mods['harness'], mods['configs']
    G = cfgs.seeded_init_(cfgs.build_generator(512), seed=81, noise_strength=0.05)
    sd = {k_: v.detach().clone() for k_, v in G.state_dict().items()}
    G = G.eval().requires_grad_(False).to(DEV)
    x, z, _, _ = hz.synthetic_items(list(range(16)), 512, 512, seed=82, device=DEV)
    img = G(x=x, z=z, c=torch.zeros(16, 0, device=DEV), noise_mode='const')
    for i in (0, 5, 10, 15):
        ref = orc.generator_forward(sd, x[i:i + 1].cpu(), z[i:i + 1].cpu(), 512, noise_mode='const').numpy()
        got = c(img[i:i + 1])
        assert rel_err(got, ref) < 1e-4, i
        assert elementwise_rel(got, ref, floor=1e-2 * float(np.abs(ref).max())) < 1e-3, i


def test_full_batch_256x32_vs_oracle_four_images(mods):
    """BASELINE config 2 at full size (FFHQ-256, batch 32): the same four-image oracle comparison as 512 x 16 above, global-max and
    element-wise criteria, on images from both ends and the middle of the batch."""
    orc, hz, cfgs = mods['orc'], mods['harness'], mods['configs']
    G = cfgs.seeded_init_(cfgs.build_generator(256), seed=83, noise_strength=0.05)
    sd = {k_: v.detach().clone() for k_, v in G.state_dict().items()}
    G = G.eval().requires_grad_(False).to(DEV)
    x, z, _, _ = hz.synthetic_items(list(range(32)), 256, 512, seed=84, device=DEV)
    img = G(x=x, z=z, c=torch.zeros(32, 0, device=DEV), noise_mode='const')
    for i in (0, 11, 20, 31):
        ref = orc.generator_forward(sd, x[i:i + 1].cpu(), z[i:i + 1].cpu(), 256, noise_mode='const').numpy()
        got = c(img[i:i + 1])
        assert rel_err(got, ref) < 1e-4, i
        assert elementwise_rel(got, ref, floor=1e-2 * float(np.abs(ref).max())) < 1e-3, i


def test_sharded_eval_full_width_512_batch16(mods):
    """One rank's share of BASELINE config 4 (full-width 512, batch 16 per GPU): 2 emulated ranks x 16 images against the
    unsharded run of the same 32-item dataset; the known pixels are bit-identical, the holes differ by <= 1 LSB."""
    hz, cfgs = mods['harness'], mods['configs']
    from shgan_amd.data import zipzap_arrange
    G = cfgs.seeded_init_(cfgs.build_generator(512), seed=83).eval().requires_grad_(False).to(DEV)
    ids1, out1 = hz.sharded_eval(G, 32, 16, 512, rank=0, world=1, seed=5, gather=False, device=DEV)
    parts = [hz.sharded_eval(G, 32, 16, 512, rank=r, world=2, seed=5, gather=False, device=DEV) for r in range(2)]
    order = zipzap_arrange([p[0] for p in parts])[:32]
    merged = zipzap_arrange([c(p[1]) for p in parts])[:32]
    assert order == ids1 == list(range(32)) and merged.shape == (32, 3, 512, 512)
    d = np.abs(merged.astype(np.int32) - c(out1).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-2
    _, _, real_u8, mask = hz.synthetic_items(list(range(32)), 512, 512, seed=5, device=DEV)
    m = c(mask).astype(bool)
    assert np.array_equal(np.where(m, merged, 0), np.where(m, c(real_u8), 0))


def _load_sd(module, g, prefix):
    sd = {k_[len(prefix):]: torch.from_numpy(g[k_]) for k_ in g.files if k_.startswith(prefix)}
    module.load_state_dict(sd, strict=True)
    return module.eval().requires_grad_(False).to(DEV)


def test_plain_stylegan2_generator_golden(mods):
    """stylegan.py:436-606: const-input synthesis, the plain Generator and the res_link block, against the reference's
    own outputs with the reference's own weights (strict state-dict load)."""
    g = load_golden('stylegan2_plain')
    sg = mods['stylegan']
    mp = sg.Mapping(z_dim=32, c_dim=0, w_dim=32, num_ws=8, num_layers=3, lr_multiplier=0.01, w_avg_beta=0.995)
    syn = sg.Synthesis(w_dim=32, resolution=32, rgb_n=3, ch_base=256, ch_max=16, use_fp16_after_res=32)
    G = _load_sd(sg.Generator(mp, syn), g, 'sd__')
    assert G.num_ws == int(g['num_ws'])
    z = torch.from_numpy(g['z']).to(DEV)
    cnd = torch.zeros(3, 0, device=DEV)
    assert rel_err(c(G(z, cnd, noise_mode='const')), g['img_const']) < 1e-4
    assert rel_err(c(G(z, cnd, noise_mode='none')), g['img_none']) < 1e-4
    assert rel_err(c(G(z, cnd, truncation_psi=0.6, truncation_cutoff=4, noise_mode='const')), g['img_trunc']) < 1e-4
    blk = _load_sd(sg.synthesis_block(8, 12, w_dim=16, resolution=16, rgb_n=3, res_link=True), g, 'rlsd__')
    xo, io = blk(torch.from_numpy(g['rl__x']).to(DEV), torch.from_numpy(g['rl__img']).to(DEV),
                 torch.from_numpy(g['rl__ws']).to(DEV), noise_mode='const')
    assert rel_err(c(xo), g['rl__x_out']) < 1e-4 and rel_err(c(io), g['rl__img_out']) < 1e-4


def test_discriminator_forward_golden(mods):
    """Next row N3, forward only: `stylegan2_discriminator` (stylegan.py:757-838) incl. `minibatch_std_layer`."""
    g = load_golden('discriminator')
    sg, k = mods['stylegan'], mods['kernels']
    x = torch.from_numpy(g['mb__x']).to(DEV)
    assert rel_err(c(k.minibatch_std(x, 4, 1)), g['mb__y_g4_f1']) < 1e-5
    assert rel_err(c(k.minibatch_std(x, 2, 3)), g['mb__y_g2_f3']) < 1e-5
    assert rel_err(c(k.minibatch_std(x, None, 2)), g['mb__y_gN_f2']) < 1e-5
    D = _load_sd(sg.Discriminator(resolution=32, ic_n=4, ch_base=256, ch_max=16, use_fp16_before_res=None,
                                  mbstd_group_size=4, mbstd_c_n=1), g, 'sd__')
    for n in (8, 2):
        logits = D(torch.from_numpy(g[f'img{n}']).to(DEV), None)
        assert tuple(logits.shape) == (n, 1)
        assert rel_err(c(logits), g[f'logits{n}']) < 1e-4, n


def test_discriminator_epilogue_conditional_projection_golden(mods):
    """`discrim_epilogue` with `cmap_dim` (stylegan.py:707-755): fromrgb + minibatch-std + conv + fc + out and the projection on the
    label embedding, against the reference; the label-conditioned critic itself raises (the reference's constructor does too)."""
    g = load_golden('discriminator_conditional')
    sg = mods['stylegan']
    ep = _load_sd(sg.discrim_epilogue(16, resolution=4, cmap_dim=8, rgb_n=3, mbstd_group_size=2, mbstd_c_n=1), g, 'sd__')
    out = ep(torch.from_numpy(g['x4']).to(DEV), torch.from_numpy(g['img4']).to(DEV), torch.from_numpy(g['cmap']).to(DEV))
    assert tuple(out.shape) == (4, 1) and rel_err(c(out), g['proj']) < 1e-4
    with pytest.raises(NotImplementedError):
        sg.Discriminator(resolution=16, ic_n=3, ch_base=128, ch_max=16, use_fp16_before_res=None, c_dim=5, cmap_dim=8)


POLY_UP_CASES = [
    # n, ci, co, h, w: ragged channel counts (I % 8, O % 64), several tiles per image, both tile shapes of both schemes
    (2, 16, 64, 32, 32), (1, 13, 70, 34, 40), (2, 72, 130, 64, 64), (1, 8, 3, 32, 128), (3, 24, 24, 66, 36), (1, 128, 64, 128, 128),
    (2, 40, 70, 16, 16), (1, 16, 64, 18, 20), (16, 64, 64, 16, 16),      # 8 x 8 block tiles of both schemes (images narrower than 32)
    (2, 512, 512, 16, 16), (4, 256, 100, 32, 32), (1, 200, 64, 16, 20),  # small grids: split along the input channels (strips included; ragged last slice)
]


def _interleave(planes, h, w):
    """[4,N,O,H+1,W+1] phase planes -> [N,O,2H+1,2W+1]."""
    _, n, o = planes.shape[:3]
    full = torch.zeros((n, o, 2 * h + 1, 2 * w + 1), dtype=planes.dtype, device=planes.device)
    for a in range(2):
        for b in range(2):
            full[:, :, a::2, b::2] = planes[a * 2 + b][:, :, :h + 1 - a, :w + 1 - b]
    return full


@pytest.mark.parametrize('n,ci,co,h,w', POLY_UP_CASES)
@pytest.mark.parametrize('flip', [False, True])
def test_polyphase_winograd_transposed_conv(mods, n, ci, co, h, w, flip):
    """shg_conv2d_up_poly_f32 (ee as F(3x3,2x2), eo/oe/oo with 16 multiplies per 2x2 block, strips) against torch CPU
    conv_transpose2d and against the direct all-phase MFMA kernel, with per-sample input scales (styles)."""
    import torch.nn.functional as F
    kk = mods['kernels']
    rs = np.random.RandomState(n * 100 + ci + co + h)
    x, wt, s = rnd(rs, n, ci, h, w), rnd(rs, co, ci, 3, 3), torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5)
    wref = wt.flip([2, 3]) if flip else wt
    ref = F.conv_transpose2d(x * s[:, :, None, None], wref.transpose(0, 1), stride=2)
    pw = kk.conv_weight_prep(wt.to(DEV), flip=flip)
    old = kk.UP_POLY
    try:
        kk.UP_POLY = True
        timer = kk.KernelTimer()
        kk.set_timer(timer)
        a = kk.conv2d(x.to(DEV), pw, mode=kk.MODE_UP2T, in_scale=s.to(DEV), planar=True)
        kk.set_timer(None)
        torch.cuda.synchronize()
        assert 'conv_poly_up' in timer.summary()                   # the polyphase route really ran
        kk.UP_POLY = False
        b = kk.conv2d(x.to(DEV), pw, mode=kk.MODE_UP2T, in_scale=s.to(DEV), planar=True)
    finally:
        kk.UP_POLY = old
        kk.set_timer(None)
    fa, fb = _interleave(a, h, w), _interleave(b, h, w)
    assert rel_err(c(fa), ref.numpy()) < 2e-5
    assert rel_err(c(fa), c(fb)) < 2e-5


POLY_DOWN_CASES = [
    # n, ci, co, h, w (input extent; output h/2 x w/2): flat tiles (output rows of 11 / 22 / 43 blocks), both rectangular
    # shapes, 8 x 8 block tiles for narrow outputs, ragged channel counts
    (2, 16, 64, 64, 64), (1, 13, 70, 36, 40), (2, 24, 130, 128, 128), (1, 8, 24, 32, 256), (1, 64, 64, 256, 256), (1, 8, 8, 64, 96),
    (3, 40, 40, 32, 32),
]


@pytest.mark.parametrize('n,ci,co,h,w', POLY_DOWN_CASES)
def test_polyphase_winograd_stride2_conv(mods, n, ci, co, h, w):
    """fir_down_planar + shg_conv2d_down_poly_f32 (P_ee term as F(3x3,2x2), the other planes with 16 multiplies per 2x2
    block, fused bias / lrelu_agc / skip) against the oracle's FIR + torch CPU strided convolution, and against the
    direct route (upfirdn2d + MFMA stride-2 kernel)."""
    import torch.nn.functional as F
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(n * 100 + ci + co + h)
    x, wt, bias = rnd(rs, n, ci, h, w), rnd(rs, co, ci, 3, 3), rnd(rs, co)
    res = rnd(rs, n, co, h // 2, w // 2)
    f = torch.from_numpy(rs.rand(4, 4).astype(np.float32))                      # asymmetric filter
    xf = orc.upfirdn2d(x, f, padding=[2, 2, 2, 2])
    ref = orc.lrelu_agc(F.conv2d(xf, wt * 0.05, stride=2) + bias.view(1, -1, 1, 1), gain=0.7) + res
    pw = kk.conv_weight_prep(wt.to(DEV), gain=0.05)
    assert kk.down_poly_supported(x.to(DEV), pw, force=True)
    # poison the allocator's free blocks: the intermediate planes come from torch.empty, and whatever they do not overwrite
    # (pitch padding, rows / columns outside a plane) must not reach a valid output through the Winograd transforms
    junk = torch.full((max(8 * n * ci * (h + 8) * (w + 16), 1 << 22),), float('nan'), device=DEV)
    del junk
    timer = kk.KernelTimer()
    kk.set_timer(timer)
    try:
        y = kk.fir_conv_down2(x.to(DEV), f.to(DEV), pw, bias=bias.to(DEV), act=True, gain=0.7, residual=res.to(DEV))
    finally:
        kk.set_timer(None)
    torch.cuda.synchronize()
    assert 'conv_poly_down' in timer.summary()
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel_err(c(y), ref.numpy()) < 2e-5
    yd = kk.conv2d(kk.upfirdn2d(x.to(DEV), f.to(DEV), padx0=2, padx1=2, pady0=2, pady1=2), pw, mode=kk.MODE_DOWN2, pad=0,
                   bias=bias.to(DEV), act=True, gain=0.7, residual=res.to(DEV))
    assert rel_err(c(y), c(yd)) < 2e-5
    # no activation, no bias, no skip: the raw sum of the two schemes
    y0 = kk.fir_conv_down2(x.to(DEV), f.to(DEV), pw)
    assert rel_err(c(y0), F.conv2d(xf, wt * 0.05, stride=2).numpy()) < 2e-5


@pytest.mark.parametrize('up,down,pad,fshape,shape', [(2, 1, [2, 1, 2, 1], (4, 4), (2, 3, 8, 9)), (1, 2, [1, 1, 1, 1], (4, 4), (1, 2, 16, 14)),
                                                      (1, 1, [2, 2, 2, 2], (4, 4), (2, 2, 9, 9)), ([2, 1], [1, 3], [3, 2, 1, 2], (3, 5), (1, 2, 7, 12))])
def test_upfirdn2d_backward_is_the_swapped_operator(mods, up, down, pad, fshape, shape):
    """Next row N3: d/dx of upfirdn2d (upfirdn2d.py:174-192) = upfirdn2d with up <-> down, flipped filter; checked against
    torch autograd through the CPU oracle."""
    orc, ufd = mods['orc'], mods['ufd']
    rs = np.random.RandomState(19)
    x = rnd(rs, *shape).requires_grad_(True)
    f = rnd(rs, *fshape)
    with torch.enable_grad():
        y = orc.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=1.7)
        dy = rnd(rs, *y.shape)
        (ref,) = torch.autograd.grad(y, x, dy)
    dx = ufd.upfirdn2d_backward(dy.to(DEV), f.to(DEV), tuple(shape), up=up, down=down, padding=pad, gain=1.7)
    assert tuple(dx.shape) == tuple(shape)
    assert rel_err(c(dx), ref.numpy()) < 1e-5


def test_shu_fused_spectral_kernel_vs_two_convolutions(mods):
    """SHU spectral stage (shgan.py:320-321 conv0 + ReLU, :143-160 heterogeneous filter + band sum): the one-launch kernel
    (shg_shu_spectral_f32 -> split with one band) against the route through two 1x1 convolutions and the six-band split, and
    against the oracle; random conv0 bias so the ReLU clips differently per channel."""
    shgan, orc = mods['shgan'], mods['orc']
    torch.manual_seed(5)
    shu = shgan.SHU(32, 32, [2, 3], 'piecewise_linear', input_res=64, lowest_res=4, tail_sigma_mult=3).to(DEV).eval()
    with torch.no_grad():
        shu.conv0.bias.normal_(0, 0.02)
        shu.df1.weight.normal_(1 / 64, 0.5 / 64)
    x = torch.randn(3, 32, 64, 64, device=DEV)
    old = shgan.SHU.FUSED_SPECTRAL
    try:
        shgan.SHU.FUSED_SPECTRAL = True
        a = shu(x)
        shgan.SHU.FUSED_SPECTRAL = False
        b = shu(x)
    finally:
        shgan.SHU.FUSED_SPECTRAL = old
    ref = orc.shu_forward({'encoder.shu.' + k: v.detach().cpu() for k, v in shu.state_dict().items()}, x.cpu())
    for r in (4, 8, 16, 32, 64):
        assert not torch.equal(a[r], b[r])
        assert rel_err(c(a[r]), c(b[r])) < 2e-5, r
        assert rel_err(c(a[r]), ref[r].numpy()) < 1e-4, r
    # a changed parameter must invalidate the packed weights
    with torch.no_grad():
        shu.conv0.weight.mul_(0.5)
    a2 = shu(x)
    assert rel_err(c(a2[64]), c(a[64])) > 1e-3


def test_rccl_collectives_on_the_gpu_single_rank_group():
    """The collectives of the multi-GPU rows through RCCL itself (backend 'nccl') on the one GPU of this box: a 1-rank group
    still runs `all_gather_into_tensor` (sharded_eval's gather), `all_reduce` (FID moments, gradient buckets) as RCCL kernels on
    device buffers -- what the gloo world-2 CPU tests cannot show."""
    import os, subprocess, sys
    script = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"])
import shgan_amd
from shgan_amd import configs, eval_harness as hz
from shgan_amd.fid_stats import FidStats
from shgan_amd.grad_sync import BucketedAllReduce
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=0, world_size=1, device_id=torch.device("cuda:0"))
kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=48, w0_dim=96)
G = configs.seeded_init_(configs.build_generator(256, **kw), seed=5).eval().requires_grad_(False).to("cuda:0")
with torch.no_grad():
    order, merged = hz.sharded_eval(G, n_items=5, batch_size=2, resolution=256, seed=3, device="cuda:0", rank=0, world=1, gather=True, z_dim=48)
    ids, local = hz.sharded_eval(G, n_items=5, batch_size=2, resolution=256, seed=3, device="cuda:0", rank=0, world=1, gather=False, z_dim=48)
assert order == ids == list(range(5)) and isinstance(merged, np.ndarray) and np.array_equal(merged, local.cpu().numpy())
st = FidStats(16, device="cuda:0")
f = torch.randn(40, 16, device="cuda:0", dtype=torch.float64)
st.add(f); st.all_reduce()
cnt, mu, cov = st.mean_cov()
fc = f.cpu().numpy()
assert cnt == 40 and np.allclose(mu, fc.mean(0)) and np.allclose(cov, np.cov(fc.T, bias=True), atol=1e-10)
net = torch.nn.Linear(64, 64).to("cuda:0")
sync = BucketedAllReduce(net.parameters(), bucket_bytes=4096, always_reduce=True)
sync.zero_grad()
x = torch.randn(8, 64, device="cuda:0")
net(x).square().mean().backward()
sync.finish()
ref = torch.nn.Linear(64, 64).to("cuda:0"); ref.load_state_dict(net.state_dict())
ref(x).square().mean().backward()
assert torch.allclose(net.weight.grad, ref.weight.grad) and torch.allclose(net.bias.grad, ref.bias.grad)
dist.destroy_process_group()
print("rccl ok")
'''
    env = dict(os.environ, SHG_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), SHG_PORT=str(35500 + os.getpid() % 2000),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b'rccl ok' in p.stdout, p.stdout.decode()[-3000:]
