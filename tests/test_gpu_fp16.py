"""The fp16 route (the reference's ``use_fp16`` branches: stylegan.py:136-138,486,660-667; comodgan.py:40-47,305; native op
upfirdn2d.cpp:59) on the NHWC fp16-MFMA kernels of csrc/conv_f16.hip, against fixtures the REFERENCE produced by running those very
branches on CPU (tests/golden/fp16.npz, tools/gen_golden.py: gen_fp16).

Tolerances (stated per test): one fp16 rounding is 2^-11 = 4.9e-4 relative.  Operator outputs are compared with the fp32 evaluation on
the same half-rounded inputs (``y32``: what fp32 accumulation + one rounding should give) at 2e-3 of the output range -- FIR-resampled
forms round the intermediate once more -- and with the reference's own half result (``y16``) at 4e-3.  Networks accumulate these over
~20 layers: images / logits 2e-2 of their range, gradients 5e-2 (the reference's own fp16-vs-fp32 distance is printed beside ours)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CL = torch.channels_last


def c(a):
    return a.detach().float().cpu().numpy()


def dev_h(a):
    return torch.from_numpy(np.asarray(a)).to(DEV).to(memory_format=CL) if np.asarray(a).ndim == 4 else torch.from_numpy(np.asarray(a)).to(DEV)


@pytest.fixture(scope='module')
def g():
    import shgan_amd  # noqa: F401
    return load_golden('fp16')


def test_conv2d_resample_halves_vs_reference(g):
    from shgan_amd.model_zoo.stylegan_utils import conv2d_resample
    f4 = torch.from_numpy(g['f']).to(DEV)
    for name in g['c2r_names']:
        k = 'c2r__' + str(name) + '__'
        up, down, pad, flipw, has_f = (int(v) for v in g[k + 'cfg'])
        x, w = dev_h(g[k + 'x']), torch.from_numpy(g[k + 'w']).to(DEV)
        assert x.dtype == torch.float16 and x.is_contiguous(memory_format=CL)
        with torch.no_grad():
            y = conv2d_resample.conv2d_resample(x=x, w=w, f=f4 if has_f else None, up=up, down=down, padding=pad, flip_weight=bool(flipw))
        assert y.dtype == torch.float16 and tuple(y.shape) == tuple(g[k + 'y32'].shape), name
        e32, e16 = rel_err(c(y), g[k + 'y32']), rel_err(c(y), g[k + 'y16'].astype(np.float32))
        print(f'{name}: vs fp32-accumulate {e32:.2e}, vs reference half {e16:.2e}')
        assert e32 < 2e-3 and e16 < 4e-3, (name, e32, e16)


def test_upfirdn2d_halves_vs_reference(g):
    from shgan_amd.model_zoo.stylegan_utils import upfirdn2d
    f4 = torch.from_numpy(g['f']).to(DEV)
    for name in g['ufd_names']:
        k = 'ufd__' + str(name) + '__'
        up, down, *pad = (int(v) for v in g[k + 'cfg'])
        with torch.no_grad():
            y = upfirdn2d.upfirdn2d(dev_h(g[k + 'x']), f4, up=up, down=down, padding=pad, gain=float(g[k + 'gain']))
        assert y.dtype == torch.float16
        assert rel_err(c(y), g[k + 'y32']) < 1e-3 and rel_err(c(y), g[k + 'y16'].astype(np.float32)) < 2e-3, name


def test_same_size_fir_marching_kernel_vs_float64():
    """fir4_march_f16_kernel (same-size 4x4 filter): outer-product filters take the separable marching path, any other 4x4 filter the 16-tap
    loop of the same launch; paddings incl. crops, flipped / asymmetric filters, gains, odd extents, 8...264 channels, every strip length."""
    from shgan_amd import kernels_f16 as kf
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(7)
    f1 = torch.tensor([1., 3., 3., 1.])
    cases = [(2, 64, 37, 53, (2, 1, 2, 1), torch.outer(f1, f1) / 64, False, 1.0), (1, 8, 5, 4, (2, 1, 1, 2), torch.outer(f1, f1) / 64, False, 4.0),
             (3, 264, 19, 18, (3, 0, 0, 3), torch.outer(torch.tensor([1., 2., -1., 0.5]), torch.tensor([0.25, 1., 3., -2.])), True, 0.7),
             (1, 16, 130, 131, (1, 2, 2, 1), torch.outer(torch.tensor([1., 2., -1., 0.5]), torch.tensor([0.25, 1., 3., -2.])), False, 1.0),
             (2, 32, 40, 33, (2, 2, 2, 2), torch.randn(4, 4, generator=gen), False, 1.3), (2, 32, 9, 70, (4, 1, -1, 3), torch.randn(4, 4, generator=gen), True, 1.0),
             (8, 64, 65, 65, (2, 2, 2, 2), torch.outer(f1, f1) / 64, False, 1.0), (1, 128, 257, 129, (2, 1, 2, 1), torch.outer(f1, f1) / 16, True, 1.0)]
    for n, ch, h, w, (px0, px1, py0, py1), f, flip, gain in cases:
        x = torch.randn(n, ch, h, w, generator=gen).half()
        y = kf.upfirdn2d(x.to(DEV).to(memory_format=CL), f.to(DEV), padx0=px0, padx1=px1, pady0=py0, pady1=py1, flip=flip, gain=gain)
        xd = x.double()
        xd = F.pad(xd, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
        xd = xd[:, :, max(-py0, 0): xd.shape[2] - max(-py1, 0), max(-px0, 0): xd.shape[3] - max(-px1, 0)]
        fk = (f if flip else f.flip([0, 1])).double() * gain
        ref = F.conv2d(xd, fk[None, None].repeat(ch, 1, 1, 1), groups=ch)
        assert tuple(y.shape) == tuple(ref.shape) and y.dtype == torch.float16
        e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        assert e < 6e-4, ((n, ch, h, w), e)                       # one half rounding of an fp32-accumulated sum


def test_up2_down2_fir_kernels_vs_oracle():
    """updn4_f16_kernel<1,2> / <2,1> (4x4 filter, down = 2 or up = 2: the skip branches of the half blocks and their gradients) against the CPU
    oracle's upfirdn2d on the same half inputs: odd extents, every padding incl. crops, flipped asymmetric filters, gains."""
    from oracle import shgan_oracle as orc
    from shgan_amd import kernels_f16 as kf
    gen = torch.Generator().manual_seed(11)
    f1 = torch.tensor([1., 3., 3., 1.])
    fa = torch.outer(torch.tensor([1., 2., -1., 0.5]), torch.tensor([0.25, 1., 3., -2.]))
    for n, ch, h, w, up, down, pad, f, flip, gain in [
            (2, 64, 32, 32, 1, 2, (1, 1, 1, 1), torch.outer(f1, f1) / 64, False, 1.0), (2, 64, 16, 16, 2, 1, (2, 1, 2, 1), torch.outer(f1, f1) / 64, False, 4.0),
            (1, 8, 7, 9, 1, 2, (2, 0, 1, 3), fa, True, 0.6), (1, 8, 7, 9, 2, 1, (3, 2, 0, 1), fa, True, 1.7), (3, 264, 13, 6, 1, 2, (0, 3, 3, 0), fa, False, 1.0),
            (3, 264, 5, 11, 2, 1, (1, 1, 2, 2), torch.randn(4, 4, generator=gen), False, 1.0), (2, 16, 33, 20, 1, 2, (1, -1, -1, 2), fa, False, 1.0),
            (2, 16, 12, 17, 2, 1, (-1, 2, 3, -2), torch.randn(4, 4, generator=gen), True, 2.0)]:
        x = torch.randn(n, ch, h, w, generator=gen).half()
        ref = orc.upfirdn2d(x.float(), f, up=up, down=down, padding=list(pad), flip_filter=flip, gain=gain)
        y = kf.upfirdn2d(x.to(DEV).to(memory_format=CL), f.to(DEV), up, up, down, down, pad[0], pad[1], pad[2], pad[3], flip, gain)
        assert tuple(y.shape) == tuple(ref.shape) and y.dtype == torch.float16, (up, down, pad)
        e = float((y.cpu().float() - ref).abs().max() / ref.abs().max())
        assert e < 6e-4, ((n, ch, h, w, up, down, pad), e)


def test_modulated_conv2d_halves_with_prenormalisation_vs_reference(g):
    from shgan_amd.model_zoo import stylegan
    f4 = torch.from_numpy(g['f']).to(DEV)
    for name in g['mc_names']:
        k = 'mc__' + str(name) + '__'
        up, demod = (int(v) for v in g[k + 'cfg'])
        w = torch.from_numpy(g[k + 'w']).to(DEV)
        noise = torch.from_numpy(g[k + 'noise']).to(DEV) if (k + 'noise') in g.files else None
        with torch.no_grad():
            y = stylegan.modulated_conv2d(x=dev_h(g[k + 'x']), weight=w, styles=torch.from_numpy(g[k + 'styles']).to(DEV), noise=noise, up=up,
                                          padding=w.shape[2] // 2, resample_filter=f4 if up > 1 else None, demodulate=bool(demod),
                                          flip_weight=(up == 1), fused_modconv=False)
        assert y.dtype == torch.float16
        e32, e16 = rel_err(c(y), g[k + 'y32']), rel_err(c(y), g[k + 'y16'].astype(np.float32))
        print(f'modconv {name}: vs fp32 {e32:.2e}, vs reference half {e16:.2e}')
        assert e32 < 4e-3 and e16 < 6e-3, (name, e32, e16)


def test_bias_act_halves(g):
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    x = dev_h(g['act__x'])
    y = grad_ops.bias_act(x, None, act=True, gain=float(np.sqrt(0.5)), alpha=0.2, act_gain=float(np.sqrt(2)), clamp=256.0)
    assert rel_err(c(y), g['act__y'].astype(np.float32)) < 1e-3
    # with a bias, and the gradient from the saved output (sign / clamp state)
    b = torch.linspace(-1, 1, 16, device=DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        y = grad_ops.bias_act(xr, b, act=True, gain=1.0, alpha=0.2, act_gain=float(np.sqrt(2)), clamp=256.0)
        y.float().sum().backward()
    with torch.enable_grad():
        xf = x.float().cpu().requires_grad_(True)
        bf = b.detach().cpu().requires_grad_(True)
        yf = (F.leaky_relu(xf + bf.view(1, -1, 1, 1), 0.2) * np.sqrt(2)).clamp(-256, 256)
        yf.sum().backward()
    assert rel_err(c(y), yf.detach().numpy()) < 1e-3
    # The backward reads the clamp state from the saved OUTPUT: a value that reaches +-256 only by the fp16 rounding of an unclamped
    # 255.9 (a sliver of one half-precision spacing below the clamp) is indistinguishable from a clamped one and gets slope 0 where
    # autograd of the float32 chain passes the gradient -- allowed on those elements only.
    gd, gr, yo = c(xr.grad), xf.grad.numpy(), np.abs(c(y))
    bad = np.abs(gd - gr) > 1e-3 * np.abs(gr).max()
    assert bad.sum() <= 4 and np.all(yo[bad] == 256.0), (int(bad.sum()), yo[bad])
    db = (gr * ~bad).sum((0, 2, 3))
    assert np.abs(c(b.grad) - db).max() < 3e-3 * np.abs(db).max()


CONV_BWD = [(2, 32, 48, 20, 24, 3, 1, 1), (1, 64, 160, 16, 16, 3, 1, 1), (2, 16, 24, 33, 33, 3, 2, 0), (2, 32, 32, 16, 18, 3, 2, 1),
            (2, 48, 40, 9, 17, 1, 1, 0), (2, 4, 32, 16, 16, 1, 1, 0), (2, 32, 3, 16, 16, 1, 1, 0), (3, 24, 40, 7, 5, 3, 1, 1)]


@pytest.mark.parametrize('n,ci,co,h,w,k,stride,pad', CONV_BWD)
def test_conv2d_halves_forward_backward_vs_float64_autograd(n, ci, co, h, w, k, stride, pad):
    """fp16-MFMA convolution, its input gradient (the opposite operator) and the fp16 weight-gradient kernel against float64 autograd
    of F.conv2d on the same half-rounded operands: 2e-3 of each result's range (fp32 accumulation, one rounding)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    rs = np.random.RandomState(n + ci + co + h + k + stride)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32)).half()
    wt = torch.from_numpy((rs.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)).half()
    with torch.enable_grad():
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        yr = F.conv2d(xr, wr, stride=stride, padding=pad)
        gy = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32)).half()
        yr.backward(gy.double())
    xd, wd = x.to(DEV).to(memory_format=CL).requires_grad_(), wt.to(DEV).requires_grad_()
    with torch.enable_grad():
        y = conv2d_gradfix.conv2d(xd, wd, stride=stride, padding=pad)
        y.backward(gy.to(DEV).to(memory_format=CL))
    assert y.dtype == torch.float16 and xd.grad.dtype == torch.float16 and wd.grad.dtype == torch.float16
    assert rel_err(c(y), yr.detach().numpy()) < 2e-3
    assert rel_err(c(xd.grad), xr.grad.numpy()) < 2e-3
    assert rel_err(c(wd.grad), wr.grad.numpy()) < 2e-3


@pytest.mark.parametrize('n,ci,co,h,w,pad', [(2, 32, 24, 8, 8, 0), (1, 48, 64, 9, 20, 0), (2, 16, 16, 16, 16, 1)])
def test_conv_transpose2d_halves_forward_backward_vs_float64_autograd(n, ci, co, h, w, pad):
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    rs = np.random.RandomState(n + ci + co + h + pad)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32)).half()
    wt = torch.from_numpy((rs.standard_normal((ci, co, 3, 3)) / np.sqrt(ci * 9)).astype(np.float32)).half()
    with torch.enable_grad():
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        yr = F.conv_transpose2d(xr, wr, stride=2, padding=pad)
        gy = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32)).half()
        yr.backward(gy.double())
    xd, wd = x.to(DEV).to(memory_format=CL).requires_grad_(), wt.to(DEV).requires_grad_()
    with torch.enable_grad():
        y = conv2d_gradfix.conv_transpose2d(xd, wd, stride=2, padding=pad)
        y.backward(gy.to(DEV).to(memory_format=CL))
    assert rel_err(c(y), yr.detach().numpy()) < 2e-3
    assert rel_err(c(xd.grad), xr.grad.numpy()) < 2e-3
    assert rel_err(c(wd.grad), wr.grad.numpy()) < 2e-3


def test_discriminator_fp16_blocks_vs_reference(g):
    """stylegan2_discriminator with use_fp16_before_res = 16 at R = 64 (blocks 64 and 32 in half): logits, loss, every parameter gradient
    and the gradient of the input image against the reference's fp16 run."""
    from shgan_amd.model_zoo import stylegan
    from oracle import shgan_oracle as orc
    D = stylegan.Discriminator(resolution=64, ic_n=4, ch_base=1024, ch_max=32, use_fp16_before_res=16, mbstd_group_size=4, mbstd_c_n=1)
    assert D.b64.use_fp16 and D.b32.use_fp16 and not D.b16.use_fp16
    orc.seeded_fill_(D, seed=71, bias_std=0.1)
    D = D.to(DEV).train().requires_grad_(True)
    img = torch.from_numpy(np.random.RandomState(72).standard_normal((4, 4, 64, 64)).astype(np.float32)).to(DEV).requires_grad_(True)
    with torch.enable_grad():
        logits = D(img, None)
        loss = F.softplus(logits).mean()
        loss.backward()
    ref_gap = rel_err(g['D__logits'], g['D__logits_fp32'])
    e = rel_err(c(logits), g['D__logits'])
    print(f'D logits: ours vs reference fp16 {e:.2e}; reference fp16 vs its fp32 {ref_gap:.2e}')
    assert e < 2e-2 and abs(float(loss) - float(g['D__loss'])) < 1e-2 * abs(float(g['D__loss']))
    assert rel_err(c(img.grad)[:, :, ::2, ::2], g['D__grad_img']) < 5e-2
    errs = {n_: rel_err(c(p.grad), g['D__grad__' + n_]) for n_, p in D.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print('D parameter gradients, largest errors:', [(k, float('%.2e' % v)) for k, v in top])
    assert all(p.grad.dtype == torch.float32 for p in D.parameters())
    assert max(errs.values()) < 5e-2, top
    with torch.no_grad():                                   # the same modules without autograd: bias + activation fused into the fp16 convolutions
        lg = D(img.detach(), None)
    assert rel_err(c(lg), g['D__logits']) < 2e-2 and rel_err(c(lg), c(logits)) < 5e-3


def test_generator_fp16_blocks_vs_reference(g):
    """SH-GAN generator with fp16 encoder blocks (256, 128) and fp16 synthesis blocks (64, 128, 256): eval and train-mode images and the
    parameter gradients of sum(img * r) / N against the reference's fp16 run (reduced width, noise_mode='const')."""
    from shgan_amd import configs, eval_harness
    from oracle import shgan_oracle as orc
    kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = configs.build_generator(256, use_fp16_before_res=64, use_fp16_after_res=32, **kw)
    G.load_state_dict(orc.init_state_dict(256, seed=int(g['G__seed']), noise_strength=0.1, bias_std=0.1, **kw), strict=True)
    assert G.encoder.b256.use_fp16 and G.encoder.b128.use_fp16 and not G.encoder.b64.use_fp16
    assert G.synthesis.b256.use_fp16 and G.synthesis.b64.use_fp16 and not G.synthesis.b32.use_fp16
    G = G.to(DEV)
    for m in G.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    rs = np.random.RandomState(74)
    real_u8 = rs.randint(0, 256, size=(2, 3, 256, 256)).astype(np.uint8)
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    mask = torch.from_numpy(np.unpackbits(g['G__mask_bits'])[: 2 * 256 * 256].reshape(2, 1, 256, 256).astype(np.float32))
    x = eval_harness.assemble_input(real, mask).to(DEV)
    z = torch.from_numpy(g['G__z']).to(DEV)
    cnd = torch.zeros(2, 0, device=DEV)
    ref_gap = rel_err(g['G__img_eval'], g['G__img_fp32'])
    with torch.no_grad():
        img = G.eval()(x=x, z=z, c=cnd, noise_mode='const')
    assert img.dtype == torch.float32 and img.is_contiguous()
    e = rel_err(c(img)[:, :, ::4, ::4], g['G__img_eval'])
    print(f'G eval image: ours vs reference fp16 {e:.2e}; reference fp16 vs its fp32 {ref_gap:.2e}')
    assert e < 2e-2
    G = G.train().requires_grad_(True)
    r = torch.from_numpy(np.random.RandomState(75).standard_normal((2, 3, 256, 256)).astype(np.float32)).to(DEV)
    with torch.enable_grad():
        img = G(x=x, z=z, c=cnd, noise_mode='const')
        ((img * r).sum() / 2).backward()
    assert rel_err(c(img)[:, :, ::4, ::4], g['G__img_train']) < 2e-2
    # Gradients: the reference's own fp16 gradients are noisy -- broadcast reductions (noise, bias, style gradients) and ~1e4-term
    # cancellations are carried out in half on both sides, in different orders -- so the yardstick is the reference's FLOAT32 gradient
    # (G__grad32__*): the MEDIAN distance to it over all parameters may be at most 1.5 x the reference-fp16 median (measured: 1.8e-2 against
    # 1.5e-2), and single parameters -- heavy-tailed on both sides: 0.25 / 0.41 at worst -- at most max(10 x the reference's own, 5e-2).
    errs, ref_errs = {}, {}
    for n_, p in G.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, n_
        r32, r16 = g['G__grad32__' + n_], g['G__grad__' + n_]
        gn = p.grad.reshape(-1)
        got = c(gn) if gn.numel() <= 512 else np.concatenate([c(gn[:256]), c(gn[-256:])])
        if float(np.abs(r32).max()) > 0:
            errs[n_], ref_errs[n_] = rel_err(got, r32), rel_err(r16, r32)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    med, med_ref = float(np.median(list(errs.values()))), float(np.median(list(ref_errs.values())))
    print(f'G parameter gradients vs reference float32: ours median {med:.2e} max {max(errs.values()):.2e} | reference fp16 median {med_ref:.2e} '
          f'max {max(ref_errs.values()):.2e}; largest ours', [(k, float('%.2e' % v), float('%.2e' % ref_errs[k])) for k, v in top])
    for n_, e_ in errs.items():
        assert e_ <= max(10 * ref_errs[n_], 5e-2), (n_, e_, ref_errs[n_])
    assert med <= 1.5 * med_ref + 1e-4, (med, med_ref)


def test_fp16_mfma_keeps_denormal_operands():
    """Half-precision gradients without loss scaling live in the denormal range (< 6.1e-5): the convolution must not flush them."""
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels_f16
    x = torch.full((1, 32, 8, 8), 3e-6, device=DEV).half().to(memory_format=CL)            # 3e-6 is a denormal half (50 ulps of 6e-8)
    assert float(x.float().max()) > 0
    w = torch.ones(32, 32, 1, 1, device=DEV).half()
    y = kernels_f16.conv2d(x, w)
    want = 32 * float(x.float()[0, 0, 0, 0])
    assert abs(float(y.float().mean()) - want) < 0.02 * want, (float(y.float().mean()), want)
    ws = torch.full((32, 32, 1, 1), 2e-6, device=DEV).half()                                 # denormal weights, normal activations
    y2 = kernels_f16.conv2d(torch.ones_like(x), ws)
    want2 = 32 * float(ws.float()[0, 0, 0, 0])
    assert abs(float(y2.float().mean()) - want2) < 0.02 * want2, (float(y2.float().mean()), want2)


@pytest.mark.parametrize('c,shared_noise', [(64, True), (128, False), (512, True), (8, False)])
def test_fused_modulation_tail_forward_backward(c, shared_noise):
    """grad_ops.modconv_tail: y = lrelu_agc(t*d + noise + bias) in one pass, its one-pass backward (gt, d/dd, d/dnoise, d/dbias) against
    float32 autograd of the same expression on the half-rounded operands, and the composed create_graph form against the fast one."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    rs = np.random.RandomState(c)
    n, h, w = 3, 12, 20
    t = torch.from_numpy(rs.standard_normal((n, c, h, w)).astype(np.float32) * 40).half()
    d = torch.from_numpy((rs.rand(n, c) + 0.5).astype(np.float32))
    nz = torch.from_numpy(rs.standard_normal((h, w) if shared_noise else (n, 1, h, w)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(c).astype(np.float32))
    gy = torch.from_numpy(rs.standard_normal((n, c, h, w)).astype(np.float32)).half()
    with torch.enable_grad():
        tr, dr, nr, br = (v.float().clone().requires_grad_(True) for v in (t, d, nz, b))
        yr = (F.leaky_relu(tr * dr.reshape(n, c, 1, 1) + nr + br.reshape(1, c, 1, 1), 0.2) * np.sqrt(2)).clamp(-256, 256)
        yr.backward(gy.float())
    td, dd_, nd, bd = (v.to(DEV).requires_grad_(True) for v in (t.to(memory_format=CL), d, nz, b))
    with torch.enable_grad():
        y = grad_ops.modconv_tail(td, d=dd_, noise=nd, bias=bd, act=True, gain=1.0, alpha=0.2, act_gain=float(np.sqrt(2)), clamp=256.0)
        y.backward(gy.to(DEV).to(memory_format=CL))
    assert rel_err(c_(y), yr.detach().numpy()) < 1e-3
    # (elements whose unclamped value rounds onto the clamp are the sliver described in test_bias_act_halves)
    bad = np.abs(c_(td.grad) - tr.grad.numpy()) > 2e-3 * np.abs(tr.grad.numpy()).max()
    assert bad.mean() < 2e-3
    assert rel_err(c_(dd_.grad), dr.grad.numpy()) < 5e-3 and rel_err(c_(nd.grad), nr.grad.numpy()) < 5e-3 and rel_err(c_(bd.grad), br.grad.numpy()) < 5e-3
    # create_graph: the composed backward must give the same first derivatives and be differentiable again
    td2, dd2 = t.to(DEV).to(memory_format=CL).requires_grad_(True), d.to(DEV).requires_grad_(True)
    with torch.enable_grad():
        y2 = grad_ops.modconv_tail(td2, d=dd2, noise=nd.detach(), bias=bd.detach(), act=True, gain=1.0, alpha=0.2, act_gain=float(np.sqrt(2)), clamp=256.0)
        (g1, g2) = torch.autograd.grad([(y2.float() * gy.to(DEV).float()).sum()], [td2, dd2], create_graph=True)
        assert rel_err(c_(g1), c_(td.grad)) < 2e-3 and rel_err(c_(g2), c_(dd_.grad)) < 5e-3
        g1.float().square().sum().backward()             # d/dd of |gz * d|^2 exists (gz is constant in t)
    assert dd2.grad is not None and torch.isfinite(dd2.grad).all() and float(dd2.grad.abs().max()) > 0


def c_(a):
    return a.detach().float().cpu().numpy()


def test_fused_inference_route_of_half_layers_matches_the_composed_route(g):
    """Half layers without autograd fuse ``x * styles``, demodulation, noise, bias, lrelu_agc and the skip-add into the fp16 convolution
    (stylegan.F16_INFER_FUSED).  Same generator, both routes: they differ only by where intermediate values are rounded to half."""
    from shgan_amd import configs, eval_harness
    from shgan_amd.model_zoo import stylegan
    from oracle import shgan_oracle as orc
    kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = configs.build_generator(256, use_fp16_before_res=64, use_fp16_after_res=32, **kw)
    G.load_state_dict(orc.init_state_dict(256, seed=int(g['G__seed']), noise_strength=0.1, bias_std=0.1, **kw), strict=True)
    G = G.to(DEV).eval().requires_grad_(False)
    x, z, _, _ = eval_harness.synthetic_batch(3, 256, 64, seed=11, device=DEV, masks='bernoulli')
    cnd = torch.zeros(3, 0, device=DEV)
    outs = {}
    for fused in (True, False):
        stylegan.F16_INFER_FUSED = fused
        try:
            with torch.no_grad():
                torch.manual_seed(5)
                outs[fused] = (G(x=x, z=z, c=cnd, noise_mode='const'), G(x=x, z=z, c=cnd, noise_mode='random'))
        finally:
            stylegan.F16_INFER_FUSED = True
    assert not torch.equal(outs[True][0], outs[False][0])                     # two different routes really ran
    assert rel_err(c(outs[True][0]), c(outs[False][0])) < 3e-3
    assert torch.isfinite(outs[True][1]).all() and outs[True][1].dtype == torch.float32


def test_randomised_shapes_of_the_fp16_entry_points():
    """tools/fuzz_f16.py: 150 random geometries (ragged extents, thin / odd channel counts, crops and zero-extensions of the transposed form,
    the fused tail, FIR factors and paddings, weight gradients, the modulation tail) against float64 on the same half operands: 3e-3 bar,
    4e-4 ... 5e-4 measured (one half rounding)."""
    import os, subprocess, sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_f16.py'), '150', '7'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


def test_fp16_blocks_full_width_512_batch16_properties():
    """The reference's use_fp16 option at BASELINE size (FFHQ-512, batch 16, full width) -- no fp16 fixture exists at this size (CPU hours), so
    size-independent properties: the fp16-block generator stays within half precision of the float32 one on the same weights and inputs
    (2e-2 of the image range, measured 1.0e-3), is bit-identical run to run, shard-invariant (a 2-image shard = rows of the full batch within one
    half rounding: the batch-global style norm), and the known pixels of the uint8 composite are exact."""
    from shgan_amd import configs, eval_harness
    G32 = configs.seeded_init_(configs.build_generator(512), seed=91, noise_strength=0.05).eval().requires_grad_(False).to(DEV)
    G16 = configs.build_generator(512, use_fp16_before_res=64, use_fp16_after_res=32)
    G16.load_state_dict(G32.state_dict(), strict=True)
    G16 = G16.eval().requires_grad_(False).to(DEV)
    x, z, real_u8, mask = eval_harness.synthetic_batch(16, 512, 512, seed=92, device=DEV, masks='bernoulli')
    cnd = torch.zeros(16, 0, device=DEV)
    with torch.no_grad():
        a = G16(x=x, z=z, c=cnd, noise_mode='const')
        b = G16(x=x, z=z, c=cnd, noise_mode='const')
        ref = G32(x=x, z=z, c=cnd, noise_mode='const')
        sub = G16(x=x[:2], z=z[:2], c=cnd[:2], noise_mode='const')
        u8 = eval_harness.run_generator(G16, x, z, noise_mode='const')
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.dtype == torch.float32
    e = rel_err(c(a), c(ref))
    print(f'fp16 blocks vs float32, 512 x 16 full width: {e:.2e}')
    assert e < 2e-2
    assert rel_err(c(sub), c(a[:2])) < 5e-3
    m = mask.astype(bool)
    assert np.array_equal(np.where(m, u8.cpu().numpy(), 0), np.where(m, real_u8, 0))


@pytest.mark.parametrize('shape', [(8, 512, 64, 64), (3, 64, 33, 17), (2, 72, 8, 8), (1, 8, 1, 1), (4, 128, 256, 256), (8, 4, 512, 512), (3, 3, 17, 9), (2, 1, 5, 5)])
def test_block_boundary_cast_kernel_is_the_torch_cast(shape):
    """`x.to(dtype)` at the block boundaries (stylegan.py:486-495,659-663) as the transposing relayout kernel: bit-identical to torch's
    `.to(dtype, memory_format)` in both directions (round-to-nearest-even, overflow to inf), the result in the layout of its dtype, and
    its gradient the opposite cast."""
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels_f16
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 300).to('cuda:0')
    x.view(-1)[::7] *= 1e3                                   # some values beyond the half range
    h = kernels_f16.relayout(x)
    ref = x.to(dtype=torch.float16, memory_format=torch.channels_last)
    assert h.dtype == torch.float16 and h.is_contiguous(memory_format=torch.channels_last) and torch.equal(h, ref)
    back = kernels_f16.relayout(ref)
    assert back.dtype == torch.float32 and back.is_contiguous() and torch.equal(back, ref.to(torch.float32))
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        y = grad_ops.to_block_dtype(xr, True)
        assert y.dtype == torch.float16 and y.grad_fn is not None and 'Relayout' in type(y.grad_fn).__name__
        w = torch.randn(shape, generator=g).to('cuda:0').to(dtype=torch.float16, memory_format=torch.channels_last)
        (gx,) = torch.autograd.grad((y * w).sum(), [xr])
        assert gx.dtype == torch.float32 and torch.equal(gx, w.to(torch.float32))
        z = grad_ops.to_block_dtype(y, False)
        assert z.dtype == torch.float32 and torch.equal(z, ref.to(torch.float32))


@pytest.mark.parametrize('o,i,k', [(64, 64, 3), (128, 40, 3), (8, 72, 1), (96, 512, 3)])
def test_direct_weight_pack_equals_the_staged_pack(o, i, k):
    """``pack_weight`` straight from the torch-layout tensor (one gather kernel; transposed / rotated forms for the transposed convolution
    and the input gradient) against the staged form (torch flip + permute-copy, then the [T][O][I] pack kernel): identical operand tensors."""
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels_f16
    g = torch.Generator(device='cpu').manual_seed(o + i + k)
    w = torch.randn(o, i, k, k, generator=g).to('cuda:0', torch.float16)
    for tr in (False, True):
        for fl in (False, True):
            src = w.transpose(0, 1).contiguous() if tr else w          # the [Cin, Cout, k, k] layout when transposed
            kernels_f16.PACK_DIRECT = True
            a = kernels_f16.pack_weight(src, transposed=tr, flip=fl)
            kernels_f16.PACK_DIRECT = False
            try:
                b = kernels_f16.pack_weight(src, transposed=tr, flip=fl)
            finally:
                kernels_f16.PACK_DIRECT = True
            assert (a.o, a.i, a.k) == (b.o, b.i, b.k) == (o, i, k) and torch.equal(a.wp, b.wp), (tr, fl)
    # and the input gradient of a half convolution through it: against torch autograd in float32
    x = torch.randn(2, i, 12, 10, generator=g).to('cuda:0')
    gy = torch.randn(2, o, 12 if k == 3 else 12, 10, generator=g).to('cuda:0')
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        y = torch.nn.functional.conv2d(xr, w.float(), padding=k // 2)
        (want,) = torch.autograd.grad(y, [xr], [gy])
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    got = conv2d_gradfix._conv_input_grad(gy.to(dtype=torch.float16, memory_format=torch.channels_last), w, x.shape, 1, k // 2)
    assert float((got.float() - want).abs().max()) <= 2e-2 * float(want.abs().max())


# ---- the reference's FUSED modulated form on halves (fused_modconv with float16: what its blocks take in eval for ONE image) ---------------

@pytest.fixture(scope='module')
def gf():
    import shgan_amd  # noqa: F401
    return load_golden('fp16_fused')


def test_fused_modulated_conv2d_halves_vs_reference(gf):
    """``modulated_conv2d(..., fused_modconv=True)`` on halves (stylegan.py:149-170,183-193: per-sample weights w * s * d rounded to half
    once, grouped convolution with groups = N) against the reference's own half run of that branch, N = 1, 2, 3, plain / up / toRGB
    (tests/golden/fp16_fused.npz, tools/gen_golden.py: gen_fp16_fused).  Same bars as the non-fused operator test above: 4e-3 of the output
    range to the fp32 evaluation on the same half inputs, 6e-3 to the reference's half result."""
    from shgan_amd.model_zoo import stylegan
    f4 = torch.from_numpy(gf['f']).to(DEV)
    for name in gf['mc_names']:
        k = 'mc__' + str(name) + '__'
        up, demod = (int(v) for v in gf[k + 'cfg'])
        w = torch.from_numpy(gf[k + 'w']).to(DEV)
        noise = torch.from_numpy(gf[k + 'noise']).to(DEV) if (k + 'noise') in gf.files else None
        kw = dict(weight=w, styles=torch.from_numpy(gf[k + 'styles']).to(DEV), noise=noise, up=up, padding=w.shape[2] // 2,
                  resample_filter=f4 if up > 1 else None, demodulate=bool(demod), flip_weight=(up == 1))
        with torch.no_grad():
            y = stylegan.modulated_conv2d(x=dev_h(gf[k + 'x']), fused_modconv=True, **kw)
            y_nf = stylegan.modulated_conv2d(x=dev_h(gf[k + 'x']), fused_modconv=False, **kw)
        assert y.dtype == torch.float16 and tuple(y.shape) == tuple(gf[k + 'y16'].shape)
        e32, e16 = rel_err(c(y), gf[k + 'y32']), rel_err(c(y), gf[k + 'y16'].astype(np.float32))
        print(f'fused modconv {name}: vs fp32 {e32:.2e}, vs reference half {e16:.2e}; non-fused form vs fp32 {rel_err(c(y_nf), gf[k + "y32"]):.2e}')
        assert e32 < 4e-3 and e16 < 6e-3, (name, e32, e16)
        assert not torch.equal(y, y_nf), name                      # the flag really selects another rounding point


@pytest.mark.parametrize('up', [1, 2])
def test_grouped_half_convolutions_are_one_launch_per_group(up):
    """conv2d_resample(groups = N) on halves (the reshape the reference's fused form performs, stylegan.py:187-190) == the per-group
    float32 convolutions of torch on the same half operands, within one half rounding of the result (+ one for the FIR)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import conv2d_resample, upfirdn2d
    rs = np.random.RandomState(7 + up)
    n, i, o, h, w = 3, 16, 24, 10, 12
    x = torch.from_numpy(rs.standard_normal((1, n * i, h, w)).astype(np.float32)).half().to(DEV).to(memory_format=CL)
    wt = torch.from_numpy((rs.standard_normal((n * o, i, 3, 3)) / np.sqrt(9 * i)).astype(np.float32)).half().to(DEV)
    f4 = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    with torch.no_grad():
        y = conv2d_resample.conv2d_resample(x=x, w=wt, f=f4 if up > 1 else None, up=up, padding=1, groups=n, flip_weight=(up == 1))
    assert y.dtype == torch.float16 and y.shape[1] == n * o
    refs = []
    for j in range(n):
        xj, wj = x[:, j * i:(j + 1) * i].double(), wt[j * o:(j + 1) * o].double()
        if up == 1:
            refs.append(F.conv2d(xj, wj, padding=1))
        else:
            mid = F.conv_transpose2d(xj, wj.transpose(0, 1), stride=2)          # flip_weight False: true convolution with the un-flipped weight
            f2 = (f4.double() * 4.0)[None, None].repeat(o, 1, 1, 1)
            refs.append(F.conv2d(F.pad(mid, [1, 1, 1, 1]), f2.flip([2, 3]), groups=o))
    ref = torch.cat(refs, 1)
    assert tuple(ref.shape) == tuple(y.shape)
    assert rel_err(c(y), c(ref)) < (1e-3 if up == 1 else 2e-3)


def test_generator_fp16_blocks_batch_of_one_takes_the_fused_form_vs_reference(gf):
    """In eval a batch of ONE image makes every half block of the reference take the fused form (stylegan.py:490, comodgan.py:309): the
    product follows the same rule (``stylegan.fused_modconv_rule``) -- checked on the SH-GAN generator with fp16 blocks against the
    reference's own run (reduced width, noise_mode='const'), image at 2e-2 of its range like the batch-of-two fixture."""
    from shgan_amd import configs, eval_harness
    from shgan_amd.model_zoo import stylegan
    from oracle import shgan_oracle as orc
    kw = dict(ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = configs.build_generator(256, use_fp16_before_res=64, use_fp16_after_res=32, **kw)
    G.load_state_dict(orc.init_state_dict(256, seed=int(gf['G__seed']), noise_strength=0.1, bias_std=0.1, **kw), strict=True)
    G = G.to(DEV).eval().requires_grad_(False)
    real_u8 = np.random.RandomState(84).randint(0, 256, size=(1, 3, 256, 256)).astype(np.uint8)
    real = torch.from_numpy(real_u8.astype(np.float32)) / 127.5 - 1.0
    mask = torch.from_numpy(np.unpackbits(gf['G__mask_bits'])[: 256 * 256].reshape(1, 1, 256, 256).astype(np.float32))
    x = eval_harness.assemble_input(real, mask).to(DEV)
    z = torch.from_numpy(gf['G__z']).to(DEV)
    seen = []
    orig = stylegan._modulated_conv2d_half_infer

    def spy(*a, **k):
        seen.append(bool(k.get('fused')))
        return orig(*a, **k)
    stylegan._modulated_conv2d_half_infer = spy
    try:
        with torch.no_grad():
            img = G(x=x, z=z, c=torch.zeros(1, 0, device=DEV), noise_mode='const')
            seen_one, seen[:] = list(seen), []
            x2 = torch.cat([x, x])
            img2 = G(x=x2, z=torch.cat([z, z]), c=torch.zeros(2, 0, device=DEV), noise_mode='const')
            seen_two = list(seen)
    finally:
        stylegan._modulated_conv2d_half_infer = orig
    # six half 3x3 layers (synthesis 64 / 128 / 256: conv0 + conv1) -- with the three toRGB layers the reference's nine half calls
    assert len(seen_one) == 6 and all(seen_one) and int(gf['G__calls_half']) == 9
    assert len(seen_two) == 6 and not any(seen_two)                       # a batch of two: the non-fused algebra, as in the reference
    e_f, e_nf = rel_err(c(img)[:, :, ::2, ::2], gf['G__img_eval']), rel_err(c(img)[:, :, ::2, ::2], gf['G__img_eval_nonfused'])
    ref_gap, ref_forms = rel_err(gf['G__img_eval'], gf['G__img_fp32']), rel_err(gf['G__img_eval'], gf['G__img_eval_nonfused'])
    print(f'G eval, one image: ours vs reference fused {e_f:.2e} (vs its non-fused run {e_nf:.2e}); reference fused vs its fp32 {ref_gap:.2e}, '
          f'its two forms apart {ref_forms:.2e}')
    assert img.dtype == torch.float32 and e_f < 2e-2
    assert rel_err(c(img2[:1])[:, :, ::2, ::2], gf['G__img_fp32']) < 2e-2
