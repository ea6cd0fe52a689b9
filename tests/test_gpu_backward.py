"""N3 (training step) building blocks on the GPU: convolution backward -- input gradient through the opposite operator, weight
gradient on shg_conv2d_wgrad_f32 -- against torch CPU autograd in float64 (conv2d_gradfix.py:107-165 is the reference's form;
its cuDNN weight-gradient call has no CPU path, so torch's own autograd of F.conv2d / F.conv_transpose2d is the oracle)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def gf():
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    return dict(gf=conv2d_gradfix, kernels=kernels)


def c(a):
    return a.detach().cpu().numpy()


CONV_CASES = [
    # n, ci, co, h, w, k, stride, pad: ragged channels (I, O % 32), rows shorter / longer than the 64-pixel chunk, both strides
    (2, 8, 16, 16, 16, 3, 1, 1), (3, 37, 70, 20, 24, 3, 1, 1), (1, 64, 64, 64, 96, 3, 1, 1), (2, 24, 40, 12, 12, 3, 1, 0),
    (2, 16, 32, 33, 33, 3, 2, 0), (1, 40, 24, 65, 129, 3, 2, 0), (2, 12, 20, 16, 16, 3, 2, 1),
    (2, 20, 36, 16, 16, 1, 1, 0), (1, 4, 64, 32, 32, 1, 1, 0),
    # small images: several output rows per chunk (packed form), incl. a partial last chunk and rows of several images in one chunk
    (3, 24, 70, 4, 4, 3, 1, 1), (5, 64, 64, 8, 8, 3, 1, 1), (3, 40, 33, 32, 32, 3, 1, 1), (7, 16, 16, 9, 9, 3, 2, 0), (3, 70, 24, 17, 17, 3, 2, 0),
    (2, 32, 32, 33, 33, 3, 2, 0), (3, 16, 24, 8, 8, 3, 2, 1),
    # 32-pixel output rows: two rows per chunk of the weight-gradient kernel (pad 1 and pad 0), and an odd row count (one-row chunks)
    (2, 24, 40, 34, 34, 3, 1, 0), (3, 70, 64, 32, 32, 3, 1, 1), (2, 16, 24, 33, 32, 3, 1, 1),
    # the unmasked form of the weight gradient (round 6): channels in whole 64-blocks, rows in whole chunks, no padding -- the FIR-padded
    # stride-2 layers (65 -> 32, 129 -> 64 with two chunks per row, non-square) and 1x1 layers; one row more than the valid extent needs
    (2, 64, 128, 65, 65, 3, 2, 0), (1, 128, 64, 33, 129, 3, 2, 0), (2, 64, 64, 66, 66, 3, 2, 0), (2, 64, 128, 16, 64, 1, 1, 0), (1, 128, 64, 8, 128, 1, 1, 0),
    # ... its narrow stride-2 forms (two / four whole output rows per chunk: 33 -> 16, 17 -> 8; three images so that a slice boundary falls inside
    # an image) and 1x1 layers whose rows are shorter than a chunk (an image = one row of H W pixels: 8 x 8, 16 x 32, 4 x 16)
    (3, 64, 128, 33, 33, 3, 2, 0), (3, 128, 64, 17, 17, 3, 2, 0), (5, 64, 64, 33, 33, 3, 2, 0), (2, 64, 128, 8, 8, 1, 1, 0), (2, 128, 64, 16, 32, 1, 1, 0),
    (4, 64, 64, 4, 16, 1, 1, 0),
]


@pytest.mark.parametrize('n,ci,co,h,w,k,stride,pad', CONV_CASES)
def test_conv2d_backward_vs_torch_autograd(gf, n, ci, co, h, w, k, stride, pad):
    rs = np.random.RandomState(n + ci + co + h + k + stride)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    xd, wd, bd = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    with torch.enable_grad():
        yr = F.conv2d(xr, wr, br, stride=stride, padding=pad)
        gy = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
        yr.backward(gy.double())
        y = gf['gf'].conv2d(xd, wd, bd, stride=stride, padding=pad)
        y.backward(gy.to(DEV))
    assert rel_err(c(y), c(yr)) < 2e-5
    assert rel_err(c(xd.grad), c(xr.grad)) < 2e-5
    assert rel_err(c(wd.grad), c(wr.grad)) < 2e-5
    assert rel_err(c(bd.grad), c(br.grad)) < 2e-5
    # deterministic: the K slices are summed in a fixed order
    assert torch.equal(gf['kernels'].conv2d_wgrad(x.to(DEV), gy.to(DEV), k, k, stride, pad), wd.grad)
    # no_weight_gradients() (conv2d_gradfix.py:25-31)
    xd2, wd2 = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
    with torch.enable_grad(), gf['gf'].no_weight_gradients():
        gf['gf'].conv2d(xd2, wd2, None, stride=stride, padding=pad).backward(gy.to(DEV))
    assert wd2.grad is None and torch.equal(xd2.grad, xd.grad)


@pytest.mark.parametrize('n,ci,co,h,w,pad', [(2, 16, 8, 8, 8, 0), (1, 37, 70, 16, 20, 0), (2, 64, 32, 32, 32, 1), (1, 24, 24, 33, 65, 1)])
def test_conv_transpose2d_backward_vs_torch_autograd(gf, n, ci, co, h, w, pad):
    rs = np.random.RandomState(n + ci + co + h + pad)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((ci, co, 3, 3)) / np.sqrt(ci * 9)).astype(np.float32))
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    xd, wd = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
    with torch.enable_grad():
        yr = F.conv_transpose2d(xr, wr, stride=2, padding=pad)
        gy = torch.from_numpy(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
        yr.backward(gy.double())
        y = gf['gf'].conv_transpose2d(xd, wd, stride=2, padding=pad)
        y.backward(gy.to(DEV))
    assert rel_err(c(y), c(yr)) < 2e-5
    assert rel_err(c(xd.grad), c(xr.grad)) < 2e-5
    assert rel_err(c(wd.grad), c(wr.grad)) < 2e-5


def test_wgrad_rejects_bad_geometry(gf):
    from shgan_amd import _lib
    kk = gf['kernels']
    x, g = torch.zeros(1, 4, 8, 8, device=DEV), torch.zeros(1, 4, 8, 8, device=DEV)
    with pytest.raises(_lib.ShgError):
        kk.conv2d_wgrad(x, g, 3, 3, 1, 0)          # an 8x8 output needs padding 1
    with pytest.raises(_lib.ShgError):
        kk.conv2d_wgrad(x, g, 5, 5, 1, 2)
    with pytest.raises(_lib.ShgError):
        kk.conv2d_wgrad(x.cpu(), g, 3, 3, 1, 1)


def test_discriminator_loss_gradients_vs_reference_golden(gf):
    """One discriminator loss evaluation (stylegan_default_loss.py:96-117, Dmain) through the module tree on the HIP kernels --
    convolutions, FIR filters and bias + lrelu_agc forward AND backward in HIP, the two dense layers as library GEMMs -- against
    logits, loss and every parameter gradient produced by the reference's modules under torch autograd
    (tests/golden/discriminator_grads.npz, tools/gen_golden.py)."""
    from conftest import load_golden
    from shgan_amd.model_zoo import stylegan
    g = load_golden('discriminator_grads')
    D = stylegan.Discriminator(resolution=32, ic_n=4, ch_base=256, ch_max=16, use_fp16_before_res=None, mbstd_group_size=4, mbstd_c_n=1)
    D.load_state_dict({k[len('sd__'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}, strict=True)
    D = D.to(DEV).train().requires_grad_(True)
    fake = torch.from_numpy(g['fake']).to(DEV).requires_grad_(True)
    real = torch.from_numpy(g['real']).to(DEV)
    with torch.enable_grad():
        lf, lr = D(fake, None), D(real, None)
        loss = F.softplus(lf).mean() + F.softplus(-lr).mean()
        loss.backward()
    assert rel_err(c(lf), g['logits_fake']) < 1e-4 and rel_err(c(lr), g['logits_real']) < 1e-4
    assert abs(float(loss) - float(g['loss'])) < 1e-5 * abs(float(g['loss']))
    assert rel_err(c(fake.grad), g['grad__fake']) < 2e-4
    worst = 0.0
    for name, p in D.named_parameters():
        assert p.grad is not None, name
        e = rel_err(c(p.grad), g['grad__' + name])
        worst = max(worst, e)
        assert e < 2e-4, (name, e)
    print('worst parameter-gradient error', worst)
    # the inference path is untouched by the same modules when no gradient is requested
    with torch.no_grad():
        assert rel_err(c(D(fake.detach(), None)), g['logits_fake']) < 1e-4


def test_generator_parameter_gradients_vs_reference_golden(gf):
    """Training row N3: d/dtheta of sum(G(x, z) * r) / N for EVERY generator parameter and for z, through this package's module
    tree with gradients enabled (non-fused modulated convolutions on the HIP conv kernels forward AND backward, FIR up /
    down-sampling and bias + lrelu_agc with their HIP backward, SHU through library FFTs) against the reference's own modules
    under torch autograd (tests/golden/generator_grads.npz).  R = 256, reduced width, noise_mode='const', eval() (no dropout)."""
    from conftest import load_golden
    from shgan_amd import configs, eval_harness
    from oracle import shgan_oracle as orc
    g = load_golden('generator_grads')
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seed']), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim,
                             noise_strength=0.1, bias_std=0.1)
    G = configs.build_generator(res, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim)
    G.load_state_dict(sd, strict=True)
    G = G.to(DEV).eval().requires_grad_(True)
    real = torch.from_numpy(g['real_u8'].astype(np.float32)) / 127.5 - 1.0
    n = real.shape[0]
    mask = torch.from_numpy(np.unpackbits(g['mask_bits'])[: n * res * res].reshape(n, 1, res, res).astype(np.float32))
    x = eval_harness.assemble_input(real, mask).to(DEV)
    r = torch.from_numpy(np.random.RandomState(int(g['r_seed'])).standard_normal((n, 3, res, res)).astype(np.float32)).to(DEV)
    z = torch.from_numpy(g['z']).to(DEV).requires_grad_(True)
    with torch.enable_grad():
        img = G(x=x, z=z, c=torch.zeros(n, 0, device=DEV), noise_mode='const')
        loss = (img * r).sum() / n
        loss.backward()
    assert rel_err(c(img)[:, :, ::4, ::4], g['img_ds']) < 1e-3
    assert abs(float(loss) - float(g['loss'])) < 1e-3 * max(abs(float(g['loss'])), 1.0)
    # Gradients here are sums of ~1e5 signed terms: the golden is the reference's FLOAT32 run and carries that round-off itself (up to
    # 6e-3 on single weight tensors, 6e-4 on z).  tests/golden/generator_grads64.npz is the same functional evaluated by the reference in
    # float64 -- the yardstick: the HIP path must be as close to it as the reference's own float32 run is (x3; floor 2e-3 per tensor -- the level of the float32 weight-gradient sums over 1e5 pixels --, 1e-3 for z), not close to
    # one particular float32 rounding.  (With a library GEMM in the dense layers the HIP path happened to follow the golden's rounding
    # to 1.5e-3; on the dense kernels it sits 5e-3 from the golden and 1e-3 from float64 on the same tensors.)
    g64 = load_golden('generator_grads64')
    ez, ez_ref = rel_err(c(z.grad), g64['grad__z']), rel_err(g['grad__z'], g64['grad__z'])
    assert ez <= max(3 * ez_ref, 1e-3), (ez, ez_ref)
    errs, ref_errs = {}, {}
    for name, p in G.named_parameters():
        assert p.grad is not None, name
        gn = p.grad.reshape(-1)
        got = c(gn) if gn.numel() <= 4096 else np.concatenate([c(gn[:2048]), c(gn[-2048:])])
        ref, (rsum, rnorm) = g['grad__' + name], g['gsum__' + name]
        r64 = g64['grad__' + name]
        if float(np.abs(r64).max()) > 0:
            errs[name], ref_errs[name] = rel_err(got, r64), rel_err(ref, r64)
        if gn.numel() > 1:
            assert abs(float(gn.double().norm()) - rnorm) <= 2e-3 * max(rnorm, 1e-12) + 1e-12, name
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    med, med_ref = float(np.median(list(errs.values()))), float(np.median(list(ref_errs.values())))
    print(f'z.grad vs float64: HIP {ez:.2e}, reference float32 {ez_ref:.2e}; parameter gradients: median HIP {med:.2e} / reference {med_ref:.2e}; '
          'largest HIP (HIP, reference float32):', [(k, float('%.2e' % v), float('%.2e' % ref_errs[k])) for k, v in top])
    for name, e in errs.items():
        # the noise strengths are single numbers = a sum of ~1e5 signed products: cancellation costs them a digit more
        assert e <= max(3 * ref_errs[name], 2e-2 if name.endswith("noise_strength") else 2e-3), (name, e, ref_errs[name])
    assert med <= 2 * med_ref + 1e-6, (med, med_ref)


def test_discriminator_r1_regulariser_second_order_vs_reference_golden(gf):
    """Dreal + Dr1 of stylegan_default_loss.py:104-127: the R1 penalty |d logits / d image|^2 (first derivative with
    create_graph, weight gradients switched off as in the reference) differentiated again with respect to every parameter --
    convolution, FIR and activation backward passes are themselves differentiable HIP operators."""
    from conftest import load_golden
    from shgan_amd.model_zoo import stylegan
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    g = load_golden('discriminator_grads')
    D = stylegan.Discriminator(resolution=32, ic_n=4, ch_base=256, ch_max=16, use_fp16_before_res=None, mbstd_group_size=4, mbstd_c_n=1)
    D.load_state_dict({k[len('sd__'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}, strict=True)
    D = D.to(DEV).train().requires_grad_(True)
    with torch.enable_grad():
        real = torch.from_numpy(g['real']).to(DEV).requires_grad_(True)
        logits = D(real, None)
        with conv2d_gradfix.no_weight_gradients():
            (r1_grads,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[real], create_graph=True, only_inputs=True)
        r1 = r1_grads.square().sum([1, 2, 3])
        (logits * 0 + F.softplus(-logits) + r1.reshape(-1, 1) * (10.0 / 2)).mean().backward()
    assert rel_err(c(r1), g['r1_penalty']) < 1e-4
    errs = {name: rel_err(c(p.grad), g['r1grad__' + name]) for name, p in D.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print('largest R1 parameter-gradient errors:', [(k, float('%.2e' % v)) for k, v in top])
    assert max(errs.values()) < 5e-4, top


def test_generator_path_length_regulariser_second_order_vs_reference_golden(gf):
    """Gpl of stylegan_default_loss.py:76-91: |d (img . noise) / d ws| (create_graph) -> penalty -> gradients of every generator
    parameter that the penalty reaches, against the reference's autograd; a second-order path through the modulated
    convolutions (HIP forward / input-gradient / weight-gradient kernels), the FIR up-sampling and the activations."""
    from conftest import load_golden
    from shgan_amd import configs, eval_harness
    from oracle import shgan_oracle as orc
    g = load_golden('generator_grads')
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seed']), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim,
                             noise_strength=0.1, bias_std=0.1)
    G = configs.build_generator(res, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim)
    G.load_state_dict(sd, strict=True)
    G = G.to(DEV).eval().requires_grad_(True)
    real = torch.from_numpy(g['real_u8'].astype(np.float32)) / 127.5 - 1.0
    n = real.shape[0]
    mask = torch.from_numpy(np.unpackbits(g['mask_bits'])[: n * res * res].reshape(n, 1, res, res).astype(np.float32))
    x = eval_harness.assemble_input(real, mask).to(DEV)
    rs = np.random.RandomState(int(g['r_seed']))
    rs.standard_normal((n, 3, res, res))                                   # (the first draw was `r` of the first-order golden)
    pl_noise = torch.from_numpy(rs.standard_normal((n, 3, res, res)).astype(np.float32) / np.sqrt(res * res)).to(DEV)
    z = torch.from_numpy(g['z']).to(DEV)
    with torch.enable_grad():
        ws = G.mapping(z, torch.zeros(n, 0, device=DEV))
        xg, feats = G.encoder(x)
        img = G.synthesis(xg, feats, ws, noise_mode='const')
        (pl_grads,) = torch.autograd.grad(outputs=[(img * pl_noise).sum()], inputs=[ws], create_graph=True, only_inputs=True)
        pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
        (img[:, 0, 0, 0] * 0 + pl_lengths.square() * 2.0).mean().backward()
    # yardstick as in the first-order test: the reference's float64 evaluation (tests/golden/generator_grads64.npz); its own float32
    # run sits up to 8e-3 from it on single weight tensors of this second-order functional (median 1.2e-3)
    g64 = load_golden('generator_grads64')
    assert rel_err(c(pl_lengths), g64['pl_lengths']) < 1e-3 and rel_err(c(pl_lengths), g['pl_lengths']) < 1e-3
    errs, ref_errs, missing = {}, {}, []
    for name, p in G.named_parameters():
        key = 'plgrad__' + name
        if key not in g.files:
            continue
        if p.grad is None:
            missing.append(name)
            continue
        gn = p.grad.reshape(-1)
        got = c(gn) if gn.numel() <= 4096 else np.concatenate([c(gn[:2048]), c(gn[-2048:])])
        if float(np.abs(g64[key]).max()) > 0:
            errs[name], ref_errs[name] = rel_err(got, g64[key]), rel_err(g[key], g64[key])
    assert not missing, missing
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    med, med_ref = float(np.median(list(errs.values()))), float(np.median(list(ref_errs.values())))
    print(f'path-length parameter gradients vs float64: median HIP {med:.2e} / reference float32 {med_ref:.2e}; largest HIP (HIP, reference):',
          [(k, float('%.2e' % v), float('%.2e' % ref_errs[k])) for k, v in top])
    for name, e in errs.items():
        # (floor 1e-2 as for the same functional at full width, tests/test_gpu_config5.py: float32 sums of ~1e5 signed second-order terms)
        assert e <= max(3 * ref_errs[name], 5e-2 if name.endswith('noise_strength') else 1e-2), (name, e, ref_errs[name])
    assert med <= 2 * med_ref + 1e-6, (med, med_ref)


def test_stylegan2_loss_phases_match_hand_written_autograd(gf):
    """``losses.StyleGAN2Loss`` (the contract of stylegan_default_loss.py:16-128) on the plain StyleGAN2 generator / discriminator at
    reduced width: every phase accumulates exactly the gradients of the formula it stands for -- Gmain = softplus(-D(G(z))),
    Dmain = softplus(D(G(z))) + softplus(-D(x)), Dreg = r1_gamma/2 |dD/dx|^2, Greg = pl_weight (|J^T y| - pl_mean)^2 -- on the network
    being trained, scaled by ``gain``; the operators behind them are pinned on the reference in the tests above."""
    from shgan_amd import losses
    from shgan_amd.model_zoo import stylegan as sg
    torch.manual_seed(3)
    mp = sg.Mapping(z_dim=32, c_dim=0, w_dim=32, num_ws=8, num_layers=3, lr_multiplier=0.01, w_avg_beta=0.995)
    syn = sg.Synthesis(w_dim=32, resolution=32, rgb_n=3, ch_base=256, ch_max=16, use_fp16_after_res=32)
    G = sg.Generator(mp, syn).to(DEV).train()
    D = sg.Discriminator(resolution=32, ic_n=3, ch_base=256, ch_max=16, use_fp16_before_res=None, mbstd_group_size=4, mbstd_c_n=1).to(DEV).train()
    noise_fix = torch.randn(2, 3, 32, 32, device=DEV)
    L = losses.StyleGAN2Loss(DEV, G.mapping, G.synthesis, D, style_mixing_prob=0, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=2)
    L.randn_like = lambda t: noise_fix[:t.shape[0]]
    z, real, cnd = torch.randn(4, 32, device=DEV), torch.randn(4, 3, 32, 32, device=DEV), torch.zeros(4, 0, device=DEV)

    def grads(mod):
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in mod.named_parameters()}

    def zero():
        G.zero_grad(set_to_none=True); D.zero_grad(set_to_none=True)

    def synth(zz):                        # the generator as the loss calls it; the noise inputs are random per call -> use 'const'
        return G.synthesis(G.mapping(zz, cnd[:zz.shape[0]]), noise_mode='const')
    L.G_synthesis = lambda ws: G.synthesis(ws, noise_mode='const')

    # ---- Gmain
    zero(); G.requires_grad_(True); D.requires_grad_(False)
    L.accumulate_gradients('Gmain', real, cnd, z, cnd, sync=True, gain=2)
    got = grads(G)
    zero()
    with torch.enable_grad():
        (F.softplus(-D(synth(z), cnd)).mean() * 2).backward()
    want = grads(G)
    assert all(p.grad is None for p in D.parameters())
    for n in want:
        assert torch.allclose(got[n], want[n], rtol=1e-4, atol=1e-7), n
    # ---- Dboth = Dmain + Dreg in one pass over the real images
    zero(); G.requires_grad_(False); D.requires_grad_(True)
    L.accumulate_gradients('Dboth', real, cnd, z, cnd, sync=True, gain=1)
    got = grads(D)
    zero()
    with torch.enable_grad():
        with torch.no_grad():
            fake = synth(z)
        F.softplus(D(fake, cnd)).mean().backward()
        rt = real.detach().requires_grad_(True)
        lr = D(rt, cnd)
        (r1g,) = torch.autograd.grad([lr.sum()], [rt], create_graph=True)
        (F.softplus(-lr) + (r1g.square().sum([1, 2, 3]) * 5.0).reshape(-1, 1)).mean().backward()
    want = grads(D)
    for n in want:
        assert torch.allclose(got[n], want[n], rtol=2e-4, atol=1e-6), n
    assert all(p.grad is None for p in G.parameters())
    assert 'Loss/r1_penalty' in L.stats and L.stats['Loss/D/loss'].shape == (4, 1)
    # ---- Greg: path length on the shrunk batch, pl_mean updated with the EMA of the lengths
    zero(); G.requires_grad_(True); D.requires_grad_(False)
    assert float(L.pl_mean) == 0.0
    L.accumulate_gradients('Greg', real, cnd, z, cnd, sync=True, gain=4)
    got = grads(G)
    zero()
    with torch.enable_grad():
        ws = G.mapping(z[:2], cnd[:2])
        img = G.synthesis(ws, noise_mode='const')
        (plg,) = torch.autograd.grad([(img * (noise_fix / 32.0)).sum()], [ws], create_graph=True)
        pll = plg.square().sum(2).mean(1).sqrt()
        plm = torch.zeros([], device=DEV).lerp(pll.mean(), 0.01)
        ((img[:, 0, 0, 0] * 0 + (pll - plm).square() * 2).mean() * 4).backward()
    want = grads(G)
    assert abs(float(L.pl_mean) - float(plm)) <= 1e-5 * abs(float(plm))
    for n in want:
        if want[n] is None:
            assert got[n] is None or float(got[n].abs().max()) == 0.0, n
        else:
            assert torch.allclose(got[n], want[n], rtol=5e-4, atol=1e-6), n
    with pytest.raises(AssertionError):
        L.accumulate_gradients('Gall', real, cnd, z, cnd)
    with pytest.raises(NotImplementedError):
        losses.StyleGAN2Loss(DEV, G.mapping, G.synthesis, D, augment_pipe=object())


@pytest.mark.parametrize('ch,shared_noise,act', [(64, True, True), (37, False, True), (512, True, True), (24, True, False)])
def test_fused_modulation_tail_float32(gf, ch, shared_noise, act):
    """grad_ops.modconv_tail on float32 NCHW tensors: y = A(t*d + noise + bias) (the fused bias_act kernel) and its one-pass first-order
    backward (modtail_backward_f32_kernel: gt, d/dd, d/dnoise, d/dbias with in-kernel reductions) against torch autograd of the same
    expression; under create_graph the composed form must give the same first derivatives and be differentiable again."""
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    rs = np.random.RandomState(ch)
    n, h, w = 3, 12, 20
    t = torch.from_numpy(rs.standard_normal((n, ch, h, w)).astype(np.float32) * 40)
    d = torch.from_numpy((rs.rand(n, ch) + 0.5).astype(np.float32))
    nz = torch.from_numpy(rs.standard_normal((h, w) if shared_noise else (n, 1, h, w)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(ch).astype(np.float32))
    gy = torch.from_numpy(rs.standard_normal((n, ch, h, w)).astype(np.float32))
    with torch.enable_grad():
        tr, dr, nr, br = (v.clone().requires_grad_(True) for v in (t, d, nz, b))
        z = tr * dr.reshape(n, ch, 1, 1) + nr + br.reshape(1, ch, 1, 1)
        yr = (F.leaky_relu(z, 0.2) * np.sqrt(2)).clamp(-256, 256) if act else z
        yr.backward(gy)
    td, dd_, nd, bd = (v.to(DEV).requires_grad_(True) for v in (t, d, nz, b))
    kw = dict(act=act, gain=1.0, alpha=0.2, act_gain=float(np.sqrt(2)), clamp=256.0)
    with torch.enable_grad():
        y = grad_ops.modconv_tail(td, d=dd_, noise=nd, bias=bd, **kw)
        y.backward(gy.to(DEV))
    assert rel_err(c(y), yr.detach().numpy()) < 1e-6
    assert rel_err(c(td.grad), tr.grad.numpy()) < 1e-6
    assert rel_err(c(dd_.grad), dr.grad.numpy()) < 1e-5 and rel_err(c(nd.grad), nr.grad.numpy()) < 1e-5 and rel_err(c(bd.grad), br.grad.numpy()) < 1e-5
    td2, dd2 = t.to(DEV).requires_grad_(True), d.to(DEV).requires_grad_(True)
    with torch.enable_grad():
        y2 = grad_ops.modconv_tail(td2, d=dd2, noise=nd.detach(), bias=bd.detach(), **kw)
        g1, g2 = torch.autograd.grad([(y2 * gy.to(DEV)).sum()], [td2, dd2], create_graph=True)
        assert rel_err(c(g1), c(td.grad)) < 1e-6 and rel_err(c(g2), c(dd_.grad)) < 1e-5
        g1.square().sum().backward()
    assert dd2.grad is not None and torch.isfinite(dd2.grad).all() and float(dd2.grad.abs().max()) > 0
