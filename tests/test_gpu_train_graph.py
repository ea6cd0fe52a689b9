"""The training stage's HIP-graph form (train_stage.PhaseGraphs) against the eager loop (train_stage.run_phases), and the
composite that InpaintingLoss feeds the critic.  Reduced-width networks: the property tested is the stage, not the kernels."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def small_networks(seed):
    import shgan_amd  # noqa: F401
    from shgan_amd import configs
    from shgan_amd.model_zoo import stylegan
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=seed,
                             noise_strength=0.1, bias_std=0.1).to(DEV).train().requires_grad_(False)
    for m in G.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    torch.manual_seed(seed + 1)
    D = stylegan.Discriminator(resolution=256, ic_n=4, ch_base=2048, ch_max=32, mbstd_group_size=4, mbstd_c_n=1).to(DEV).train().requires_grad_(False)
    return G, D


def real_batch(n, seed):
    rs = np.random.RandomState(seed)
    real = torch.from_numpy(rs.uniform(-1, 1, size=(n, 3, 256, 256)).astype(np.float32))
    mask = torch.from_numpy((rs.uniform(size=(n, 1, 256, 256)) < 0.7).astype(np.float32))
    return torch.cat([mask - 0.5, real], dim=1).to(DEV)


def test_phase_graphs_follow_the_eager_loop():
    """Six iterations (Gmain + Dmain every time, Greg at 0 / 4, Dreg at 0) from the same state: eager ``run_phases`` vs ``PhaseGraphs``
    (two eager runs of a phase, then capture + replays).  Deterministic setting (noise_mode 'const', no style mixing, host latents from
    a seeded generator): the replays run the same kernels on the same data, so the parameters agree to round-off.
    A comparison of this repository with itself: the eager loop it follows is pinned on the reference by tests/test_gpu_config5.py (the four
    phases of the full-width step against the reference's autograd, float64 yardstick)."""
    from shgan_amd import losses, train_stage as ts
    G, D = small_networks(5)
    g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
    real4 = real_batch(4, 6)
    kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8, capturable=True)
    order = [0, 1, 2, 4, 5, 8, 9]           # Greg needs three visits (two eager, one captured) before its replay at 8 is a pure replay
    out = []
    for graphed in (False, True):
        G.load_state_dict(g0); D.load_state_dict(d0)
        torch.manual_seed(11)
        L = losses.InpaintingLoss(DEV, G, D, composite_fake=True, noise_mode='const', style_mixing_prob=0)
        phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
        pg = ts.PhaseGraphs(phases, L, 4, 64, tuple(real4.shape), DEV) if graphed else None
        ran = []
        for idx in order:
            ran.append(pg.run(real4, idx) if graphed else ts.run_phases(real4, 64, phases, batch_idx=idx, loss=L, batch_gpu=4, device=DEV))
        torch.cuda.synchronize()
        assert ran[0] == ['Gmain', 'Greg', 'Dmain', 'Dreg'] and ran[1] == ['Gmain', 'Dmain'] and ran[3] == ['Gmain', 'Greg', 'Dmain']
        if graphed:
            assert set(pg.graphs) == {'Gmain', 'Greg', 'Dmain'}        # Dreg ran once: still in its eager warm-up
        out.append({n: p.detach().clone() for n, p in list(G.named_parameters()) + [('D.' + n, p) for n, p in D.named_parameters()]})
        for ph in phases:
            if ph.sync is not None:
                ph.sync.remove()
    worst = 0.0
    for n in out[0]:
        a, b = out[0][n], out[1][n]
        assert torch.isfinite(b).all(), n
        worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-12)))
    moved = sum(int((out[1][n] - g0[n].to(DEV)).abs().max() > 0) for n in g0 if n in out[1])
    print(f'PhaseGraphs vs eager after {len(order)} iterations: worst relative parameter difference {worst:.2e}; {moved} G parameters moved')
    assert worst < 1e-4 and moved > 50


def test_inpainting_loss_composites_the_fake_with_the_known_pixels():
    """What the critic sees for a generated image: cat([mask - 0.5, G(x) * (1 - mask) + real * mask]) (CoModGAN; the reference's
    evaluation composite shgan_default.py:259); ``composite_fake=False`` hands it the raw output."""
    from shgan_amd import losses
    G, D = small_networks(7)
    real4 = real_batch(2, 8)
    seen = []

    class Spy(torch.nn.Module):
        def forward(self, img, c):
            seen.append(img.detach().clone())
            return img.mean(dim=(1, 2, 3)).reshape(-1, 1)
    z = torch.randn(2, 64, device=DEV)
    for comp in (True, False):
        L = losses.InpaintingLoss(DEV, G, Spy(), noise_mode='const', style_mixing_prob=0, composite_fake=comp)
        G.requires_grad_(True)
        L.accumulate_gradients('Gmain', real4, torch.zeros(2, 0, device=DEV), z, torch.zeros(2, 0, device=DEV))
        G.requires_grad_(False)
    comp_in, raw_in = seen
    m = real4[:, 0:1] + 0.5
    assert torch.equal(comp_in[:, 0:1], real4[:, 0:1]) and torch.equal(raw_in[:, 0:1], real4[:, 0:1])
    assert torch.allclose(comp_in[:, 1:4], raw_in[:, 1:4] * (1 - m) + real4[:, 1:4] * m, atol=1e-6)
    known = m.expand(-1, 3, -1, -1) > 0
    assert torch.equal(comp_in[:, 1:4][known], real4[:, 1:4][known])              # the known region is exactly the real image
    assert not torch.equal(raw_in[:, 1:4][known], real4[:, 1:4][known])
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in G.synthesis.parameters())


@pytest.mark.parametrize('n,k,m', [(8, 512, 512), (3, 1536, 70), (17, 33, 129), (1, 64, 5)])
def test_dense_ops_first_and_second_order_vs_torch_float64(n, k, m):
    """dense_ops.linear (x W^T + b with both gains folded in) and its closed family of gradient kernels against torch's float64 autograd:
    output, first-order gradients of (x, W, b), and the gradient of a function of the first-order input gradient (what the path-length
    regulariser does to the style affines)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import dense_ops
    rs = np.random.RandomState(n * 1000 + k + m)
    x0, w0, b0 = rs.standard_normal((n, k)), rs.standard_normal((m, k)), rs.standard_normal(m)
    gy0, u0 = rs.standard_normal((n, m)), rs.standard_normal((n, k))
    wg, bg = 0.37, 1.9

    def run(dt, dev, fn):
        x, w, b = (torch.tensor(v, dtype=dt, device=dev, requires_grad=True) for v in (x0, w0, b0))
        gy, u = torch.tensor(gy0, dtype=dt, device=dev), torch.tensor(u0, dtype=dt, device=dev)
        y = fn(x, w, b)
        gx, gw, gb = torch.autograd.grad((y * gy).sum(), [x, w, b], create_graph=True)
        # second order: a scalar of the input gradient (depends on W only through gx = wg * gy @ W), differentiated w.r.t. W; plus the
        # mixed term through y^2
        pen = (gx * u).square().sum() + (y.square() * gy).sum()
        hw, hx = torch.autograd.grad(pen, [w, x])
        return [t.detach().cpu().double().numpy() for t in (y, gx, gw, gb, hw, hx)]
    with torch.enable_grad():
        ref = run(torch.float64, 'cpu', lambda x, w, b: x @ (w * wg).t() + b * bg)
        got = run(torch.float32, DEV, lambda x, w, b: dense_ops.linear(x, w, b, wg, bg))
    for name, r, g in zip(('y', 'gx', 'gw', 'gb', 'hw', 'hx'), ref, got):
        err = np.abs(r - g).max() / (np.abs(r).max() + 1e-30)
        assert err < 2e-5, (name, err)


def test_dense_layer_training_route_uses_no_library_gemm():
    """``dense.forward`` under autograd == its inference kernel, and ``torch.profiler`` sees no rocBLAS kernel in forward + backward."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan
    from torch.profiler import profile, ProfilerActivity
    torch.manual_seed(3)
    layer = stylegan.dense(512, 384, activation='lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)', lr_multi=0.01).to(DEV)
    x = torch.randn(8, 512, device=DEV)
    with torch.no_grad():
        y_inf = layer(x)
    layer.requires_grad_(True)
    for attempt in range(3):                     # (the tracer occasionally drops the events of a short region)
        with torch.enable_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
            y = layer(x.clone().requires_grad_(True))
            y.square().sum().backward()
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
        if any('dense_kernel' in k for k in names):
            break
    assert torch.allclose(y.detach(), y_inf, rtol=1e-5, atol=1e-6)
    assert any('dense_kernel' in k for k in names) and any('matmul_tn_kernel' in k for k in names), names
    assert not any(k.startswith('Cijk_') or 'rocblas' in k.lower() for k in names), names


def test_shu_training_route_matches_the_fft_formulation():
    """SHU.forward under autograd (transform stages = the inference kernels + their transposes, csrc/shu.hip) against the reference's
    formulation with torch.fft (shgan.py:312-336) evaluated in float64 on the CPU: the five hints, and the gradients of a random
    functional w.r.t. the input and the SHU's parameters.  No rocFFT / rocBLAS kernel may run in the HIP version."""
    import shgan_amd  # noqa: F401
    from shgan_amd import configs
    from torch.profiler import profile, ProfilerActivity
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=9, bias_std=0.1)
    shu = G.encoder.shu.to(DEV)
    rs = np.random.RandomState(10)
    x0 = rs.standard_normal((3, shu.conv0.weight.shape[1] // 2, 64, 64))
    ws = {r: rs.standard_normal((3, shu.out_channels, r, r)) for r in shu.reslist}

    def fft_form(x, params, gauss, cw, dt):
        w0, b0, w1 = params
        sp = torch.fft.rfftn(x, dim=(2, 3), norm='forward')
        sp = torch.cat([sp[:, :, 33:], sp[:, :, :33]], dim=2)
        t = torch.cat([sp.real, sp.imag], dim=1)
        t = torch.relu(torch.nn.functional.conv2d(t, w0 * shu.conv0.weight_gain, b0))
        y = torch.nn.functional.conv2d(t, w1.t()[:, :, None, None])             # df1.weight is [in, out * bands]; flat output channel = o * bands + k
        nb = cw.shape[0]
        y = (y.reshape(y.shape[0], -1, nb, 64, 33) * cw[None, None]).sum(2)
        c = y.shape[1] // 2
        sp = torch.complex(y[:, :c], y[:, c:])
        out = {}
        for r in shu.reslist:
            s_ = sp[:, :, 32 - r // 2: 32 + r // 2, 0: r // 2 + 1] * gauss[r][None, None]
            s_ = torch.cat([s_[:, :, r - r // 2 - 1:], s_[:, :, :r - r // 2 - 1]], dim=2)
            out[r] = torch.fft.irfftn(s_, dim=(2, 3), norm='forward')
        return out

    with torch.enable_grad():
        # float64 reference on the CPU
        xr = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
        pr = [p.detach().cpu().double().requires_grad_(True) for p in (shu.conv0.weight, shu.conv0.bias, shu.df1.weight)]
        gauss = {r: getattr(shu, f'_gauss{r}').cpu().double() for r in shu.reslist}
        ref = fft_form(xr, pr, gauss, shu._cw.cpu().double(), torch.float64)
        loss_r = sum((ref[r] * torch.tensor(ws[r])).sum() for r in shu.reslist)
        gref = torch.autograd.grad(loss_r, [xr] + pr)
        # HIP
        shu.requires_grad_(True)
        xg = torch.tensor(x0, dtype=torch.float32, device=DEV, requires_grad=True)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            got = shu(xg)
            loss_g = sum((got[r] * torch.tensor(ws[r], dtype=torch.float32, device=DEV)).sum() for r in shu.reslist)
            ggot = torch.autograd.grad(loss_g, [xg, shu.conv0.weight, shu.conv0.bias, shu.df1.weight])
            torch.cuda.synchronize()
        shu.requires_grad_(False)
    for r in shu.reslist:
        e = float((got[r].detach().cpu().double() - ref[r].detach()).abs().max() / ref[r].detach().abs().max())
        assert e < 2e-5, (r, e)
    for name, a, b in zip(('x', 'conv0.weight', 'conv0.bias', 'df1.weight'), ggot, gref):
        e = float((a.detach().cpu().double() - b).abs().max() / b.abs().max())
        assert e < 5e-5, (name, e)
    names = [e.key for e in prof.key_averages()]
    assert any('shu_split_adjoint_kernel' in k for k in names), names
    assert not any(k.startswith('Cijk_') or ('fft' in k.lower() and not k.startswith('shu_')) for k in names), names       # (rocBLAS / rocFFT kernel names)


def test_layer_routes_are_decided_in_one_place_and_traceable():
    """``stylegan.layer_route`` is the single dispatcher of the layer classes; ``ROUTE_TRACE`` records what every call took: float32 under
    no_grad -> the fused inference kernels, gradients requested -> the differentiable operators, fp16 blocks under no_grad -> the fused
    half kernels."""
    import shgan_amd  # noqa: F401
    from shgan_amd import configs
    from shgan_amd.model_zoo import stylegan
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=1).to(DEV).eval()
    G16 = configs.seeded_init_(configs.build_generator(256, ch_base=4096, ch_max=64, w_dim=64, z_dim=64, w0_dim=128, use_fp16_before_res=64,
                                                       use_fp16_after_res=32), seed=1).to(DEV).eval().requires_grad_(False)
    x = real_batch(2, 3)
    z = torch.randn(2, 64, device=DEV)
    cnd = torch.zeros(2, 0, device=DEV)
    try:
        stylegan.ROUTE_TRACE = []
        G.requires_grad_(False)
        G(x=x, z=z, c=cnd, noise_mode='const')
        infer = stylegan.ROUTE_TRACE
        stylegan.ROUTE_TRACE = []
        G.requires_grad_(True)
        with torch.enable_grad():
            G(x=x, z=z, c=cnd, noise_mode='const')
        train = stylegan.ROUTE_TRACE
        stylegan.ROUTE_TRACE = []
        G16(x=x, z=z, c=cnd, noise_mode='const')
        half = stylegan.ROUTE_TRACE
    finally:
        stylegan.ROUTE_TRACE = None
        G.requires_grad_(False)
    assert len(infer) > 30 and {r for _, r, _, _ in infer} == {'f32_fused'}
    assert len(train) == len(infer) and {r for _, r, _, _ in train} == {'generic'}
    routes16 = {(r, d) for _, r, d, _ in half}
    assert ('f16_fused', 'float16') in routes16 and ('f32_fused', 'float32') in routes16 and not any(r == 'generic' for r, _ in routes16)


def test_graph_pipeline_replays_the_evaluation_step_bit_exactly():
    """``eval_harness.GraphPipeline`` (every slot's forward captured once as a HIP graph) against the eager ``run_generator`` on a sequence
    of different batches: identical uint8 results (noise_mode 'const': no device RNG in the step), also after the parameters changed
    and the slots were re-captured."""
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, eval_harness as hz
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=2,
                             noise_strength=0.1).to(DEV).eval().requires_grad_(False)
    batches = [hz.synthetic_batch(2, 256, 64, seed=s, device=DEV, masks='bernoulli')[:2] for s in (1, 2, 3, 4, 5)]

    def step(x, z):
        return hz.run_generator(G, x, z, noise_mode='const')
    eager = [step(x, z).clone() for x, z in batches]
    pipe = hz.GraphPipeline(DEV, step, batches[0], depth=2)
    got = []
    for x, z in batches:
        out = pipe.run(x, z)
        pipe.join()
        got.append(out.clone())
    for a, b in zip(eager, got):
        assert torch.equal(a, b)
    assert not torch.equal(got[0], got[1])


def test_graph_pipeline_recaptures_when_the_parameters_move():
    """A captured slot holds the ADDRESSES of the per-parameter caches (prepared weight layouts, the host-read noise strength).  With
    ``watch`` the pipeline notices an in-place parameter update (version counters: optimiser step, EMA, ``load_state_dict``) and a
    replayed training graph (``_ParamCache.invalidate_all()``) and re-captures; the replay then equals the eager step on the NEW weights.
    Carried by ``test_graph_pipeline_replays_the_evaluation_step_bit_exactly`` (eager == replay) and, through ``run_generator``, by the
    reference-pinned generator goldens of tests/test_gpu_generator.py."""
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, eval_harness as hz
    from shgan_amd.model_zoo.stylegan import _ParamCache
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=2,
                             noise_strength=0.1).to(DEV).eval().requires_grad_(False)
    x, z = hz.synthetic_batch(2, 256, 64, seed=7, device=DEV, masks='bernoulli')[:2]

    def step(x_, z_):
        return hz.run_generator(G, x_, z_, noise_mode='const')
    pipe = hz.GraphPipeline(DEV, step, (x, z), depth=2, watch=list(G.parameters()) + list(G.buffers()))
    before = pipe.run(x, z)
    pipe.join()                                    # (the replay runs on the slot's stream: join before the current stream reads the output)
    before = before.clone()
    assert pipe.captures == 1 and torch.equal(before, step(x, z))
    for p in G.parameters():                       # what an EMA update / optimiser step does: in place, new version counters
        p.mul_(1.25)
    for _ in range(3):
        after = pipe.run(x, z)
        pipe.join()
        after = after.clone()
    assert pipe.captures == 2                      # one re-capture, then plain replays
    assert torch.equal(after, step(x, z)) and not torch.equal(after, before)
    _ParamCache.invalidate_all()                   # what train_stage.PhaseGraphs does around its replays
    again = pipe.run(x, z)
    pipe.join()
    again = again.clone()
    assert pipe.captures == 3 and torch.equal(again, after)


@pytest.mark.parametrize('o,i,k,half', [(64, 32, 3, False), (512, 512, 3, True), (70, 13, 3, True), (128, 64, 1, False)])
def test_fused_demodulation_weight_kernel_vs_tensor_ops(o, i, k, half):
    """``_weight_factors`` under autograd: the one-kernel form (csrc/dense.hip demod_weight / demod_weight_backward) against the tensor-op
    composition of stylegan.py:136-138,146,150-155 in float64 -- outputs (wn, wsq) and the gradient of a random functional of both."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan
    rs = np.random.RandomState(o + i + k)
    w0 = rs.standard_normal((o, i, k, k))
    a0, b0 = rs.standard_normal((o, i, k, k)), rs.standard_normal((o, i))
    with torch.enable_grad():
        w = torch.tensor(w0, dtype=torch.float32, device=DEV, requires_grad=True)
        keep = stylegan.FUSED_DEMOD_WEIGHT
        try:
            stylegan.FUSED_DEMOD_WEIGHT = True
            wn, wsq = stylegan._weight_factors(half, w, True)
        finally:
            stylegan.FUSED_DEMOD_WEIGHT = keep
        assert type(wn.grad_fn).__name__ == '_DemodWeightFnBackward'
        (gw,) = torch.autograd.grad((wn * torch.tensor(a0, dtype=torch.float32, device=DEV)).sum()
                                    + (wsq * torch.tensor(b0, dtype=torch.float32, device=DEV)).sum(), [w])
        w64 = torch.tensor(w0, dtype=torch.float64, requires_grad=True)
        old = stylegan.FUSED_DEMOD_WEIGHT
        try:
            stylegan.FUSED_DEMOD_WEIGHT = False
            wn64, wsq64 = stylegan._weight_factors(half, w64, True)
        finally:
            stylegan.FUSED_DEMOD_WEIGHT = old
        (gw64,) = torch.autograd.grad((wn64 * torch.tensor(a0)).sum() + (wsq64 * torch.tensor(b0)).sum(), [w64])
    for name, got, ref in (('wn', wn, wn64), ('wsq', wsq, wsq64), ('gw', gw, gw64)):
        e = float((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        assert e < 1e-5, (name, e)


@pytest.mark.parametrize('fp16', [False, True])
def test_dmain_one_critic_pass_over_the_stacked_batch_equals_the_two_passes_of_the_reference(fp16):
    """stylegan_default_loss.py:84-106 judges the generated and the real batch in two critic passes with a backward each.  The product
    stacks them (``Discriminator.forward(segments=2)``: minibatch statistic per half) and runs one backward of the summed losses.  No
    other layer mixes samples: the logits must be IDENTICAL, the parameter gradients equal up to the order of the weight-gradient sums.
    Self-comparison (stacked pass vs the two passes of THIS repository); the two-pass form is what tests/test_gpu_config5.py holds against
    the reference's autograd (Dmain gradients of the full-width critic), and that test runs with the stacked pass switched on."""
    import shgan_amd  # noqa: F401
    from shgan_amd import losses
    from shgan_amd.model_zoo import stylegan
    G, _ = small_networks(21)
    torch.manual_seed(22)
    D = stylegan.Discriminator(resolution=256, ic_n=4, ch_base=2048, ch_max=32, mbstd_group_size=4, mbstd_c_n=1,
                               use_fp16_before_res=(32 if fp16 else None)).to(DEV).train().requires_grad_(False)
    real4 = real_batch(8, 23)
    z, c = torch.randn(8, 64, device=DEV), torch.zeros(8, 0, device=DEV)
    res = []
    for batched in (False, True):
        L = losses.InpaintingLoss(DEV, G, D, composite_fake=True, noise_mode='const', style_mixing_prob=0, batch_critic=batched)
        D.requires_grad_(True)
        for p in D.parameters():
            p.grad = None
        L.accumulate_gradients('Dmain', real4, c, z, c)
        D.requires_grad_(False)
        res.append(({k: v.clone() for k, v in L.stats.items()}, {n: p.grad.clone() for n, p in D.named_parameters()}))
    (s0, g0), (s1, g1) = res
    for k in ('Loss/scores/fake', 'Loss/scores/real', 'Loss/D/loss'):
        assert torch.equal(s0[k], s1[k]), k
    # the stacked statistic is not the statistic of the stacked batch: a plain 16-sample pass gives other logits
    with torch.no_grad():
        plain = D(torch.cat([real4, real4.flip(0)]), None)
        halves = D(torch.cat([real4, real4.flip(0)]), None, segments=2)
        assert torch.equal(halves[:8], D(real4, None)) and not torch.equal(plain[:8], halves[:8])
    worst = max(float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-20)) for n in g0)
    print(f'Dmain, one stacked critic pass vs two passes ({"fp16 blocks" if fp16 else "float32"}): worst relative gradient difference {worst:.2e}')
    assert worst < (2e-2 if fp16 else 1e-5)


@pytest.mark.parametrize('half', [False, True])
@pytest.mark.parametrize('n,i,o', [(8, 512, 512), (4, 64, 128), (16, 512, 64), (3, 100, 70)])
def test_fused_style_factors_vs_float64_autograd(half, n, i, o):
    """``_StyleFactorsFn`` (csrc/dense.hip style_factors kernels: normalised styles + demodulation coefficients, stylegan.py:138,147,155)
    against the same formulas evaluated by torch autograd in float64: both outputs, the first-order gradients with respect to the
    styles and to wsq, and a second-order quantity (the gradient of a function of the styles' gradient -- what the path-length
    regulariser needs; that pass runs the composed form under create_graph)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan as sg
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + i + o)
    s64 = (torch.randn(n, i, generator=g, dtype=torch.float64) + 1.0).to(DEV)
    w64 = torch.rand(o, i, generator=g, dtype=torch.float64).to(DEV) * 0.01
    a64, b64 = torch.randn(n, i, generator=g, dtype=torch.float64).to(DEV), torch.randn(n, o, generator=g, dtype=torch.float64).to(DEV)

    def ref(s, w):
        if half:
            s = s / s.norm(float('inf'), dim=1, keepdim=True)
        s = s * s.square().mean().rsqrt()
        return s, (s.square().matmul(w.t()) + 1e-8).rsqrt()

    def run(fn, s, w, a, b):
        s, w = s.clone().requires_grad_(True), w.clone().requires_grad_(True)
        with torch.enable_grad():
            sn, d = fn(s, w)
            loss = (sn * a).sum() + (d * b).sum()
            gs, gw = torch.autograd.grad(loss, [s, w], create_graph=False)
            sn2, d2 = fn(s, w)
            (g1,) = torch.autograd.grad((sn2 * a).sum() + (d2 * b).sum(), [s], create_graph=True)
            (gg,) = torch.autograd.grad(g1.square().sum(), [s])
        return sn.detach(), d.detach(), gs, gw, gg
    want = run(ref, s64, w64, a64, b64)
    got = run(lambda s, w: sg._StyleFactorsFn.apply(s, w, half), s64.float(), w64.float(), a64.float(), b64.float())
    for name, x, y in zip(('sn', 'dcoefs', 'g_styles', 'g_wsq', 'second-order g_styles'), got, want):
        err = float((x.double() - y).abs().max() / (y.abs().max() + 1e-30))
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize('half,n,i,o', [(False, 4, 512, 512), (True, 4, 128, 64), (False, 8, 64, 128), (True, 3, 40, 24), (False, 1, 32, 3)])
def test_closed_double_backward_of_the_style_factors_vs_float64_autograd(half, n, i, o):
    """``_StyleFactorsBwdFn`` (the first-order backward of the style factors as one node whose own backward is the closed form of the
    second derivative): gradients of <A, g_styles> + <B, g_wsq> with respect to the styles, the weight squares AND the incoming
    gradients, against torch's float64 double backward of the tensor-op formulation (stylegan.py:138,147,155) -- and against the composed
    route of this package (switch off).  Reference pin of the users: the path-length phase of tests/test_gpu_config5.py."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan as sg
    g = torch.Generator(device='cpu').manual_seed(n * 977 + i + 3 * o)
    mk = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64).to(DEV)           # noqa: E731
    s64, w64 = mk(n, i) + 1.0, torch.rand(o, i, generator=g, dtype=torch.float64).to(DEV) * 0.01
    a64, b64, A64, B64 = mk(n, i), mk(n, o), mk(n, i), mk(o, i)

    def ref(s, w):
        if half:
            s = s / s.norm(float('inf'), dim=1, keepdim=True)
        s = s * s.square().mean().rsqrt()
        return s, (s.square().matmul(w.t()) + 1e-8).rsqrt()

    def run(fn, dt):
        s, w, a, b = (t.to(dt).clone().requires_grad_(True) for t in (s64, w64, a64, b64))
        with torch.enable_grad():
            sn, d = fn(s, w)
            gs, gw = torch.autograd.grad([sn, d], [s, w], [a, b], create_graph=True)
            phi = (gs * A64.to(dt)).sum() + (gw * B64.to(dt)).sum()
            return [gs.detach(), gw.detach()] + list(torch.autograd.grad(phi, [s, w, a, b]))
    want = run(ref, torch.float64)
    fused = lambda s, w: sg._StyleFactorsFn.apply(s, w, half)                               # noqa: E731
    assert sg.CLOSED_STYLE_FACTORS_BACKWARD
    got = run(fused, torch.float32)
    sg.CLOSED_STYLE_FACTORS_BACKWARD = False
    try:
        composed = run(fused, torch.float32)
    finally:
        sg.CLOSED_STYLE_FACTORS_BACKWARD = True
    names = ('g_styles', 'g_wsq', 'second-order: styles', 'wsq', 'incoming g_sn', 'incoming g_d')
    for name, x, y, z in zip(names, got, want, composed):
        err, err_c = rel_err(x.double().cpu(), y.cpu()), rel_err(z.double().cpu(), y.cpu())
        assert err < 5e-5, (name, err, err_c)
        assert err < 4 * err_c + 1e-5, (name, err, err_c)          # (no worse than the composed float32 route it replaces)


@pytest.mark.parametrize('fp16', [False, True])
def test_residual_block_input_gradients_joined_in_the_convolution_kernel(fp16):
    """``grad_ops.InputGradJoin``: in the critic's residual blocks the skip branch's input gradient is added by conv0's input-gradient
    kernel (its ``residual`` operand) instead of by autograd's accumulation pass.  Same gradients as the ordinary path (switch off) for
    the parameters and for the image; the kernel path is really taken (every block joins once); the R1 pass (create_graph) is unchanged.
    Self-comparison (switch on vs off); the reference pin is tests/test_gpu_config5.py (critic gradients, first and second order, against
    the reference's autograd at full width -- run with the join on) and the discriminator goldens of tests/test_gpu_backward.py."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    torch.manual_seed(31)
    D = stylegan.Discriminator(resolution=256, ic_n=4, ch_base=2048, ch_max=32, mbstd_group_size=4, mbstd_c_n=1,
                               use_fp16_before_res=(32 if fp16 else None)).to(DEV).train()
    img = real_batch(4, 32)
    calls = []
    orig = conv2d_gradfix._conv_input_grad

    def spy(g, weight, x_shape, stride, padding, residual=None):
        calls.append(residual is not None)
        return orig(g, weight, x_shape, stride, padding, residual=residual)
    res = {}
    for on in (True, False):
        stylegan.JOIN_INPUT_GRADS = on
        calls.clear()
        conv2d_gradfix._conv_input_grad = spy
        try:
            with torch.enable_grad():
                x = img.clone().requires_grad_(True)
                for p in D.parameters():
                    p.grad = None
                torch.nn.functional.softplus(D(x, None)).mean().backward()
                first = ([p.grad.clone() for p in D.parameters()], x.grad.clone(), sum(calls))
                x2 = img.clone().requires_grad_(True)
                with conv2d_gradfix.no_weight_gradients():
                    (r1,) = torch.autograd.grad(D(x2, None).sum(), [x2], create_graph=True)
                for p in D.parameters():
                    p.grad = None
                r1.square().sum().backward()
                second = [None if p.grad is None else p.grad.clone() for p in D.parameters()]
        finally:
            conv2d_gradfix._conv_input_grad = orig
            stylegan.JOIN_INPUT_GRADS = True
        res[on] = (first, second)
    (ga, xa, na), sa = res[True]
    (gb, xb, nb), sb = res[False]
    assert na == 6 and nb == 0, (na, nb)                   # six residual blocks (256 .. 8), each joined once
    tol = 2e-2 if fp16 else 1e-5
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-20)) for a, b in zip(ga, gb))
    assert worst < tol and float((xa - xb).abs().max() / xb.abs().max()) < tol, worst
    for a, b in zip(sa, sb):
        assert (a is None) == (b is None) and (a is None or float((a - b).abs().max() / (b.abs().max() + 1e-20)) < tol)


@pytest.mark.parametrize('fp16', [False, True])
def test_conv_bias_act_as_one_training_node_equals_the_two_nodes(fp16):
    """3x3 stride-1 layers under autograd: ``_ConvBiasActFn`` (bias + lrelu_agc in the convolution's store pass, one node) against
    convolution node + bias/activation node -- critic logits, parameter gradients, image gradient, and the R1 second-order gradients.
    Self-comparison (one node vs two nodes of this repository); the two-node form is pinned on the reference's discriminator golden
    (tests/test_gpu_backward.py) and both run under tests/test_gpu_config5.py (Dmain / Dreg vs the reference's autograd, fused switch on)."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix
    torch.manual_seed(41)
    D = stylegan.Discriminator(resolution=256, ic_n=4, ch_base=2048, ch_max=32, mbstd_group_size=4, mbstd_c_n=1,
                               use_fp16_before_res=(32 if fp16 else None)).to(DEV).train()
    with torch.no_grad():
        for n_, p_ in D.named_parameters():
            if n_.endswith('bias'):
                p_.normal_(0, 0.2)
    img = real_batch(4, 42)
    res, taken = {}, {}
    orig = conv2d_gradfix.conv2d_bias_act

    def counted(*a, **k):
        taken[True] = taken.get(True, 0) + 1
        return orig(*a, **k)
    for on in (True, False):
        conv2d_gradfix.FUSED_CONV_ACT = on
        conv2d_gradfix.conv2d_bias_act = counted if on else orig
        try:
            with torch.enable_grad():
                x = img.clone().requires_grad_(True)
                for p in D.parameters():
                    p.grad = None
                logits = D(x, None)
                torch.nn.functional.softplus(logits).mean().backward()
                first = [logits.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in D.parameters()]
                x2 = img.clone().requires_grad_(True)
                with conv2d_gradfix.no_weight_gradients():
                    (r1,) = torch.autograd.grad(D(x2, None).sum(), [x2], create_graph=True)
                for p in D.parameters():
                    p.grad = None
                r1.square().sum().backward()
                second = [r1.detach().clone()] + [p.grad.clone() for p in D.parameters() if p.grad is not None]
        finally:
            conv2d_gradfix.FUSED_CONV_ACT = True
            conv2d_gradfix.conv2d_bias_act = orig
        res[on] = first + second
    assert taken.get(True, 0) == 2 * 14                    # fromrgb, conv0 and skip of the six residual blocks, the 4x4 tail's convolution; two critic passes
    tol = 3e-2 if fp16 else 2e-5
    assert len(res[True]) == len(res[False])
    worst = max(float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-20)) for a, b in zip(res[True], res[False]))
    print(f'fused conv+bias+act node vs two nodes ({"fp16 blocks" if fp16 else "float32"}): worst relative difference {worst:.2e}')
    assert worst < tol


def test_encoder_feature_gradients_joined_in_the_down_layer():
    """The co-modulation encoder's feature maps feed the block's stride-2 layer and the synthesis network: the gradient arriving from the
    synthesis side is added by the down layer's input-gradient kernel (``upfir_planar`` residual operand) -- same generator gradients as
    with the switch off, and the kernel path is taken once per float32 encoder block.
    Self-comparison (switch on vs off); the generator gradients themselves are pinned on the reference by tests/test_gpu_config5.py
    (Gmain / Greg against the reference's autograd, run with the join on)."""
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels
    from shgan_amd.model_zoo import stylegan
    G, _ = small_networks(51)
    real4 = real_batch(2, 52)
    x = torch.cat([real4[:, 0:1], real4[:, 1:4] * (real4[:, 0:1] + 0.5)], dim=1)
    z, c = torch.randn(2, 64, device=DEV), torch.zeros(2, 0, device=DEV)
    w = torch.randn(2, 3, 256, 256, device=DEV)
    used = []
    orig = kernels.upfir_planar

    def spy(*a, **k):
        used.append(k.get('residual') is not None)
        return orig(*a, **k)
    res = {}
    for on in (True, False):
        stylegan.JOIN_INPUT_GRADS = on
        used.clear()
        kernels.upfir_planar = spy
        try:
            G.requires_grad_(True)
            for p in G.parameters():
                p.grad = None
            with torch.enable_grad():
                img = G(x=x, z=z, c=c, noise_mode='const')
                (img * w).sum().backward()
            res[on] = ([None if p.grad is None else p.grad.clone() for p in G.parameters()], sum(used))
        finally:
            kernels.upfir_planar = orig
            stylegan.JOIN_INPUT_GRADS = True
            G.requires_grad_(False)
    (ga, na), (gb, nb) = res[True], res[False]
    assert na == 6 and nb == 0, (na, nb)                 # encoder blocks 256 .. 8
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12


@pytest.mark.parametrize('half', [False, True])
@pytest.mark.parametrize('with_d', [True, False])
def test_closed_tail_backward_node_vs_tensor_operators(half, with_d):
    """``grad_ops._ModTailBwdFn`` (the modulation tail's first-order backward as one differentiable node on the fused kernel; its own
    backward = the same kernel with the optional second product) against the tensor-operator composition it replaces under
    ``create_graph``: a path-length-shaped double backward of ``modconv_tail`` -- first derivative with respect to (t, d), then the gradient
    of a random functional of that derivative with respect to t, d, bias and the noise.  A self-comparison of two routes of this repo; the
    reference-pinned carriers are tests/test_gpu_config5.py (Greg / Dreg of the full-width step against the reference's autograd) and
    tests/test_gpu_backward.py::test_stylegan2_loss_phases_match_hand_written_autograd, both of which run with the closed node on."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    rs = np.random.RandomState(5 + half + 2 * with_d)
    n, c, h, w = 3, 32, 16, 24
    dt = torch.float16 if half else torch.float32

    def leaf(a, dtype=torch.float32):
        t_ = torch.from_numpy(a.astype(np.float32)).to(DEV).to(dtype)
        if dtype == torch.float16:
            t_ = t_.contiguous(memory_format=torch.channels_last)
        return t_.requires_grad_(True)
    t0, d0 = rs.standard_normal((n, c, h, w)), rs.uniform(0.5, 1.5, (n, c))
    b0, z0 = rs.standard_normal(c) * 0.1, rs.standard_normal((h, w)) * 0.1
    r1, r2 = rs.standard_normal((n, c, h, w)), rs.standard_normal((n, c))
    gy0 = rs.standard_normal((n, c, h, w))
    res = {}
    for closed in (True, False):
        grad_ops.CLOSED_TAIL_BACKWARD = closed
        try:
            t, d, b, z = leaf(t0, dt), (leaf(d0) if with_d else None), leaf(b0), leaf(z0)
            with torch.enable_grad():
                y = grad_ops.modconv_tail(t, d=d, noise=z, bias=b, act=True, gain=1.0)
                gy = torch.from_numpy(gy0.astype(np.float32)).to(DEV).to(dt)
                ins = [t] + ([d] if with_d else [])
                g1 = torch.autograd.grad([(y * gy).sum()], ins, create_graph=True)
                f = (g1[0].float() * torch.from_numpy(r1.astype(np.float32)).to(DEV)).sum()
                if with_d:
                    f = f + (g1[1] * torch.from_numpy(r2.astype(np.float32)).to(DEV)).sum()
                g2 = torch.autograd.grad([f], [t] + ([d] if with_d else []), allow_unused=True)
            res[closed] = [g.float() for g in g1] + [torch.zeros(1, device=DEV) if g is None else g.float() for g in g2]
        finally:
            grad_ops.CLOSED_TAIL_BACKWARD = True
    tol = 4e-3 if half else 2e-5
    for a, bb in zip(res[True], res[False]):
        assert rel_err(a.detach().cpu().numpy(), bb.detach().cpu().numpy()) < tol or float(bb.abs().max()) == 0.0


def test_phase_graphs_two_ranks_split_around_the_all_reduce():
    """More than one rank: a phase is two HIP graphs with the bucket all-reduce between them on the host side (train_stage.PhaseGraphs,
    ``split``).  Two ranks share the test box's one device (gloo; RCCL wants a device per rank), each with its own data: after seven
    iterations the split-graph form has the parameters of the eager loop (hook-launched reductions under backward) to round-off, and the
    two ranks hold bit-identical parameters in both forms (same averaged gradients, same optimiser arithmetic)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import copy, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SHG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SHG_ROOT"], "tests"))
import shgan_amd
from shgan_amd import losses, train_stage as ts
from test_gpu_train_graph import small_networks, real_batch, DEV
r = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["SHG_PORT"], rank=r, world_size=2)
G, D = small_networks(5)
g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
init = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone()
real4 = real_batch(4, 60 + r)                          # per-rank data
kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8, capturable=True)
order = [0, 1, 2, 4, 5, 8, 9]
out = []
for graphed in (False, True):
    G.load_state_dict(g0); D.load_state_dict(d0)
    torch.manual_seed(11 + r)
    L = losses.InpaintingLoss(DEV, G, D, composite_fake=True, noise_mode="const", style_mixing_prob=0)
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16, bucket_bytes=1 << 16)
    assert all(ph.sync is not None and ph.sync.reduce and len(ph.sync.buckets) > 1 for ph in phases)
    pg = ts.PhaseGraphs(phases, L, 4, 64, tuple(real4.shape), DEV) if graphed else None
    assert pg is None or pg.split
    for idx in order:
        pg.run(real4, idx) if graphed else ts.run_phases(real4, 64, phases, batch_idx=idx, loss=L, batch_gpu=4, device=DEV)
    torch.cuda.synchronize()
    if graphed:
        assert set(pg.graphs) == {"Gmain", "Greg", "Dmain"} and all(isinstance(g, tuple) and len(g) == 2 for g in pg.graphs.values())
    out.append(torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone())
    for ph in phases:
        ph.sync.remove()
worst = float(((out[0] - out[1]).abs().max() / out[0].abs().max()))
moved = float((out[1] - init).abs().max())
for form in out:                                       # both ranks hold the same parameters, bit for bit
    both = [torch.zeros_like(form.cpu()) for _ in range(2)]
    dist.all_gather(both, form.cpu())
    assert torch.equal(both[0], both[1]), "ranks diverged"
assert torch.isfinite(out[1]).all() and worst < 1e-4 and moved > 0, (worst, moved)
dist.destroy_process_group()
print("rank", r, "ok", "worst %.2e" % worst)
'''
    port = str(38500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), SHG_ROOT=root, SHG_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        o, _ = p.communicate(timeout=1200)
        assert p.returncode == 0 and f'rank {rank} ok'.encode() in o, o.decode()[-3000:]
        print(o.decode().strip().splitlines()[-1])
