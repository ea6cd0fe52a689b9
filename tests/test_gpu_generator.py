"""GPU parity of the assembled generator (called through the reference's module API) against golden
vectors from the reference, the CPU oracle, and size-independent properties at BASELINE.json's sizes.
Bar (north_star): <= 1e-3 relative fp32 on the image, bit-exact uint8 on the known region."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def env():
    import shgan_amd  # noqa: F401
    from shgan_amd import eval_harness, kernels
    from test_host_logic import build_generator
    from oracle import shgan_oracle as orc
    return dict(build=build_generator, orc=orc, kernels=kernels, harness=eval_harness)


def make_G(env, resolution, sd, **kw):
    G = env['build'](resolution, **kw)
    G.load_state_dict(sd, strict=True)
    return G.eval().requires_grad_(False).to(DEV)


def c(a):
    return a.detach().cpu().numpy()


def test_generator_small_golden(env):
    g = load_golden('generator_small')
    orc = env['orc']
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seed']), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim,
                             w0_dim=w0_dim, noise_strength=0.1, bias_std=0.1)
    G = make_G(env, res, sd, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim)
    real = torch.from_numpy(g['real_u8'].astype(np.float32)) / 127.5 - 1.0
    n = real.shape[0]
    mask = torch.from_numpy(np.unpackbits(g['mask_bits'])[: n * res * res].reshape(n, 1, res, res).astype(np.float32))
    x = env['harness'].assemble_input(real, mask).to(DEV)
    z = torch.from_numpy(g['z']).to(DEV)
    cnd = torch.zeros(n, 0, device=DEV)
    ws = G.mapping(z, cnd)
    assert rel_err(c(ws), g['ws']) < 1e-5
    xg, feats = G.encoder(x)
    assert rel_err(c(xg), g['xg']) < 1e-4
    for r in (4, 8, 16, 32, 64):
        assert rel_err(c(feats[r]), g[f'feat{r}']) < 1e-4, r
    img = G(x=x, z=z, c=cnd, noise_mode='const')
    assert rel_err(c(img), g['img_const']) < 1e-3
    img_none = G(x=x, z=z, c=cnd, noise_mode='none')
    assert rel_err(c(img_none)[:, :, ::4, ::4], g['img_none_ds']) < 1e-3
    u8 = env['kernels'].composite_u8(x, img)
    m = mask.numpy().astype(bool)
    assert np.array_equal(np.where(m, c(u8), 0), np.where(m, g['comb_u8'], 0))          # known region: bit-exact
    d = np.abs(c(u8).astype(np.int32) - g['comb_u8'].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3                                        # hole: +-1 LSB at truncation edges
    # 'random' noise differs from 'const' but is finite and of the same scale
    img_r = G(x=x, z=z, c=cnd, noise_mode='random')
    assert torch.isfinite(img_r).all() and rel_err(c(img_r), c(img)) > 1e-4


def test_generator_small1024_golden(env):
    """The shipped 1024 configuration (configs/model/shgan.yaml:94-124) at reduced width against the reference's own run."""
    import hashlib
    g = load_golden('generator_small1024')
    orc = env['orc']
    res, ch_base, ch_max, w_dim, z_dim, w0_dim = [int(v) for v in g['cfg']]
    sd = orc.init_state_dict(res, seed=int(g['seeds'][0]), ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim,
                             noise_strength=0.1, bias_std=0.1)
    G = make_G(env, res, sd, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, z_dim=z_dim, w0_dim=w0_dim)
    assert G.num_ws == 18 and G.img_resolution == 1024
    x, z, real_u8, mask = orc.synthetic_batch(1, res, z_dim, seed=int(g['seeds'][1]))
    x, z = x.to(DEV), z.to(DEV)
    cnd = torch.zeros(1, 0, device=DEV)
    assert rel_err(c(G.mapping(z, cnd)), g['ws']) < 1e-5
    xg, feats = G.encoder(x)
    assert rel_err(c(xg), g['xg']) < 1e-4
    for r in (4, 16, 64):
        assert rel_err(c(feats[r]), g[f'feat{r}']) < 1e-4, r
    for r in (256, 512, 1024):
        f = feats[r]
        st = np.array([f.mean().item(), f.std().item(), f.min().item(), f.max().item()])
        assert np.allclose(st, g[f'feat{r}_stats'], rtol=1e-3, atol=1e-3), r
    img = G(x=x, z=z, c=cnd, noise_mode='const')
    assert rel_err(c(img)[:, :, ::4, ::4], g['img_ds']) < 1e-3
    assert rel_err(c(img).flatten()[g['sample_idx']], g['sample_val']) < 1e-3
    u8 = env['kernels'].composite_u8(x, img)
    known = c(u8) * mask.astype(np.uint8)
    assert hashlib.sha256(known.tobytes()).hexdigest() == str(g['known_sha256'])


def test_generator_full_width_1024(env):
    """shgan_g1024 at FULL width (32-channel 1024^2 layers, below the 64-channel tile of every convolution kernel: zero-padded
    operand layouts), batch 2 on the device, one image against the CPU oracle; known pixels exact, shard-invariant."""
    from shgan_amd import configs
    orc, hz = env['orc'], env['harness']
    assert configs.model_cfg('shgan_g1024')['args']['mapping']['args']['num_ws'] == 18
    sd = orc.init_state_dict(1024, seed=81, noise_strength=0.05)
    G = make_G(env, 1024, sd)
    x, z, real_u8, mask = hz.synthetic_batch(2, 1024, 512, seed=82, device=DEV, masks='bernoulli')
    cnd = torch.zeros(2, 0, device=DEV)
    a = G(x=x, z=z, c=cnd, noise_mode='const')
    assert tuple(a.shape) == (2, 3, 1024, 1024) and torch.isfinite(a).all()
    one = G(x=x[:1], z=z[:1], c=cnd[:1], noise_mode='const')
    assert rel_err(c(one), c(a[:1])) < 1e-4
    u8 = hz.run_generator(G, x, z, noise_mode='const')
    m = mask.astype(bool)
    assert np.array_equal(np.where(m, c(u8), 0), np.where(m, real_u8, 0))
    ref = orc.generator_forward(sd, x[:1].cpu(), z[:1].cpu(), 1024, noise_mode='const')
    assert rel_err(c(one), ref.numpy()) < 1e-3


def test_generator_full_width_256_golden(env):
    """BASELINE config 1 shape on the GPU: full-width 256x256, batch 2, weights from the seed."""
    g = load_golden('generator_full256_stats')
    orc = env['orc']
    sd = orc.init_state_dict(256, seed=int(g['seed']))
    G = make_G(env, 256, sd)
    x, z, _, _ = orc.synthetic_batch(2, 256, 512, seed=int(g['input_seed']))
    img = G(x=x.to(DEV), z=z.to(DEV), c=torch.zeros(2, 0, device=DEV), noise_mode='const')
    assert rel_err(c(img)[:, :, ::8, ::8], g['img_ds']) < 1e-3
    assert rel_err(c(img).flatten()[g['sample_idx']], g['sample_val']) < 1e-3
    st = np.array([img.mean().item(), img.std().item(), img.min().item(), img.max().item()])
    assert np.allclose(st, g['stats'], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('resolution,batch', [(256, 32), (512, 16)])
def test_full_size_properties(env, resolution, batch):
    """BASELINE configs 2/3 (full sizes): properties that need no full-size oracle run --
    determinism, batch-shard invariance, known-pixel exactness, plus a 1-image oracle check."""
    orc, hz = env['orc'], env['harness']
    sd = orc.init_state_dict(resolution, seed=41, noise_strength=0.05)
    G = make_G(env, resolution, sd)
    x, z, real_u8, mask = hz.synthetic_batch(batch, resolution, 512, seed=42, device=DEV, masks='bernoulli')
    cnd = torch.zeros(batch, 0, device=DEV)
    a = G(x=x, z=z, c=cnd, noise_mode='const')
    b = G(x=x, z=z, c=cnd, noise_mode='const')
    assert torch.equal(a, b)                                       # run-to-run bit-identical
    assert torch.isfinite(a).all()
    sub = G(x=x[:2], z=z[:2], c=cnd[:2], noise_mode='const')       # a 2-image shard of the same batch
    assert rel_err(c(sub), c(a[:2])) < 1e-4
    u8 = hz.run_generator(G, x, z, noise_mode='const')
    m = mask.astype(bool)
    assert np.array_equal(np.where(m, c(u8), 0), np.where(m, real_u8, 0))
    ref = orc.generator_forward(sd, x[:1].cpu(), z[:1].cpu(), resolution, noise_mode='const')
    assert rel_err(c(sub[:1]), ref.numpy()) < 1e-3


def test_winograd_and_direct_paths_agree_full_size(env):
    """The stride-1 3x3 layers run on the Winograd F(2x2,3x3) kernel by default; the direct implicit-GEMM kernel is
    the same function (SHG_WINO=0).  Full-width 256x256 generator, both routes, same weights and inputs."""
    from shgan_amd import kernels as kk
    orc, hz = env['orc'], env['harness']
    sd = orc.init_state_dict(256, seed=61, noise_strength=0.05)
    G = make_G(env, 256, sd)
    x, z, _, _ = hz.synthetic_batch(4, 256, 512, seed=62, device=DEV, masks='bernoulli')
    cnd = torch.zeros(4, 0, device=DEV)
    old = kk.WINO
    try:
        kk.WINO = True
        a = G(x=x, z=z, c=cnd, noise_mode='const')
        kk.WINO = False
        b = G(x=x, z=z, c=cnd, noise_mode='const')
    finally:
        kk.WINO = old
    assert not torch.equal(a, b)                 # two different algorithms really ran
    assert rel_err(c(a), c(b)) < 1e-5


def test_mapping_truncation_and_batch_shapes(env):
    """Mapping.forward (stylegan.py:394-430): truncation is the lerp towards w_avg of the untruncated ws (also with a
    cutoff), and ragged batch sizes (1, 3, 17 > the 16-row slab of the dense kernel) give the rows of the full batch."""
    orc = env['orc']
    kw = dict(ch_base=1024, ch_max=32, w_dim=64, z_dim=48, w0_dim=96)
    sd = orc.init_state_dict(256, seed=71, **kw)
    sd['mapping.w_avg'] = torch.linspace(-1, 1, 64)
    G = make_G(env, 256, sd, **kw)
    z = torch.randn(17, 48, generator=torch.Generator().manual_seed(5)).to(DEV)
    cnd = torch.zeros(17, 0, device=DEV)
    ws = G.mapping(z, cnd)
    assert tuple(ws.shape) == (17, G.num_ws, 64)
    ref = orc.mapping(sd, z.cpu(), G.num_ws)
    assert rel_err(c(ws), ref.numpy()) < 1e-5
    w_avg = sd['mapping.w_avg'].to(DEV)
    wt = G.mapping(z, cnd, truncation_psi=0.7)
    assert rel_err(c(wt), c(w_avg.lerp(ws, 0.7))) < 1e-6
    wc = G.mapping(z, cnd, truncation_psi=0.5, truncation_cutoff=3)
    exp = ws.clone(); exp[:, :3] = w_avg.lerp(ws[:, :3], 0.5)
    assert rel_err(c(wc), c(exp)) < 1e-6
    for n in (1, 3):
        assert rel_err(c(G.mapping(z[:n], cnd[:n])), c(ws[:n])) < 1e-6


def test_sharded_eval_matches_unsharded(env):
    """Index path of the eval loop on one GPU: emulated 2-rank shards re-interleave to the 1-rank result."""
    orc, hz = env['orc'], env['harness']
    sd = orc.init_state_dict(256, seed=51, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = make_G(env, 256, sd, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    ids1, out1 = hz.sharded_eval(G, 5, 2, 256, rank=0, world=1, seed=3, gather=False, device=DEV)
    parts = [hz.sharded_eval(G, 5, 2, 256, rank=r, world=2, seed=3, gather=False, device=DEV) for r in range(2)]
    from shgan_amd.data import zipzap_arrange
    order = zipzap_arrange([p[0] for p in parts])[:5]
    merged = zipzap_arrange([c(p[1]) for p in parts])[:5]
    assert order == ids1 == list(range(5))
    # per-item inputs are identical; batch composition differs -> +-1 LSB at most (batch-global style norm, 8e)
    d = np.abs(merged.astype(np.int32) - c(out1).astype(np.int32))
    assert d.max() <= 1


def test_stream_pipeline_matches_plain_loop_bit_exact(env):
    """The evaluation loop issues consecutive batches round-robin on several HIP streams (eval_harness.StreamPipeline): same
    bytes as the one-stream loop, for every depth, with the batches' inputs built on the caller's stream in between."""
    orc, hz = env['orc'], env['harness']
    sd = orc.init_state_dict(256, seed=52, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    G = make_G(env, 256, sd, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128)
    ref_ids, ref = hz.sharded_eval(G, 13, 3, 256, seed=4, gather=False, device=DEV, pipeline_depth=1)
    for depth in (2, 3, 5):
        ids, out = hz.sharded_eval(G, 13, 3, 256, seed=4, gather=False, device=DEV, pipeline_depth=depth)
        assert ids == ref_ids
        assert torch.equal(out, ref), depth
    # the helper itself: results of every run() are valid after join(), whatever the stream they were computed on
    pipe = hz.StreamPipeline(DEV, depth=3)
    xs = [torch.full((1 << 20,), float(k), device=DEV) for k in range(7)]
    outs = [pipe.run(lambda t: (t * 2 + 1).cumsum(0)[-1:], x) for x in xs]
    pipe.join()
    assert [float(o) for o in outs] == [float((2 * k + 1) * (1 << 20)) for k in range(7)]
