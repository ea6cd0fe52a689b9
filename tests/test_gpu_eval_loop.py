"""GPU: the streamed evaluation loop (``eval_harness.EvalLoop`` = lib/experiments/shgan_default.py:264-300, BASELINE config 4) on the
product kernels -- uint8 hand-off, device masks, three-stream G + composite INTO the result buffer, FID moments on the statistics
stream, end-of-run collectives in a 1-rank RCCL group -- against the plain per-batch calls it replaces."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def small_g():
    import shgan_amd  # noqa: F401
    from shgan_amd import configs
    G = configs.seeded_init_(configs.build_generator(256, ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128), seed=5, noise_strength=0.1,
                             bias_std=0.1)
    return G.eval().requires_grad_(False).to(DEV)


def _latents(ids, b, z_dim=64):
    out = torch.empty(b, z_dim)
    g = torch.Generator()
    for k, i in enumerate(ids):
        g.manual_seed(500 + int(i))
        out[k].normal_(generator=g)
    return out.to(DEV)


def test_u8_hand_off_is_bit_identical_to_the_host_formatter_route():
    """uint8 pixels -> x = cat([mask - .5, real * mask]) in one kernel == ToTensor (/255), *2-1 (ds_ffhq.py:318-326,338) on the host, then
    the float kernel / the reference's cat (shgan_default.py:267-274)."""
    import shgan_amd  # noqa: F401
    from shgan_amd import eval_harness as hz, kernels
    rs = np.random.RandomState(3)
    u8 = torch.from_numpy(rs.randint(0, 256, size=(3, 3, 64, 96)).astype(np.uint8))
    u8[0, 0, 0, :256 % 96] = 0
    u8.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)          # every code occurs
    mask = torch.from_numpy((rs.rand(3, 1, 64, 96) < 0.6).astype(np.float32))
    real = u8.to(torch.float32).div(255) * 2 - 1
    want = torch.cat([mask - 0.5, real * mask], dim=1)
    got = hz.assemble_input(u8.to(DEV), mask.to(DEV))
    assert torch.equal(got.cpu(), want)
    assert torch.equal(hz.assemble_input(u8, mask), want)             # the host branch takes decoded pixels too
    assert torch.equal(kernels.assemble_input(real.to(DEV), mask.to(DEV)).cpu(), want)
    from shgan_amd import _lib
    with pytest.raises(_lib.ShgError):
        kernels.assemble_input(u8.to(DEV), mask.to(DEV), lut=torch.zeros(10, device=DEV))


def test_composite_writes_into_a_caller_buffer(small_g):
    from shgan_amd import eval_harness as hz, kernels, _lib
    x, z, _, _ = hz.synthetic_batch(2, 256, 64, seed=6, device=DEV, masks='bernoulli')
    ref = hz.run_generator(small_g, x, z, noise_mode='const')
    buf = torch.zeros(5, 3, 256, 256, dtype=torch.uint8, device=DEV)
    out = hz.run_generator(small_g, x, z, noise_mode='const', out=buf[2:4])
    assert out.data_ptr() == buf[2:4].data_ptr() and torch.equal(buf[2:4], ref) and int(buf[:2].max()) == 0 and int(buf[4:].max()) == 0
    with pytest.raises(_lib.ShgError):
        kernels.composite_u8(x, torch.zeros(2, 3, 256, 256, device=DEV), out=torch.zeros(2, 3, 256, 256, device=DEV))


@pytest.mark.parametrize('depth', [1, 3])
def test_eval_loop_equals_the_plain_per_batch_calls(small_g, depth):
    """Same ids, same numpy mask draws, same latents, noise_mode 'const': the loop's result buffer == run_generator batch by batch (bit for
    bit), its moments == FidStats fed the same features in one go, a ragged last batch and an ``on_batch`` consumer included."""
    from shgan_amd import eval_harness as hz, masks
    from shgan_amd.fid_stats import FidStats
    n_items, b, R = 11, 4, 256
    feats_fn = lambda u8: hz.standin_features(u8, 128)        # noqa: E731
    seen = []
    loop = hz.EvalLoop(small_g, DEV, R, n_items, noise_mode='const', depth=depth, feature_fn=feats_fn, fid_dim=128, latent_fn=_latents,
                       on_batch=lambda ids, out, ev: seen.append((list(ids), out.data_ptr(), ev)))
    np.random.seed(77)
    loop.run(hz.PinnedU8Loader(loop.ids, b, R, seed=9))
    images, fid = loop.gather()
    torch.cuda.synchronize()
    assert loop.ids == list(range(n_items)) and [i for ids, _, _ in seen for i in ids] == loop.ids and all(ev is not None for _, _, ev in seen)
    # the plain route
    np.random.seed(77)
    outs, feats = [], []
    for img, ids in hz.PinnedU8Loader(list(range(n_items)), b, R, seed=9):
        m = masks.random_masks(len(ids), R, [0, 1], device=DEV)
        x = hz.assemble_input((img.to(torch.float32).div(255) * 2 - 1).to(DEV), m)
        outs.append(hz.run_generator(small_g, x, _latents(ids, len(ids)), noise_mode='const'))
        feats.append(feats_fn(outs[-1]))
    want = torch.cat(outs)
    assert images.dtype == torch.uint8 and tuple(images.shape) == (n_items, 3, R, R)
    assert torch.equal(images, want), int((images != want).sum())
    ref = FidStats(128, device=DEV)
    ref.add(torch.cat(feats))
    n, mu, sg = fid.mean_cov()
    n0, mu0, sg0 = ref.mean_cov()
    assert n == n0 == n_items and np.allclose(mu, mu0, rtol=0, atol=1e-10) and np.allclose(sg, sg0, rtol=0, atol=1e-8)


def test_eval_loop_collectives_in_a_one_rank_rccl_group(small_g):
    """The end-of-run collectives on the RCCL backend itself (a 1-rank ``nccl`` group: all_gather_into_tensor of the uint8 results,
    all_reduce of the moments) + ``broadcast_state`` -- the same calls N ranks make."""
    import torch.distributed as dist
    from shgan_amd import eval_harness as hz
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(37500 + os.getpid() % 2000)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        before = {k: v.clone() for k, v in small_g.state_dict().items()}
        nbytes = hz.broadcast_state(small_g, src=0)
        assert nbytes >= sum(p.numel() for p in small_g.parameters()) * 4
        assert all(torch.equal(v, before[k]) for k, v in small_g.state_dict().items())
        loop = hz.EvalLoop(small_g, DEV, 256, 6, noise_mode='const', feature_fn=lambda u8: hz.standin_features(u8, 64), fid_dim=64, latent_fn=_latents)
        np.random.seed(5)
        loop.run(hz.PinnedU8Loader(loop.ids, 4, 256, seed=2))
        images, fid = loop.gather()
        torch.cuda.synchronize()
        assert torch.equal(images, loop.images) and fid.mean_cov()[0] == 6
        with pytest.raises(ValueError):
            hz.EvalLoop(small_g, DEV, 256, 6, noise_mode='const').gather()          # nothing processed yet
    finally:
        dist.destroy_process_group()


def test_eval_loop_full_width_512_batch16_random_noise():
    """One rank's share of BASELINE config 4 at full width: three 16-image batches through the loop with the reference's settings
    (noise_mode 'random', z ~ N(0,1)); known pixels of every result equal the loader's bytes (the composite is exact there) and the
    moments count every sample."""
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, eval_harness as hz
    G = configs.seeded_init_(configs.build_generator(512), seed=0).eval().requires_grad_(False).to(DEV)
    kept = {}
    loop = hz.EvalLoop(G, DEV, 512, 48, noise_mode='random', feature_fn=hz.standin_features)
    loader = hz.PinnedU8Loader(loop.ids, 16, 512, seed=4)
    np.random.seed(11)
    loop.run(loader)
    images, fid = loop.gather()
    torch.cuda.synchronize()
    assert fid.mean_cov()[0] == 48
    # u8 -> float -> (x*127.5+127.5) truncation returns the loader's byte where the mask keeps the pixel: check on a re-drawn mask set
    from shgan_amd import masks
    np.random.seed(11)
    k0 = 0
    for img, ids in hz.PinnedU8Loader(loop.ids, 16, 512, seed=4):
        m = masks.random_masks(len(ids), 512, [0, 1], device=DEV).bool()
        got = images[k0:k0 + len(ids)]
        src = img.to(DEV)
        assert torch.equal(torch.where(m, got, torch.zeros_like(got)), torch.where(m, src, torch.zeros_like(src))), k0
        k0 += len(ids)
    del kept
