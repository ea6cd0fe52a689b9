"""Winograd-domain weight gradient of the stride-1 3x3 'same' layers (csrc/conv_wgrad_wino.hip: the transpose of F(4x4,3x3), both operands
transformed in the kernel) against torch's float64 weight gradient on the CPU (the reference's Conv2dGradWeight is a cuDNN call,
conv2d_gradfix.py:140-146: torch autograd of F.conv2d is the oracle, as in tests/test_gpu_backward.py) -- at that file's tolerance --
and against the direct kernel it replaces.  The second-order users (R1 / path length) are pinned by tests/test_gpu_config5.py, which runs
with the Winograd route on."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [
    # n, ci, co, h, w: ragged channel counts (blocks of 32 / 64), widths of one / several / partial chunks (8 tiles of 4 columns; 2 x 4
    # tiles below 32 columns), heights that are no multiple of the chunk or of the tile, one-tile-high images
    (1, 32, 64, 16, 16), (2, 8, 16, 16, 16), (3, 37, 70, 20, 24), (1, 64, 64, 64, 96), (2, 24, 40, 12, 16), (2, 33, 65, 18, 20),
    (1, 40, 24, 65, 128), (2, 12, 20, 16, 32), (2, 20, 36, 7, 16), (1, 4, 64, 32, 32), (3, 24, 70, 4, 16), (5, 64, 64, 8, 16),
    (3, 40, 33, 32, 32), (7, 16, 16, 9, 36), (3, 70, 24, 17, 28), (2, 32, 32, 33, 44), (3, 16, 24, 5, 100), (2, 24, 40, 34, 64),
    (3, 70, 64, 32, 32), (2, 16, 24, 33, 32), (1, 128, 96, 30, 40), (2, 96, 160, 16, 16), (1, 3, 5, 64, 64), (4, 65, 33, 24, 24),
    # images narrower than 16 columns: chunks of four tiles (2 x 2 at 8 .. 12 columns, 4 x 1 at 4)
    (3, 24, 70, 4, 4), (5, 64, 64, 8, 8), (8, 512, 512, 4, 4), (2, 40, 33, 12, 8), (7, 16, 16, 9, 12), (16, 96, 32, 8, 8), (2, 33, 70, 20, 4),
    # model shapes at reduced batch (the full ones are timed by tools/wgrad_bench.py)
    (1, 64, 64, 256, 256), (2, 128, 128, 128, 128), (2, 512, 512, 16, 16), (1, 256, 256, 64, 64),
]


@pytest.fixture(scope='module')
def kk():
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels
    return kernels


@pytest.mark.parametrize('n,ci,co,h,w', CASES)
def test_wgrad_wino_vs_float64_and_the_direct_kernel(kk, n, ci, co, h, w):
    rs = np.random.RandomState(n + ci + co + h + w)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((n, co, h, w)).astype(np.float32))
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), g.double(), stride=1, padding=1)
    xd, gd = x.to(DEV), g.to(DEV)
    assert kk.WGRAD_WINO and kk._lib.get_lib().shg_conv2d_wgrad_wino_supported(h, w, h, w, 3, 3, 1, 1) == 1
    got = kk.conv2d_wgrad(xd, gd, 3, 3, 1, 1)
    assert rel_err(got.cpu().numpy(), ref.numpy()) < 2e-5
    assert torch.equal(got, kk.conv2d_wgrad(xd, gd, 3, 3, 1, 1))               # deterministic (fixed-order slice reduction)
    kk.WGRAD_WINO = False
    try:
        direct = kk.conv2d_wgrad(xd, gd, 3, 3, 1, 1)
    finally:
        kk.WGRAD_WINO = True
    assert rel_err(direct.cpu().numpy(), ref.numpy()) < 2e-5
    assert rel_err(got.cpu().numpy(), direct.cpu().numpy()) < 2e-5


def test_wgrad_wino_geometry_gate(kk):
    lib = kk._lib.get_lib()
    assert lib.shg_conv2d_wgrad_wino_supported(64, 64, 64, 64, 3, 3, 1, 1) == 1
    for (h, w, oh, ow, k, s, p) in [(64, 64, 64, 64, 1, 1, 0), (65, 65, 32, 32, 3, 2, 0), (64, 64, 62, 62, 3, 1, 0), (64, 18, 64, 18, 3, 1, 1),
                                    (2, 16, 2, 16, 3, 1, 1)]:
        assert lib.shg_conv2d_wgrad_wino_supported(h, w, oh, ow, k, k, s, p) == 0
    # a geometry the Winograd form does not serve goes to the direct kernel, silently and correctly
    x, g = torch.randn(2, 16, 8, 6, device=DEV), torch.randn(2, 24, 8, 6, device=DEV)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (24, 16, 3, 3), g.double().cpu(), stride=1, padding=1)
    assert rel_err(kk.conv2d_wgrad(x, g, 3, 3, 1, 1).cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize('n,i,o,h,w', [(2, 4, 64, 32, 32), (3, 64, 3, 64, 64), (2, 128, 3, 32, 64), (1, 8, 8, 32, 32), (2, 5, 70, 64, 64), (8, 4, 64, 128, 128),
                                       (2, 3, 512, 32, 32), (2, 200, 1, 64, 32), (2, 4, 64, 30, 34), (1, 9, 9, 32, 32)])
def test_thin_1x1_weight_gradient_streaming_kernel_vs_float64(n, i, o, h, w):
    """fromRGB / toRGB weight gradients (1x1, one side of at most 8 channels: ``wgrad_thin_kernel``, csrc/conv_wgrad.hip -- the fat tensor
    is streamed once, per-wave dot products reduced across lanes, fixed-order sum of the pixel chunks) against torch's float64 weight
    gradient; the last two shapes stay on the MFMA-tile kernel (pixels not in whole 1024-chunks / both sides wider than 8)."""
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels
    g = torch.Generator(device='cpu').manual_seed(n * 131 + i * 7 + o)
    x = torch.randn(n, i, h, w, generator=g).to(DEV)
    gy = torch.randn(n, o, h, w, generator=g).to(DEV)
    ref = torch.nn.grad.conv2d_weight(x.double(), (o, i, 1, 1), gy.double())
    got = kernels.conv2d_wgrad(x, gy, 1, 1, 1, 0)
    assert tuple(got.shape) == (o, i, 1, 1)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    again = kernels.conv2d_wgrad(x, gy, 1, 1, 1, 0)
    assert torch.equal(got, again)                      # deterministic: fixed reduction order
