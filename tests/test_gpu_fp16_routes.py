"""The three kernels behind the fp16 convolutions give the same bits: the persistent LDS-DMA ring kernel (3x3 stride 1, csrc/conv_f16_ring.hip)
and the merged-phase kernel of the stride-2 transposed form (csrc/conv_f16_upring.hip) against the gather kernel conv_f16_kernel they replace
(``shg_conv2d_f16_set_routes``), incl. the fused layer tail and the input scale -- and all of them against float64 torch convolutions of the
same half operands (reference: the cuDNN half kernels behind stylegan.py:136-138,172-181, conv2d_resample.py:125-137; torch is the oracle as
in tests/test_gpu_fp16.py, whose reference-generated fixtures run through these routes too)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CL = torch.channels_last


@pytest.fixture(scope='module')
def kf():
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels_f16
    return kernels_f16


def both_routes(kf, fn):
    lib = kf._lib.get_lib()
    old = lib.shg_conv2d_f16_set_routes(7)
    try:
        y_new = fn()
        lib.shg_conv2d_f16_set_routes(0)
        y_old = fn()
    finally:
        lib.shg_conv2d_f16_set_routes(old)
    torch.cuda.synchronize()
    return y_new, y_old


RING = [(1, 32, 64, 16, 32, {}), (2, 64, 64, 64, 64, dict(bias=True)), (1, 64, 64, 40, 72, dict(act=True)), (3, 128, 96, 40, 52, dict(act=True, bias=True, d=True, noise=1)),
        (2, 256, 256, 32, 36, dict(act=False, gain=0.5, d=True, noise=2)), (4, 512, 512, 16, 16, dict(act=True, bias=True, d=True, noise=2, clamp=0.7)),
        (2, 64, 64, 100, 132, dict(act=True, bias=True, noise=1)), (1, 32, 72, 21, 20, dict(act=True, bias=True, d=True)), (2, 96, 200, 19, 45, {}),
        (1, 32, 8, 7, 5, {}), (2, 256, 512, 33, 31, dict(bias=True)), (2, 64, 64, 256, 256, dict(act=True, bias=True))]


@pytest.mark.parametrize('n,i,o,h,w,t', RING)
def test_ring_kernel_equals_gather_kernel_and_float64(kf, n, i, o, h, w, t):
    torch.manual_seed(n * 77 + i + o + h)
    x = torch.randn(n, i, h, w, device=DEV).half().to(memory_format=CL)
    wt = (torch.randn(o, i, 3, 3, device=DEV) / (i * 9) ** 0.5).half()
    b = torch.randn(o, device=DEV) if t.get('bias') else None
    kw = {}
    if 'act' in t:
        kw.update(act=t['act'], gain=t.get('gain', 1.0), clamp=t.get('clamp', 256.0))
    if t.get('d'):
        kw['out_scale'] = torch.rand(n, o, device=DEV) + 0.5
    if t.get('noise') == 1:
        kw['noise'], kw['noise_strength'] = torch.randn(h, w, device=DEV), 0.3
    if t.get('noise') == 2:
        kw['noise'], kw['noise_strength'] = torch.randn(n, 1, h, w, device=DEV), 0.3
    y_new, y_old = both_routes(kf, lambda: kf.conv2d(x, wt, b, 1, 1, **kw))
    assert torch.equal(y_new, y_old)
    if not kw:                                   # plain convolution (+ bias): float64 torch on the same half operands
        ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), 1, 1)
        assert float((y_new.double() - ref).abs().max() / ref.abs().max()) < 2e-3


UP = [(1, 32, 8, 4, 4, 0, None, False), (2, 64, 64, 16, 16, 0, None, False), (1, 32, 72, 7, 5, 1, None, True), (3, 128, 96, 20, 33, 0, None, True),
      (2, 64, 40, 31, 32, 1, None, False), (2, 96, 64, 16, 31, 1, (32, 62), False), (1, 64, 64, 17, 40, 0, (33, 81), True),
      (2, 512, 256, 16, 16, 0, None, True), (2, 128, 64, 128, 128, 0, None, False), (4, 256, 128, 64, 64, 1, (128, 128), False)]


@pytest.mark.parametrize('n,i,o,h,w,pad,out_hw,sc', UP)
def test_merged_phase_transposed_kernel_equals_per_phase_launches_and_float64(kf, n, i, o, h, w, pad, out_hw, sc):
    torch.manual_seed(n * 100 + i + o + h + w)
    x = torch.randn(n, i, h, w, device=DEV).half().to(memory_format=CL)
    wt = (torch.randn(i, o, 3, 3, device=DEV) / (i * 9 / 4) ** 0.5).half()
    s = (torch.rand(n, i, device=DEV) + 0.5) if sc else None
    y_new, y_old = both_routes(kf, lambda: kf.conv_transpose2d(x, wt, None, pad, out_hw, in_scale=s))
    assert torch.equal(y_new, y_old)
    xs = x.double() if s is None else (x * s.half().reshape(n, i, 1, 1)).double()
    ref = F.conv_transpose2d(xs, wt.double(), stride=2, padding=0)
    oh, ow = out_hw if out_hw else (ref.shape[2] - 2 * pad, ref.shape[3] - 2 * pad)
    full = torch.zeros(n, o, pad + oh + 4, pad + ow + 4, dtype=torch.float64, device=DEV)
    full[:, :, :ref.shape[2], :ref.shape[3]] = ref
    ref = full[:, :, pad:pad + oh, pad:pad + ow]
    assert float((y_new.double() - ref).abs().max() / ref.abs().max()) < 2e-3


DOWN = [(1, 32, 128, 17, 17, {}), (2, 64, 128, 33, 65, dict(bias=True)), (1, 64, 72, 40, 73, dict(act=True, bias=True)), (3, 128, 256, 21, 33, dict(act=True, bias=True, d=True)),
        (2, 32, 8, 9, 9, dict(act=False, gain=0.5)), (2, 128, 512, 17, 17, dict(act=True, bias=True, clamp=0.7)), (2, 64, 128, 129, 129, dict(act=True, bias=True)),
        (1, 48, 40, 12, 20, dict(bias=True, pad=1)), (2, 96, 136, 31, 18, dict(act=True, pad=1))]


@pytest.mark.parametrize('n,i,o,h,w,t', DOWN)
def test_stride2_persistent_kernel_vs_gather_kernel_and_float64(kf, n, i, o, h, w, t):
    """csrc/conv_f16_down.hip sums the same products k-step-major (the gather kernel: 32-channel chunk -> tap -> k-step): not bit-identical;
    both are held against float64 torch, and against each other at half-precision rounding."""
    torch.manual_seed(n * 31 + i + o + h + w)
    pad = t.get('pad', 0)
    x = torch.randn(n, i, h, w, device=DEV).half().to(memory_format=CL)
    wt = (torch.randn(o, i, 3, 3, device=DEV) / (i * 9) ** 0.5).half()
    b = torch.randn(o, device=DEV) if t.get('bias') else None
    kw = {}
    if 'act' in t:
        kw.update(act=t['act'], gain=t.get('gain', 1.0), clamp=t.get('clamp', 256.0))
    if t.get('d'):
        kw['out_scale'] = torch.rand(n, o, device=DEV) + 0.5
    y_new, y_old = both_routes(kf, lambda: kf.conv2d(x, wt, b, 2, pad, **kw))
    d = (y_new.float() - y_old.float()).abs()
    assert float(d.max()) <= 2e-3 * float(y_old.float().abs().max()) and float((d > 0).float().mean()) < 0.05
    if not kw:
        ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), 2, pad)
        for y in (y_new, y_old):
            assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-3
