"""Per-shape timing of the fp16 convolution family (csrc/conv_f16.hip): forward, transposed, weight gradient.  usage: python tools/conv_f16_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):        # A/B knobs: python sh-gan_amd/build.py --variant=<tag> -D...
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import kernels_f16 as kf

dev = 'cuda:0'
CL = torch.channels_last


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = 8
FWD_ONLY = bool(os.environ.get('SHG_F16_FWD_ONLY'))     # ablation runs (tools/f16_abl.sh): forward shapes only
for (i, o, r, k, s) in [(64, 64, 512, 3, 1), (128, 128, 256, 3, 1), (256, 256, 128, 3, 1), (512, 512, 64, 3, 1), (64, 128, 513, 3, 2), (128, 256, 257, 3, 2),
                         (64, 64, 512, 1, 1), (512, 512, 64, 1, 1)]:
    x = torch.randn(N, i, r, r, device=dev).half().to(memory_format=CL)
    w = (torch.randn(o, i, k, k, device=dev) / (i * k * k) ** 0.5).half()
    pad = 1 if (k == 3 and s == 1) else 0
    y = kf.conv2d(x, w, None, s, pad)
    fl = 2.0 * N * o * i * k * k * y.shape[2] * y.shape[3]
    by = 2.0 * (x.numel() + y.numel())
    t = timeit(lambda: kf.conv2d(x, w, None, s, pad))
    g = torch.randn_like(y)
    tw = 0.0 if FWD_ONLY else timeit(lambda: kf.conv2d_wgrad(x, g, k, s, pad))
    print(f'conv {i:4d}->{o:4d} {r:4d}^2 k{k} s{s}: fwd {t:8.1f} us {fl / t / 1e6:7.1f} TFLOP/s {by / t / 1e3:7.1f} GB/s | wgrad {tw:8.1f} us {fl / max(tw, 1e-9) / 1e6:7.1f} TFLOP/s')
if FWD_ONLY:
    sys.exit(0)
for (i, o, r) in [(128, 64, 256), (256, 128, 128), (512, 256, 64)]:
    x = torch.randn(N, i, r, r, device=dev).half().to(memory_format=CL)
    w = (torch.randn(i, o, 3, 3, device=dev) / (i * 9) ** 0.5).half()
    y = kf.conv_transpose2d(x, w, None, 0)
    fl = 2.0 * N * o * i * 9 * r * r
    t = timeit(lambda: kf.conv_transpose2d(x, w, None, 0))
    print(f'convT {i:4d}->{o:4d} {r:4d}^2: {t:8.1f} us {fl / t / 1e6:7.1f} TFLOP/s {2.0 * (x.numel() + y.numel()) / t / 1e3:7.1f} GB/s')
x = torch.randn(N, 64, 513, 513, device=dev).half().to(memory_format=CL)
f = torch.tensor([1., 3., 3., 1.], device=dev)
f = torch.outer(f, f) / 64
t = timeit(lambda: kf.upfirdn2d(x, f, padx0=2, padx1=2, pady0=2, pady1=2))
print(f'upfirdn2d pad2 64ch 513^2: {t:8.1f} us {2.0 * 2 * x.numel() / t / 1e3:7.1f} GB/s')
for (c, r) in [(64, 512), (128, 256)]:
    xx = torch.randn(N, c, r, r, device=dev).half().to(memory_format=CL)
    t = timeit(lambda: kf.upfirdn2d(xx, f, downx=2, downy=2, padx0=1, padx1=1, pady0=1, pady1=1))
    print(f'upfirdn2d down 2 {c}ch {r}^2: {t:8.1f} us {2.0 * 1.25 * xx.numel() / t / 1e3:7.1f} GB/s')
    xh = xx[:, :, ::2, ::2].contiguous(memory_format=CL)
    t = timeit(lambda: kf.upfirdn2d(xh, f, upx=2, upy=2, padx0=2, padx1=1, pady0=2, pady1=1, gain=4.0))
    print(f'upfirdn2d up 2 {c}ch {r // 2}^2: {t:8.1f} us {2.0 * 5 * xh.numel() / t / 1e3:7.1f} GB/s')
for (c, r) in [(64, 513), (128, 257), (256, 128)]:
    x = torch.randn(N, c, r, r, device=dev).half().to(memory_format=CL)
    b, d = torch.randn(c, device=dev), torch.rand(N, c, device=dev) + 0.5
    nz = torch.randn(N, 1, r, r, device=dev)
    t = timeit(lambda: kf.bias_act(x, b))
    print(f'bias_act {c}ch {r}^2: {t:8.1f} us {2.0 * 2 * x.numel() / t / 1e3:7.1f} GB/s')
    y = kf.bias_act(x, b)
    t = timeit(lambda: kf.bias_act_backward(x, y))
    print(f'bias_act_backward {c}ch {r}^2: {t:8.1f} us {2.0 * 3 * x.numel() / t / 1e3:7.1f} GB/s')
    t = timeit(lambda: kf.modtail(x, d, nz, b, act=True))
    print(f'modtail {c}ch {r}^2: {t:8.1f} us {2.0 * 2 * x.numel() / t / 1e3:7.1f} GB/s')
    y = kf.modtail(x, d, nz, b, act=True)
    t = timeit(lambda: kf.modtail_backward(x, y, x, d, want_sums=True, want_noise=True, act=True))
    print(f'modtail_backward {c}ch {r}^2: {t:8.1f} us {2.0 * 4 * x.numel() / t / 1e3:7.1f} GB/s')
    t = timeit(lambda: kf.upfirdn2d(x, f, padx0=2, padx1=1, pady0=2, pady1=1))
    print(f'upfirdn2d same-size {c}ch {r}^2: {t:8.1f} us {2.0 * 2 * x.numel() / t / 1e3:7.1f} GB/s')
