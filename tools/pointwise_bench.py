"""Streaming passes of the float32 (NCHW) training / inference path: bias + activation forward / backward, the fused modulation tail, channel
scaling.  GB/s = algorithmic read + write bytes / time.  usage: [SHG_VARIANT=<tag>] python tools/pointwise_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import kernels as kk

dev = 'cuda:0'


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


for (n, c, r) in [(8, 64, 512), (8, 128, 256), (8, 256, 128), (8, 512, 64), (8, 512, 16), (16, 64, 512)]:
    x = torch.randn(n, c, r, r, device=dev)
    b, d, nz = torch.randn(c, device=dev), torch.rand(n, c, device=dev) + 0.5, torch.randn(n, 1, r, r, device=dev)
    by = 4.0 * x.numel()
    t1 = timeit(lambda: kk.bias_act(x, b))
    y = kk.bias_act(x, b)
    t2 = timeit(lambda: kk.bias_act_backward(x, y))
    t3 = timeit(lambda: kk.bias_act(x, b, scale=d, noise=nz, noise_strength=0.3))
    t4 = timeit(lambda: kk.modtail_backward(x, y, x, d, want_sums=True, want_noise=True))
    t5 = timeit(lambda: kk.scale_channels(x, d))
    print(f'[{n},{c},{r},{r}]  bias_act {t1:7.1f} us {2 * by / t1 / 1e3:6.0f} GB/s | backward {t2:7.1f} us {3 * by / t2 / 1e3:6.0f} | fused tail {t3:7.1f} us {2 * by / t3 / 1e3:6.0f} | '
          f'tail backward {t4:7.1f} us {4 * by / t4 / 1e3:6.0f} | scale_channels {t5:7.1f} us {2 * by / t5 / 1e3:6.0f}')
