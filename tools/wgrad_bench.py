"""Weight-gradient kernels (shg_conv2d_wgrad_f32; shg_conv2d_wgrad_wino_f32 where it applies) on the layer shapes of the 256x256 / 512x512 models,
batch 8 (16 with --batch 16: the critic's stacked pass).  TFLOP/s are direct-form (the Winograd route executes a quarter of them).
SHG_WGRAD_DIRECT=1: the direct kernel everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import kernels as kk
N = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 8
if os.environ.get('SHG_WGRAD_DIRECT'):
    kk.WGRAD_WINO = False
CASES = [('3x3 s1 64ch 512^2', 64, 64, 512, 3, 1, 1), ('3x3 s1 128ch 256^2', 128, 128, 256, 3, 1, 1), ('3x3 s1 256ch 128^2', 256, 256, 128, 3, 1, 1),
         ('3x3 s1 512ch 64^2', 512, 512, 64, 3, 1, 1), ('3x3 s1 512ch 32^2', 512, 512, 32, 3, 1, 1), ('3x3 s1 512ch 16^2', 512, 512, 16, 3, 1, 1),
         ('3x3 s1 512ch 8^2', 512, 512, 8, 3, 1, 1), ('3x3 s1 512ch 4^2', 512, 512, 4, 3, 1, 1),
         ('3x3 s2 128->256 257->128', 128, 256, 257, 3, 2, 0), ('3x3 s2 512ch 65->32', 512, 512, 65, 3, 2, 0),
         ('convT as s2: g 256->128ch 257', 128, 256, 257, 3, 2, 0), ('1x1 s1 128->3 256^2', 128, 3, 256, 1, 1, 0), ('1x1 s1 4->128 256^2', 4, 128, 256, 1, 1, 0)]
tot = 0.0
for name, ci, co, h, k, s, p in CASES:
    oh = (h + 2 * p - k) // s + 1
    x = torch.randn(N, ci, h, h, device='cuda')
    g = torch.randn(N, co, oh, oh, device='cuda')
    for _ in range(2):
        kk.conv2d_wgrad(x, g, k, k, s, p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        kk.conv2d_wgrad(x, g, k, k, s, p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * N * co * ci * k * k * oh * oh
    tot += ms
    print(f'{name:34s} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s', flush=True)
print(f'TOTAL {tot:.2f} ms')
