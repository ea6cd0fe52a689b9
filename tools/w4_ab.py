"""A/B timing of the F(4x4) convolution kernel on the generator's stride-1 layer shapes (batch 16): one process per library
(SHG_VARIANT=<tag> selects tools/_variants/libshgan_hip_<tag>.so, default = the product library).  Box-to-box clock differences are
+-5 %, so variants are only comparable inside ONE gpurun call; run the tags interleaved and twice.
usage: [SHG_VARIANT=tag] python tools/w4_ab.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib, kernels as kk
tag = os.environ.get('SHG_VARIANT', '')
if tag:
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % tag))
N = 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
LAYERS = [('enc512', 64, 512, False), ('syn512', 64, 512, True), ('enc256', 128, 256, False), ('syn256', 128, 256, True),
          ('enc128', 256, 128, False), ('enc64', 512, 64, False), ('syn64', 512, 64, True), ('enc32', 512, 32, False)]
out = []
tot = 0.0
for name, ch, h, syn in LAYERS:
    x = torch.randn(N, ch, h, h, device='cuda')
    pw = kk.conv_weight_prep(torch.randn(ch, ch, 3, 3, device='cuda'), demod=syn)
    kw = dict(bias=torch.randn(ch, device='cuda'), act=True)
    if syn:
        kw.update(in_scale=torch.rand(N, ch, device='cuda') + 0.5, out_scale=torch.rand(N, ch, device='cuda') + 0.5,
                  noise=torch.randn(N, 1, h, h, device='cuda'), noise_strength=0.1, residual=torch.randn(N, ch, h, h, device='cuda'))
    for _ in range(3):
        y = kk.conv2d(x, pw, mode=0, pad=1, **kw)
    best = 1e9
    for _trial in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = kk.conv2d(x, pw, mode=0, pad=1, **kw)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    out.append(f'{name} {best * 1e3:7.1f}')
    tot += best * (1 if h == 32 else 2 if h >= 256 else 1)
    del x, y, kw
print(f'{tag or "product":10s} ' + ' | '.join(out) + f' | sum {tot:.3f} ms', flush=True)
