"""Mapping-network chain (8 dense layers [16,512] -> [16,512] + activation) and the grouped style kernels in a replayed HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):
    _lib.use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import kernels as kk
x = torch.randn(16, 512, device='cuda')
ws = [torch.randn(512, 512, device='cuda') * 0.04 for _ in range(8)]
bs = [torch.randn(512, device='cuda') for _ in range(8)]
def chain():
    y = x
    for w, b in zip(ws, bs):
        y = kk.dense(y, w, b, act=True)
    return y
for _ in range(3): chain()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): chain()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print(f'dense chain of 8 layers: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  ({e0.elapsed_time(e1) / 400 * 1e3:.1f} us per layer)')
