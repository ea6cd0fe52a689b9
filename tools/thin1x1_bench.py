"""conv1x1_thin_in (fromRGB 4 -> 64 and the toRGB input gradient 3 -> C) at every resolution of the 512 networks, batch 8: us and GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i, o, r in [(4, 64, 512), (3, 64, 512), (3, 128, 256), (3, 256, 128), (3, 512, 64), (3, 512, 32), (3, 512, 16), (3, 512, 8), (3, 512, 4)]:
    x, w = torch.randn(n, i, r, r, device='cuda'), torch.randn(o, i, device='cuda')
    for _ in range(3):
        kernels.conv1x1_thin_in(x, w, None, act=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = kernels.conv1x1_thin_in(x, w, None, act=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'{i} -> {o:3d} @ {r:3d}^2 x {n}: {us:8.1f} us  {4 * (x.numel() + y.numel()) / us / 1e3:8.1f} GB/s')
