"""Device time of the style side of one modulated layer (forward + first-order backward), fused kernels vs composed operators."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import shgan_amd
from shgan_amd.model_zoo import stylegan as sg
for n, i, o in ((8, 512, 512), (8, 1024, 512), (8, 128, 64), (16, 512, 512), (4, 512, 512)):
    s = (torch.randn(n, i, device='cuda') + 1).requires_grad_(True)
    w = (torch.rand(o, i, device='cuda') * 0.01).requires_grad_(True)
    a, b = torch.randn(n, i, device='cuda'), torch.randn(n, o, device='cuda')
    out = []
    for name, fn in (('fused', lambda: sg._StyleFactorsFn.apply(s, w, True)), ('composed', lambda: sg._style_factors_composed(True, s, w))):
        def step():
            sn, d = fn()
            torch.autograd.grad((sn * a).sum() + (d * b).sum(), [s, w])
        for _ in range(3):
            step()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(f'{name} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us')
    print(f'N {n:2d} I {i:4d} O {o:3d}: ' + '   '.join(out) + '   (forward + backward incl. the 4 loss ops, replayed HIP graph)', flush=True)
