import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import shgan_amd
from shgan_amd import kernels_f16
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for shape in ((8, 512, 64, 64), (8, 512, 32, 32), (16, 512, 32, 32), (16, 512, 64, 64), (8, 64, 512, 512)):
    x = torch.randn(shape, device='cuda'); h = x.to(dtype=torch.float16, memory_format=torch.channels_last)
    gb = x.numel() * 6 / 1e9
    a, b = t(lambda: kernels_f16.relayout(x)), t(lambda: x.to(dtype=torch.float16, memory_format=torch.channels_last))
    c, d = t(lambda: kernels_f16.relayout(h)), t(lambda: h.to(dtype=torch.float32, memory_format=torch.contiguous_format))
    print(f'{str(shape):22s} to half: kernel {a:7.1f} us ({gb / a * 1e3:5.2f} TB/s)  torch {b:7.1f} us | to float: kernel {c:7.1f} us ({gb / c * 1e3:5.2f} TB/s)  torch {d:7.1f} us', flush=True)
