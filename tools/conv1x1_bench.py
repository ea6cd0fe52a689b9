"""1x1 convolutions of the critic's skip branch / the layers' 1x1 forms (float32, conv_mfma_kernel<1, 32, ...>): us, TFLOP/s and the HBM time of
the tensors (x read once, y written once) at 5.2 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import kernels


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (8, 16):
    for i, o, r in [(64, 128, 256), (128, 256, 128), (256, 512, 64), (512, 512, 32), (512, 512, 16), (64, 64, 512), (128, 128, 256)]:
        x = torch.randn(n, i, r, r, device='cuda')
        pw = kernels.conv_weight_prep(torch.randn(o, i, 1, 1, device='cuda'))
        us = t(lambda: kernels.conv2d(x, pw, mode=kernels.MODE_SAME, pad=0))
        fl, by = 2.0 * n * o * i * r * r, 4.0 * n * (i + o) * r * r
        print(f'1x1 {i:3d} -> {o:3d} @ {r:3d}^2 x {n:2d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF  | tensors {by / 1e6:7.1f} MB = {by / 5.2e6:6.1f} us at 5.2 TB/s | MFMA floor {fl / 157.3e6:6.1f} us')
