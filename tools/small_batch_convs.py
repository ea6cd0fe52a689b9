"""The 512-channel layers at 64^2 .. 16^2 for batches 2 .. 16: where a kernel's grid no longer fills the chip (the path-length pass runs
on batch / 2 = 4 images).  us per launch and direct-form TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import kernels
from shgan_amd.model_zoo.stylegan_utils import upfirdn2d as ufd


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


w = torch.randn(512, 512, 3, 3, device='cuda') / 68.0
pw = kernels.conv_weight_prep(w)
pwt = kernels.conv_weight_prep(w.transpose(0, 1).contiguous())
for res in ((16, 8, 4) if '--small' in sys.argv else (64, 32, 16, 8, 4)):
    for n in (2, 4, 8, 16):
        x = torch.randn(n, 512, res, res, device='cuda')
        xs2 = torch.randn(n, 512, res + 1, res + 1, device='cuda')
        f = 2.0 * n * 512 * 512 * 9
        a = t(lambda: kernels.conv2d(x, pw, mode=kernels.MODE_SAME, pad=1))
        b = t(lambda: kernels.conv2d(xs2, pw, mode=kernels.MODE_DOWN2, pad=0))
        c = t(lambda: kernels.conv2d(x, pwt, mode=kernels.MODE_UP2T, planar=True))
        print(f'{res:3d}^2 x {n:2d}: 3x3 s1 {a:7.1f} us {f * res * res / a / 1e6:6.1f} TF | s2 ({res + 1} -> {res // 2}) {b:7.1f} us {f * (res // 2) ** 2 / b / 1e6:6.1f} TF | '
              f'up ({res} -> {2 * res + 1}, planes) {c:7.1f} us {f * res * res / c / 1e6:6.1f} TF')
