"""Times the two SHU kernels at the generator's shape (N=16, C=32, 64x64) and reports traffic / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import kernels as kk
from oracle import shgan_oracle as orc
N, C = 16, 32
x = torch.randn(N, C, 64, 64, device='cuda')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = kk.shu_rfft2_shift(x)
us = timeit(lambda: kk.shu_rfft2_shift(x))
by = (x.numel() + t.numel()) * 4
print(f'shu_rfft2_shift      {us:7.1f} us  {by/1e6:6.1f} MB  {by/us/1e6:6.3f} TB/s')
ref = torch.fft.rfft2(x.cpu().double(), norm='forward')
ref = torch.roll(ref, 31, dims=2)
got = torch.complex(t[:, :C].cpu().double(), t[:, C:].cpu().double())
print('   max abs err vs torch.fft (float64):', float((got - ref).abs().max()), ' ref max', float(ref.abs().max()))
cw = orc.make_cweight_closed_form().cuda()
gt = orc.gaussian_split_tables()
gauss = [gt[r].cuda().contiguous() if isinstance(gt, dict) else gt[i].cuda().contiguous() for i, r in enumerate((4, 8, 16, 32, 64))]
y = torch.randn(N, 2 * C * 6, 64, 33, device='cuda')
outs = [torch.zeros(N, C, r, r, device='cuda') for r in (4, 8, 16, 32, 64)]
fn = lambda: kk.shu_split_irfft2(y, cw, gauss, outs, False)
try:
    us = timeit(fn)
    by = (y.numel() + sum(o.numel() for o in outs)) * 4
    print(f'shu_split_irfft2     {us:7.1f} us  {by/1e6:6.1f} MB  {by/us/1e6:6.3f} TB/s')
except Exception as e:
    print('split bench skipped:', repr(e)[:200])
