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

# whole SHU forward (rfft2 -> spectral stage -> split + irfft2), fused spectral kernel vs two 1x1 convolutions
from shgan_amd.model_zoo import shgan
shu = shgan.SHU(32, 32, [2, 3], 'piecewise_linear', input_res=64, lowest_res=4, tail_sigma_mult=3).cuda().eval()
for fused in (True, False):
    shgan.SHU.FUSED_SPECTRAL = fused
    us = timeit(lambda: shu(x))
    print(f'SHU.forward N={N} fused_spectral={fused}: {us:7.1f} us (incl. host launch overhead of {3 if fused else 4} launches + 5 output allocations)')
shgan.SHU.FUSED_SPECTRAL = True
t = kk.shu_rfft2_shift(x)
w0p, b0, w1p = shu._packed()
us = timeit(lambda: kk.shu_spectral(t, w0p, b0, w1p, shu._cw))
print(f'shu_spectral         {us:7.1f} us  ({2 * N * 2112 * (64 * 64 + 6 * 64 * 64) / us / 1e6:6.1f} TFLOP/s)')
s1 = kk.shu_spectral(t, w0p, b0, w1p, shu._cw)
us = timeit(lambda: kk.shu_split_irfft2(s1, None, gauss, outs, False))
print(f'shu_split_irfft2 B=1 {us:7.1f} us')
