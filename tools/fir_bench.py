"""Micro-benchmark of the HBM-bound kernels on the generator's shapes (512x512, batch 16): GB/s = algorithmic
bytes (every input read once, output written once) / HIP-event time.  usage: python tools/fir_bench.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib, kernels as kk
if os.environ.get('SHG_VARIANT'):     # A/B runs: python sh-gan_amd/build.py --variant=<name> -DKNOB=1
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
if os.environ.get('SHG_ABLATE'):
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_ablate.so'))

N = 16
dev = 'cuda'
flt = sys.argv[1] if len(sys.argv) > 1 else ''
f4 = torch.tensor([1., 3., 3., 1.], device=dev)
f4 = torch.outer(f4, f4); f4 = f4 / f4.sum()
CH = {512: 64, 256: 128, 128: 256, 64: 512, 32: 512}


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = 0.0
cases = []
for r in (512, 256, 128, 64):
    c = CH[r]
    h = r // 2
    def mk_up(r=r, c=c, h=h):
        mid = torch.randn(4, N, c, h + 1, h + 1, device=dev)
        res = torch.randn(N, c, r, r, device=dev)
        nz = torch.randn(N, 1, r, r, device=dev)
        sc = torch.rand(N, c, device=dev) + 0.5
        b = torch.randn(c, device=dev)
        by = (mid.numel() + 2 * res.numel() + nz.numel()) * 4
        return (lambda: kk.upfir_planar(mid, f4, scale=sc, bias=b, noise=nz, noise_strength=0.1, residual=res, act=True)), by
    cases.append((f'fir_up_planar {c}ch ->{r}', mk_up))
    def mk_same(r=r, c=c):
        x = torch.randn(N, c, r, r, device=dev)
        by = (x.numel() + N * c * (r + 1) * (r + 1)) * 4
        return (lambda: kk.upfirdn2d(x, f4, padx0=2, padx1=2, pady0=2, pady1=2)), by
    cases.append((f'fir_same {c}ch {r}->{r+1}', mk_same))
    def mk_rgb(r=r, c=c):
        x = torch.randn(N, c, r, r, device=dev)
        w = torch.randn(3, c, device=dev)
        st = torch.randn(N, c, device=dev)
        b = torch.randn(3, device=dev)
        base = torch.randn(N, 3, r // 2, r // 2, device=dev)
        by = (x.numel() + N * 3 * r * r + base.numel()) * 4
        return (lambda: kk.torgb(x, w, styles=st, bias=b, base_up=base, f=f4)), by
    cases.append((f'torgb {c}ch {r}', mk_rgb))
def mk_from():
    x = torch.randn(N, 4, 512, 512, device=dev)
    w = torch.randn(64, 4, device=dev)
    b = torch.randn(64, device=dev)
    by = (x.numel() + N * 64 * 512 * 512) * 4
    return (lambda: kk.conv1x1_thin_in(x, w, bias=b, wgain=0.5, act=True)), by
cases.append(('fromrgb 4->64 512', mk_from))
for name, mk in cases:
    if flt and flt not in name:
        continue
    fn, by = mk()
    ms = timeit(fn)
    tot += ms
    print(f'{name:30s} {ms*1e3:9.1f} us  {by/ms/1e9*1e3/1e3:8.2f} TB/s  ({by/1e6:8.1f} MB)', flush=True)
    del fn
    torch.cuda.empty_cache()
print(f'TOTAL {tot:.2f} ms')
