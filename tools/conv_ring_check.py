"""Ring kernel (csrc/conv_f16_ring.hip) against conv_f16_kernel (variant library built with -DSHG_F16_NO_RING) -- bit-exact by construction
(same MFMA sequence per accumulator) -- and against a float64 torch convolution of the same half operands.
usage: python sh-gan_amd/build.py --variant=noring -DSHG_F16_NO_RING=1 && python tools/conv_ring_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import _lib
from shgan_amd import kernels_f16 as kf

dev = 'cuda:0'
CL = torch.channels_last
VAR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_noring.so')
PROD = _lib.LIB_PATH


def run(lib_path, fn):
    if lib_path:
        _lib.use_library(lib_path)
    out = fn()
    torch.cuda.synchronize()
    return out


bad = 0
cases = [(1, 32, 64, 16, 32, True), (2, 64, 64, 64, 64, False), (1, 64, 64, 40, 72, True), (3, 128, 128, 32, 32, True), (2, 96, 200, 19, 45, False),
         (8, 64, 64, 512, 512, True), (8, 128, 128, 256, 256, False), (8, 512, 512, 64, 64, True), (2, 256, 512, 33, 31, True), (1, 32, 8, 7, 5, False)]
for (n, i, o, h, w, with_bias) in cases:
    torch.manual_seed(n * 1000 + i + o + h)
    x = torch.randn(n, i, h, w, device=dev).half().to(memory_format=CL)
    wt = (torch.randn(o, i, 3, 3, device=dev) / (i * 9) ** 0.5).half()
    b = torch.randn(o, device=dev) if with_bias else None
    y_new = run(None, lambda: kf.conv2d(x, wt, b, 1, 1))
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), 1, 1)
    err = float((y_new.double() - ref).abs().max() / ref.abs().max())
    line = f'n{n} {i:4d}->{o:4d} {h:4d}x{w:<4d} bias={int(with_bias)}: vs float64 {err:.2e}'
    if os.path.exists(VAR):
        _lib.use_library(VAR)
        y_old = kf.conv2d(x, wt, b, 1, 1)
        torch.cuda.synchronize()
        _lib.use_library(PROD)
        same = torch.equal(y_old, y_new)
        nd = int((y_old != y_new).sum())
        line += f' | bit-exact vs conv_f16_kernel: {same} ({nd} of {y_new.numel()} differ)'
        bad += 0 if same else 1
    bad += 0 if err < 2e-3 else 1
    print(line, flush=True)
# the fused layer tail (everything but in_scale / residual): against conv_f16_kernel's store pass, bit for bit
tails = [(2, 64, 64, 64, 64, dict(act=True)), (8, 64, 64, 512, 512, dict(act=True, bias=True)), (3, 128, 96, 40, 52, dict(act=True, bias=True, d=True, noise=1)),
         (2, 256, 256, 32, 36, dict(act=False, gain=0.5, d=True, noise=2)), (4, 512, 512, 16, 16, dict(act=True, bias=True, d=True, noise=2, clamp=0.7)),
         (2, 64, 64, 100, 132, dict(act=True, bias=True, noise=1)), (1, 32, 72, 21, 20, dict(act=True, bias=True, d=True)), (8, 128, 128, 256, 256, dict(act=True, bias=True))]
for (n, i, o, h, w, t) in tails:
    torch.manual_seed(n * 77 + i + o + h)
    x = torch.randn(n, i, h, w, device=dev).half().to(memory_format=CL)
    wt = (torch.randn(o, i, 3, 3, device=dev) / (i * 9) ** 0.5).half()
    kw = dict(act=t.get('act'), gain=t.get('gain', 1.0), clamp=t.get('clamp', 256.0))
    b = torch.randn(o, device=dev) if t.get('bias') else None
    if t.get('d'):
        kw['out_scale'] = torch.rand(n, o, device=dev) + 0.5
    if t.get('noise') == 1:
        kw['noise'], kw['noise_strength'] = torch.randn(h, w, device=dev), 0.3
    if t.get('noise') == 2:
        kw['noise'], kw['noise_strength'] = torch.randn(n, 1, h, w, device=dev), 0.3
    y_new = run(None, lambda: kf.conv2d(x, wt, b, 1, 1, **kw))
    line = f'tail n{n} {i:4d}->{o:4d} {h:4d}x{w:<4d} {t}:'
    if os.path.exists(VAR):
        _lib.use_library(VAR)
        y_old = kf.conv2d(x, wt, b, 1, 1, **kw)
        torch.cuda.synchronize()
        _lib.use_library(PROD)
        same = torch.equal(y_old, y_new)
        line += f' bit-exact vs conv_f16_kernel: {same} ({int((y_old != y_new).sum())} of {y_new.numel()} differ, max {float((y_old.float() - y_new.float()).abs().max()):.3e})'
        bad += 0 if same else 1
    else:
        line += ' (no noring variant built)'
    print(line, flush=True)
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
