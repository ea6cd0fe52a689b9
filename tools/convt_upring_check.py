"""Merged-phase transposed convolution (csrc/conv_f16_upring.hip) against the per-phase launches of conv_f16_kernel (variant library built with
-DSHG_F16_NO_UPRING) -- bit-exact by construction -- and against a float64 torch conv_transpose2d of the same half operands; then timings.
usage: python sh-gan_amd/build.py --variant=noupring -DSHG_F16_NO_UPRING=1 && python tools/convt_upring_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import _lib
from shgan_amd import kernels_f16 as kf

dev = 'cuda:0'
CL = torch.channels_last
VAR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_noupring.so')
PROD = _lib.LIB_PATH
bad = 0
# n, i, o, h, w, padding, out_hw, in_scale
cases = [(1, 32, 8, 4, 4, 0, None, False), (2, 64, 64, 16, 16, 0, None, False), (1, 32, 72, 7, 5, 1, None, True), (3, 128, 96, 20, 33, 0, None, True),
         (2, 64, 40, 31, 32, 1, None, False), (2, 96, 64, 16, 31, 1, (32, 62), False), (1, 64, 64, 17, 40, 0, (33, 81), True),
         (2, 512, 256, 16, 16, 0, None, True), (8, 128, 64, 256, 256, 0, None, False), (8, 512, 512, 32, 32, 0, None, True), (4, 256, 128, 64, 64, 1, (128, 128), False)]
for (n, i, o, h, w, pad, out_hw, sc) in cases:
    torch.manual_seed(n * 100 + i + o + h + w)
    x = torch.randn(n, i, h, w, device=dev).half().to(memory_format=CL)
    wt = (torch.randn(i, o, 3, 3, device=dev) / (i * 9 / 4) ** 0.5).half()
    s = (torch.rand(n, i, device=dev) + 0.5) if sc else None
    y_new = kf.conv_transpose2d(x, wt, None, pad, out_hw, in_scale=s)
    torch.cuda.synchronize()
    xs = x.double() if s is None else (x * s.half().reshape(n, i, 1, 1)).double()
    ref = F.conv_transpose2d(xs, wt.double(), stride=2, padding=0)
    if pad or out_hw:
        oh, ow = out_hw if out_hw else (ref.shape[2] - 2 * pad, ref.shape[3] - 2 * pad)
        full = torch.zeros(n, o, pad + oh + 4, pad + ow + 4, dtype=torch.float64, device=dev)
        full[:, :, :ref.shape[2], :ref.shape[3]] = ref
        ref = full[:, :, pad:pad + oh, pad:pad + ow]
    err = float((y_new.double() - ref).abs().max() / ref.abs().max())
    line = f'n{n} {i:4d}->{o:4d} {h:4d}x{w:<4d} pad {pad} out {out_hw} scale {int(sc)}: vs float64 {err:.2e}'
    if os.path.exists(VAR):
        _lib.use_library(VAR)
        y_old = kf.conv_transpose2d(x, wt, None, pad, out_hw, in_scale=s)
        torch.cuda.synchronize()
        _lib.use_library(PROD)
        same = torch.equal(y_old, y_new)
        line += f' | bit-exact vs per-phase launches: {same} ({int((y_old != y_new).sum())} of {y_new.numel()} differ)'
        bad += 0 if same else 1
    bad += 0 if err < 2e-3 else 1
    print(line, flush=True)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, i, o, r) in [(8, 128, 64, 256), (8, 256, 128, 128), (8, 512, 256, 64), (8, 512, 512, 32), (16, 128, 64, 256), (16, 512, 512, 32)]:
    x = torch.randn(n, i, r, r, device=dev).half().to(memory_format=CL)
    pk = kf.pack_weight((torch.randn(i, o, 3, 3, device=dev) / (i * 9 / 4) ** 0.5).half(), transposed=True)
    fl = 2.0 * n * r * r * i * o * 9
    line = f'convT n{n} {i:4d}->{o:4d} {r:4d}^2:'
    for name, path in (('merged', PROD), ('per-phase', VAR)):
        if not os.path.exists(path):
            continue
        _lib.use_library(path)
        us = timeit(lambda: kf.conv_transpose2d(x, pk, None, 0, None))
        line += f'  {name} {us:7.1f} us {fl / us / 1e6:7.1f} TFLOP/s'
    _lib.use_library(PROD)
    print(line, flush=True)
print('FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
