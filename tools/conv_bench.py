"""Micro-benchmark of the MFMA conv kernel on the generator's layer shapes (512x512, batch 16).
Prints TFLOP/s (algorithmic flops / HIP-event time) per layer.  usage: python tools/conv_bench.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib, kernels as kk
if os.environ.get('SHG_VARIANT'):     # A/B runs: python sh-gan_amd/build.py --variant=<name> -DKNOB=1
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
if os.environ.get('SHG_ABLATE'):      # timing studies: the -DSHG_ABLATE build (python sh-gan_amd/build.py --ablate) honours SHG_*_DBG
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_ablate.so'))

N = 16
# name, I, O, H(in), mode, modulated
LAYERS = [
    ('enc512.conv0 64->64 s1', 64, 64, 512, 0, False), ('enc512.conv1 64->128 s2', 64, 128, 513, 1, False),
    ('enc256.conv0 128->128', 128, 128, 256, 0, False), ('enc256.conv1 s2', 128, 256, 257, 1, False),
    ('enc128.conv0 256', 256, 256, 128, 0, False), ('enc128.conv1 s2', 256, 512, 129, 1, False),
    ('enc64.conv0 512', 512, 512, 64, 0, False), ('enc64.conv1 s2', 512, 512, 65, 1, False),
    ('enc32.conv0', 512, 512, 32, 0, False), ('enc32.conv1 s2', 512, 512, 33, 1, False),
    ('enc16.conv0', 512, 512, 16, 0, False), ('enc16.conv1 s2', 512, 512, 17, 1, False),
    ('enc8.conv0', 512, 512, 8, 0, False), ('enc8.conv1 s2', 512, 512, 9, 1, False), ('b4.conv', 512, 512, 4, 0, True),
    ('syn8.up 4->8', 512, 512, 4, 2, True), ('syn16.up', 512, 512, 8, 2, True), ('syn32.up', 512, 512, 16, 2, True),
    ('syn64.up', 512, 512, 32, 2, True), ('syn64.conv1', 512, 512, 64, 0, True),
    ('syn128.up 512->256', 512, 256, 64, 2, True), ('syn128.conv1', 256, 256, 128, 0, True),
    ('syn256.up 256->128', 256, 128, 128, 2, True), ('syn256.conv1', 128, 128, 256, 0, True),
    ('syn512.up 128->64', 128, 64, 256, 2, True), ('syn512.conv1', 64, 64, 512, 0, True),
]
for ci in (8, 16, 32, 64, 128, 256):
    LAYERS.append((f'fit64 I={ci} O=512', ci, 512, 64, 0, False))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
if not flt:
    LAYERS = [l for l in LAYERS if not l[0].startswith('fit')]
dev = 'cuda'
tot_ms = tot_fl = 0.0
for name, ci, co, h, mode, mod in LAYERS:
    if flt and flt not in name:
        continue
    x = torch.randn(N, ci, h, h, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev)
    pw = kk.conv_weight_prep(w, demod=mod)
    s_in = torch.rand(N, ci, device=dev) + 0.5 if mod else None
    s_out = torch.rand(N, co, device=dev) + 0.5 if mod else None
    bias = torch.randn(co, device=dev)
    def run():
        if mode == 2:
            return kk.conv2d(x, pw, mode=2, in_scale=s_in, planar=True)
        return kk.conv2d(x, pw, mode=mode, pad=(1 if mode == 0 else 0), in_scale=s_in, out_scale=s_out, bias=bias, act=True)
    for _ in range(2):
        y = run()
    reps, ms = 5, 1e9
    for _trial in range(3):          # best of three (clock / cache state varies by a few per cent between trials)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = run()
        e1.record(); torch.cuda.synchronize()
        ms = min(ms, e0.elapsed_time(e1) / reps)
    pix = h * h if mode == 2 else (y.shape[-1] * y.shape[-2])
    fl = 2.0 * N * co * ci * 9 * pix
    tot_ms += ms; tot_fl += fl
    print(f'{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({fl/1e9:7.1f} GFLOP)', flush=True)
    del x, w, y
print(f'TOTAL {tot_ms:.2f} ms  {tot_fl/tot_ms/1e9:.1f} TFLOP/s')
