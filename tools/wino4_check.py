"""F(4x4,3x3) Winograd route: agreement with the direct kernel / torch fp64 and timing against F(2x2,3x3).
usage: python tools/wino4_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import _lib, kernels as kk
if os.environ.get('SHG_VARIANT'):
    _lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))

dev = 'cuda'
torch.manual_seed(0)


def run(x, pw, route, **kw):
    kk.WINO, kk.WINO4 = route != 'direct', route == 'wino4'
    try:
        return kk.conv2d(x, pw, mode=0, pad=1, **kw)
    finally:
        kk.WINO, kk.WINO4 = True, True


if os.environ.get('SHG_VARIANT'):
    CASES = []
else:
    CASES = [(2, 8, 64, 32, 32), (1, 13, 70, 36, 40), (3, 64, 64, 64, 64), (2, 100, 130, 48, 96), (1, 512, 512, 32, 32), (2, 24, 3, 34, 52)]
# ---- correctness: odd sizes, ragged channels, every fused operand
for (n, ci, co, h, w) in CASES:
    x = torch.randn(n, ci, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    pw = kk.conv_weight_prep(wt)
    s_in = torch.rand(n, ci, device=dev) + 0.5
    s_out = torch.rand(n, co, device=dev) + 0.5
    bias = torch.randn(co, device=dev)
    noise = torch.randn(n, 1, h, w, device=dev)
    res = torch.randn(n, co, h, w, device=dev)
    kw = dict(in_scale=s_in, out_scale=s_out, bias=bias, noise=noise, noise_strength=0.3, act=True, residual=res)
    y4 = run(x, pw, 'wino4', **kw)
    y2 = run(x, pw, 'wino', **kw)
    yd = run(x, pw, 'direct', **kw)
    ref = F.conv2d((x * s_in[:, :, None, None]).double(), wt.double(), padding=1) * s_out[:, :, None, None].double()
    ref = (F.leaky_relu(ref + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5).clamp(-256, 256) + res.double()
    sc = ref.abs().max().item()
    print(f'N{n} I{ci} O{co} {h}x{w}: wino4 {(y4.double()-ref).abs().max().item()/sc:.2e}  wino2 {(y2.double()-ref).abs().max().item()/sc:.2e}'
          f'  direct {(yd.double()-ref).abs().max().item()/sc:.2e}', flush=True)

# ---- timing on the generator's stride-1 layers (512x512, batch 16)
N = 16
LAYERS = [('enc512.conv0', 64, 64, 512, False), ('enc256.conv0', 128, 128, 256, False), ('enc128.conv0', 256, 256, 128, False),
                             ('enc64.conv0', 512, 512, 64, False), ('enc32.conv0', 512, 512, 32, False), ('syn64.conv1', 512, 512, 64, True),
                             ('syn128.conv1', 256, 256, 128, True), ('syn256.conv1', 128, 128, 256, True), ('syn512.conv1', 64, 64, 512, True)]
if os.environ.get('SHG_VARIANT'):
    LAYERS = [LAYERS[0], LAYERS[1], LAYERS[3]]
for name, ci, co, h, mod in LAYERS:
    x = torch.randn(N, ci, h, h, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev)
    pw = kk.conv_weight_prep(wt, demod=mod)
    s_in = torch.rand(N, ci, device=dev) + 0.5 if mod else None
    s_out = torch.rand(N, co, device=dev) + 0.5 if mod else None
    bias = torch.randn(co, device=dev)
    out = []
    for route in (('wino4', 'wino4') if os.environ.get('SHG_VARIANT') else ('wino', 'wino4')):
        for _ in range(2):
            run(x, pw, route, in_scale=s_in, out_scale=s_out, bias=bias, act=True)
        ms = 1e9
        for _t in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(x, pw, route, in_scale=s_in, out_scale=s_out, bias=bias, act=True)
            e1.record(); torch.cuda.synchronize()
            ms = min(ms, e0.elapsed_time(e1) / 5)
        out.append(ms)
    fl = 2.0 * N * co * ci * 9 * h * h
    print(f'{name:14s} F(2x2) {out[0]*1e3:8.1f} us {fl/out[0]/1e9:6.1f} TF | F(4x4) {out[1]*1e3:8.1f} us {fl/out[1]/1e9:6.1f} TF (direct-form)  x{out[0]/out[1]:.2f}', flush=True)

