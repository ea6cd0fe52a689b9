"""Micro-benchmark of the encoder's FIR + stride-2 3x3 convolutions (512x512, batch 16): polyphase-Winograd route vs the
direct route.  Prints direct-form TFLOP/s (algorithmic flops / HIP-event time, FIR included in the time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import _lib, kernels as kk
if os.environ.get('SHG_VARIANT'):     # A/B runs: python sh-gan_amd/build.py --variant=<name> -DKNOB=1
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd.model_zoo.stylegan_utils import upfirdn2d as ufd

N = 16
LAYERS = [('enc512.conv1 64->128', 64, 128, 512), ('enc256.conv1 128->256', 128, 256, 256), ('enc128.conv1 256->512', 256, 512, 128),
          ('enc64.conv1 512', 512, 512, 64), ('enc32.conv1 512', 512, 512, 32), ('enc16.conv1 512', 512, 512, 16)]
f = ufd.setup_filter([1, 3, 3, 1]).cuda()
for name, ci, co, h in LAYERS:
    x = torch.randn(N, ci, h, h, device='cuda')
    w = torch.randn(co, ci, 3, 3, device='cuda')
    pw = kk.conv_weight_prep(w, gain=0.01)
    bias = torch.randn(co, device='cuda')
    def poly():
        return kk.fir_conv_down2(x, f, pw, bias=bias, act=True)
    def direct():
        return kk.conv2d(kk.upfirdn2d(x, f, padx0=2, padx1=2, pady0=2, pady1=2), pw, mode=1, pad=0, bias=bias, act=True)
    out = []
    for fn in ([poly] if kk.down_poly_supported(x, pw, force=True) else []) + [direct]:
        for _ in range(2):
            fn()
        ms = 1e9
        for _t in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = min(ms, e0.elapsed_time(e1) / 5)
        out.append((fn.__name__, ms))
    fl = 2.0 * N * co * ci * 9 * (h // 2) ** 2
    print(f'{name:26s} ' + '  '.join(f'{k}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF' for k, ms in out), flush=True)
