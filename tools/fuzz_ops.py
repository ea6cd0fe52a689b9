"""Randomised shape sweep of the convolution / FIR entry points against torch CPU (fp32), for hunting edge-case bugs
beyond the fixed parity cases of tests/test_gpu_ops.py.  usage: python tools/fuzz_ops.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import kernels as kk
from oracle import shgan_oracle as orc

DEV = 'cuda'
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


worst = 0.0
for case in range(n_cases):
    kind = rs.choice(['same', 'down', 'up', 'fir'])
    n = int(rs.choice([1, 2, 3, 5]))
    ci = int(rs.choice([3, 8, 9, 16, 24, 40, 64, 72, 130]))
    co = int(rs.choice([3, 16, 33, 64, 70, 128, 192]))
    h = int(rs.choice([4, 7, 8, 16, 31, 32, 33, 40, 64, 68]))
    w = int(rs.choice([4, 8, 12, 16, 31, 32, 36, 44, 64, 65]))
    if kind == 'same' and rs.rand() < 0.4:      # wide images: the 8 x 64 and 4 x 128 tile shapes of the F(4x4) kernel
        w, h = int(rs.choice([128, 132, 200, 256, 260])), int(rs.choice([8, 12, 20, 33]))
        ci, co, n = int(rs.choice([8, 16, 24])), int(rs.choice([16, 64, 70])), int(rs.choice([1, 2]))
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    if kind == 'fir':
        f = torch.from_numpy(rs.rand(4, 4).astype(np.float32))
        pad = [int(v) for v in rs.randint(0, 4, 4)]
        up, down = int(rs.choice([1, 2])), int(rs.choice([1, 2]))
        flip, gain = bool(rs.randint(2)), float(rs.choice([1.0, 4.0, 0.37]))
        ref = orc.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        y = kk.upfirdn2d(x.to(DEV), f.to(DEV), upx=up, upy=up, downx=down, downy=down, padx0=pad[0], padx1=pad[1],
                         pady0=pad[2], pady1=pad[3], flip=flip, gain=gain)
        desc = f'fir n{n} c{ci} {h}x{w} up{up} down{down} pad{pad} flip{flip}'
    else:
        wt = torch.from_numpy(rs.standard_normal((co, ci, 3, 3)).astype(np.float32))
        mod = bool(rs.randint(2))
        s_in = torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5) if mod else None
        s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5) if mod else None
        bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
        flip = bool(rs.randint(2)) and kind != 'up'
        wref = wt.flip([2, 3]) if flip else wt
        xin = x * s_in[:, :, None, None] if mod else x
        if kind == 'same':
            ref = F.conv2d(xin, wref * 0.1, padding=1); mode, pad = 0, 1
        elif kind == 'down':
            if h < 3 or w < 3:
                continue
            ref = F.conv2d(xin, wref * 0.1, stride=2); mode, pad = 1, 0
        else:
            ref = F.conv_transpose2d(xin, (wt * 0.1).transpose(0, 1), stride=2); mode, pad = 2, 0
        if mod:
            ref = ref * s_out[:, :, None, None]
        ref = orc.lrelu_agc(ref + bias.view(1, -1, 1, 1), gain=0.7)
        pw = kk.conv_weight_prep(wt.to(DEV), gain=0.1, flip=flip)
        y = kk.conv2d(x.to(DEV), pw, mode=mode, pad=pad, in_scale=None if s_in is None else s_in.to(DEV),
                      out_scale=None if s_out is None else s_out.to(DEV), bias=bias.to(DEV), act=True, gain=0.7)
        desc = f'{kind} n{n} {ci}->{co} {h}x{w} mod{mod} flip{flip} wino{pw.wu is not None}'
    assert tuple(y.shape) == tuple(ref.shape), (desc, y.shape, ref.shape)
    e = rel(y.cpu().numpy(), ref.numpy())
    worst = max(worst, e)
    flag = '' if e < 2e-5 else '   <<<<<< FAIL'
    print(f'{case:3d} {desc:60s} rel {e:.2e}{flag}', flush=True)
    if e >= 2e-5:
        sys.exit(1)
print(f'all {n_cases} cases ok, worst rel err {worst:.2e}')
