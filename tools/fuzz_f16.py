"""Randomised shape sweep of the fp16 entry points (csrc/conv_f16.hip) against float64 torch CPU on the same half-rounded operands:
convolution (stride 1 / 2, 1x1 / 3x3, fused tail), transposed convolution (crop / zero-extension), weight gradient, FIR, modulation tail
(forward / backward).  usage: python tools/fuzz_f16.py [n_cases] [seed]   -> prints the worst error per kind; exit code 1 above 3e-3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import kernels_f16 as kf

DEV, CL = 'cuda', torch.channels_last
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def H(*shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).half()


def d(t):
    return t.to(DEV).to(memory_format=CL) if t.ndim == 4 else t.to(DEV)


worst = {}
for case in range(n_cases):
    kind = rs.choice(['conv', 'conv_fused', 'convT', 'wgrad', 'fir', 'tail'])
    n = int(rs.choice([1, 2, 3]))
    ci = int(rs.choice([4, 8, 16, 24, 32, 40, 64, 96, 136]))
    co = int(rs.choice([3, 8, 16, 24, 32, 40, 64, 72, 136]))
    h = int(rs.choice([4, 5, 8, 9, 16, 17, 31, 32, 33, 40]))
    w = int(rs.choice([4, 7, 8, 16, 17, 31, 33, 36, 48, 65]))
    k = int(rs.choice([1, 3])); s = int(rs.choice([1, 2]))
    if kind in ('conv', 'conv_fused'):
        pad = int(rs.choice([0, 1])) if k == 3 else 0
        if (h + 2 * pad - k) // s + 1 < 1 or (w + 2 * pad - k) // s + 1 < 1 or (k == 1 and s == 2):
            continue
        x, wt, b = H(n, ci, h, w), H(co, ci, k, k, scale=1 / np.sqrt(ci * k * k)), torch.from_numpy(rs.standard_normal(co).astype(np.float32))
        ref = F.conv2d(x.double(), wt.double(), None, stride=s, padding=pad)
        if kind == 'conv':
            ref = ref + b.double().view(1, -1, 1, 1)
            y = kf.conv2d(d(x), wt.to(DEV), b.to(DEV), s, pad)
        else:
            si = torch.from_numpy((rs.rand(n, ci) + 0.5).astype(np.float32)); so = torch.from_numpy((rs.rand(n, co) + 0.5).astype(np.float32))
            nz = torch.from_numpy(rs.standard_normal((n, 1, ref.shape[2], ref.shape[3])).astype(np.float32))
            res = H(*ref.shape)
            xs = (x.float() * si.half().float().view(n, ci, 1, 1)).half()
            ref = F.conv2d(xs.double(), wt.double(), None, stride=s, padding=pad).half().double()
            ref = (F.leaky_relu(ref * so.double().view(n, co, 1, 1) + nz.double() * 0.5 + b.double().view(1, -1, 1, 1), 0.2) * np.sqrt(2)).clamp(-256, 256)
            ref = ref.half().double() + res.double()
            y = kf.conv2d(d(x), wt.to(DEV), b.to(DEV), s, pad, in_scale=si.to(DEV), out_scale=so.to(DEV), noise=nz.to(DEV), noise_strength=0.5, act=True,
                          residual=d(res))
        e = rel(y, ref)
    elif kind == 'convT':
        pad = int(rs.choice([0, 1]))
        x, wt = H(n, ci, h, w), H(ci, co, 3, 3, scale=1 / np.sqrt(ci * 9))
        full = F.conv_transpose2d(x.double(), wt.double(), stride=2)
        oh, ow = int(rs.choice([2 * h + 1 - 2 * pad, 2 * h + 2 - pad, 2 * h - 1])), int(rs.choice([2 * w + 1 - 2 * pad, 2 * w + 2 - pad, 2 * w - 1]))
        ref = torch.zeros(n, co, oh, ow, dtype=torch.float64)
        hh, ww = min(oh, full.shape[2] - pad), min(ow, full.shape[3] - pad)
        ref[:, :, :hh, :ww] = full[:, :, pad:pad + hh, pad:pad + ww]
        e = rel(kf.conv_transpose2d(d(x), wt.to(DEV), None, pad, (oh, ow)), ref)
    elif kind == 'wgrad':
        pad = int(rs.choice([0, 1])) if k == 3 else 0
        if (k == 1 and s == 2) or (h + 2 * pad - k) // s + 1 < 1 or (w + 2 * pad - k) // s + 1 < 1:
            continue
        x = H(n, ci, h, w)
        g = H(n, co, (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1)
        wr = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
        with torch.enable_grad():
            F.conv2d(x.double(), wr, stride=s, padding=pad).backward(g.double())
        e = rel(kf.conv2d_wgrad(d(x), d(g), k, s, pad), wr.grad)
    elif kind == 'fir':
        c8 = int(rs.choice([8, 16, 24, 64]))
        x = H(n, c8, h, w)
        f = torch.from_numpy(rs.rand(int(rs.choice([1, 3, 4])), int(rs.choice([2, 4]))).astype(np.float32))
        pad = [int(v) for v in rs.randint(0, 4, 4)]
        up, down = int(rs.choice([1, 1, 2])), int(rs.choice([1, 1, 2]))
        flip, gain = bool(rs.randint(2)), float(rs.choice([1.0, 4.0, 0.37]))
        from oracle import shgan_oracle as orc
        try:
            ref = orc.upfirdn2d(x.float(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        except Exception:
            continue
        if ref.shape[2] < 1 or ref.shape[3] < 1:
            continue
        e = rel(kf.upfirdn2d(d(x), f.to(DEV), up, up, down, down, pad[0], pad[1], pad[2], pad[3], flip, gain), ref)
    else:
        c8 = int(rs.choice([8, 16, 32, 64, 128, 512]))
        hw = (int(rs.choice([3, 8, 12])), int(rs.choice([5, 16, 20])))
        t, dd, b = H(n, c8, *hw, scale=30), torch.from_numpy((rs.rand(n, c8) + 0.5).astype(np.float32)), torch.from_numpy(rs.standard_normal(c8).astype(np.float32))
        nz = torch.from_numpy(rs.standard_normal((n, 1) + hw).astype(np.float32))
        gy = H(n, c8, *hw)
        with torch.enable_grad():
            tr, dr = t.double().requires_grad_(True), dd.double().requires_grad_(True)
            yr = (F.leaky_relu(tr * dr.view(n, c8, 1, 1) + nz.double() + b.double().view(1, -1, 1, 1), 0.2) * np.sqrt(2)).clamp(-256, 256)
            yr.backward(gy.double())
        y = kf.modtail(d(t), dd.to(DEV), nz.to(DEV), b.to(DEV), act=True)
        gt, s1, s0, gn = kf.modtail_backward(d(gy), y, d(t), dd.to(DEV), want_sums=True, want_noise=True, act=True)
        bad = (gt.float().cpu() - tr.grad.float()).abs() > 3e-3 * tr.grad.abs().max()       # (clamp sliver, tests/test_gpu_fp16.py)
        e = max(rel(y, yr), rel(s1, dr.grad), 0.0 if bad.float().mean() < 3e-3 else 1.0)
    worst[kind] = max(worst.get(kind, 0.0), e)
    if e > 3e-3:
        print(f'case {case} {kind}: n{n} ci{ci} co{co} {h}x{w} k{k} s{s}  rel err {e:.3e}')
print('worst relative error per kind:', {k_: float('%.2e' % v) for k_, v in worst.items()})
sys.exit(1 if max(worst.values()) > 3e-3 else 0)
