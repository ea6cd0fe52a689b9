"""Random-shape fuzz of the round-6 kernels against torch CPU (float64 accumulate): the unmasked weight gradient and its dispatch boundary
(channels in / not in 64-blocks, rows in / not in whole chunks, 2 / 4 rows per chunk, 1x1 layers as one row of H W pixels, FIR-padded widths and
widths one off) and the 1x1 GEMM form (whole / ragged tiles, bias, activation, gain, skip tensor, unaligned views).
  python tools/fuzz_round6.py [cases]   -> worst relative error per family; exits non-zero above 3e-5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import shgan_amd  # noqa: F401
from shgan_amd import kernels as kk

DEV = 'cuda:0'
rs = np.random.RandomState(606)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


worst = {'wgrad s2': 0.0, 'wgrad 1x1': 0.0, 'conv 1x1': 0.0}
for c in range(cases):
    # ---- stride-2 3x3 weight gradient
    n = int(rs.randint(1, 5)); ci = int(rs.choice([64, 128, 192, 48])); co = int(rs.choice([64, 128, 40]))
    ow = int(rs.choice([4, 8, 16, 32, 64, 96, 24])); oh = int(rs.choice([4, 8, 10, 16, 33]))
    w_ = 2 * ow + int(rs.choice([1, 1, 1, 2])); h_ = 2 * oh + int(rs.choice([1, 2]))
    x = torch.from_numpy(rs.standard_normal((n, ci, h_, w_)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((n, co, (h_ - 3) // 2 + 1, (w_ - 3) // 2 + 1)).astype(np.float32))
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), g.double(), stride=2, padding=0)
    got = kk.conv2d_wgrad(x.to(DEV), g.to(DEV), 3, 3, 2, 0)
    e = rel(got, ref); worst['wgrad s2'] = max(worst['wgrad s2'], e)
    assert e < 3e-5, ('wgrad s2', n, ci, co, h_, w_, e)
    # ---- 1x1 weight gradient
    hh, ww = [(8, 8), (16, 32), (4, 16), (24, 24), (64, 64), (5, 13)][rs.randint(6)]
    ci = int(rs.choice([64, 128, 72])); co = int(rs.choice([64, 192, 24]))
    x = torch.from_numpy(rs.standard_normal((n, ci, hh, ww)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((n, co, hh, ww)).astype(np.float32))
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 1, 1), g.double())
    e = rel(kk.conv2d_wgrad(x.to(DEV), g.to(DEV), 1, 1, 1, 0), ref); worst['wgrad 1x1'] = max(worst['wgrad 1x1'], e)
    assert e < 3e-5, ('wgrad 1x1', n, ci, co, hh, ww, e)
    # ---- 1x1 forward (GEMM form where whole tiles fill the chip, tap-list kernel otherwise)
    n = int(rs.choice([1, 4, 8, 16])); ci = int(rs.choice([16, 32, 64, 80, 128])); co = int(rs.choice([64, 128, 192, 256, 100]))
    hh, ww = [(16, 16), (32, 32), (64, 32), (128, 128), (24, 24), (64, 64)][rs.randint(6)]
    x = torch.from_numpy(rs.standard_normal((n, ci, hh, ww)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((co, ci, 1, 1)) / np.sqrt(ci)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32)) if rs.rand() < 0.5 else None
    res = torch.from_numpy(rs.standard_normal((n, co, hh, ww)).astype(np.float32)) if rs.rand() < 0.5 else None
    act = bool(rs.rand() < 0.5); gain = float(rs.choice([1.0, 0.5, np.sqrt(0.5)]))
    z = F.conv2d(x.double(), wt.double(), None if bias is None else bias.double())
    if act:
        z = torch.where(z < 0, z * 0.2, z) * (np.sqrt(2.0) * gain)
        z = z.clamp(-256.0 * gain, 256.0 * gain)
    else:
        z = z * gain
    if res is not None:
        z = z + res.double()
    xd = x.to(DEV)
    if rs.rand() < 0.2:                                  # a view that is not 16-byte aligned
        xd = torch.zeros(x.numel() + 1, device=DEV)[1:].view_as(x).copy_(xd)
    got = kk.conv2d(xd, kk.conv_weight_prep(wt.to(DEV)), mode=0, pad=0, bias=None if bias is None else bias.to(DEV), act=act, gain=gain,
                    residual=None if res is None else res.to(DEV))
    e = rel(got, z); worst['conv 1x1'] = max(worst['conv 1x1'], e)
    assert e < 3e-5, ('conv 1x1', n, ci, co, hh, ww, act, gain, bias is not None, res is not None, e)
print(f'{cases} cases per family, worst relative error:', {k: f'{v:.2e}' for k, v in worst.items()})
