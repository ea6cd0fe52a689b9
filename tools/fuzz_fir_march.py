"""Randomised sweep of the row-marching FIR kernels (csrc/fir_march.h) against the CPU oracle: pad-2 pre-filter (plain and polyphase
planes with random pitches), x2 down / up resampling, FIR-from-phase-planes with random subsets of the fused tail; random plane
counts (groups of planes per wave only partly filled), row counts (partial last segments), separable taps, flips, gains.
usage: python tools/fuzz_fir_march.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import shgan_amd
from shgan_amd import kernels as kk
from oracle import shgan_oracle as orc

DEV = 'cuda'
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = kk._lib.get_lib()


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def taps():
    if rs.randint(2):
        return orc.setup_filter([1, 3, 3, 1])
    while True:          # exactly representable taps: the outer product stays rank 1 in fp32
        a, b = rs.randint(-8, 9, 4) / 8.0, rs.randint(-8, 9, 4) / 4.0
        if abs(a.sum() * b.sum()) > 0.1:
            return torch.from_numpy(np.outer(a, b).astype(np.float32))


worst, ran = 0.0, {}
for case in range(n_cases):
    kind = rs.choice(['pad2', 'planes', 'dn2', 'up2', 'upfir'])
    n, c = int(rs.choice([1, 2, 3, 7])), int(rs.choice([1, 2, 3, 5, 13, 37]))
    f = taps()
    flip, gain = bool(rs.randint(2)), float(rs.choice([1.0, 4.0, 0.37]))
    if kind in ('pad2', 'planes', 'dn2'):
        w = int(rs.choice([8, 16, 32, 64, 128, 256, 512]))
        h = int(rs.choice([2, 4, 6, 10, 16, 34, 64, 130])) if kind != 'pad2' else int(rs.choice([2, 3, 5, 9, 16, 33, 64, 131]))
        if w >= 256:
            n, c = min(n, 2), min(c, 5)
        x = torch.from_numpy(rs.standard_normal((n, c, h, w)).astype(np.float32))
        if kind == 'dn2':
            ref = orc.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1], flip_filter=flip, gain=gain).numpy()
            assert lib.shg_fir_resample2_sep_supported(h, w, 1)
            y = kk.upfirdn2d(x.to(DEV), f.to(DEV), downx=2, downy=2, padx0=1, padx1=1, pady0=1, pady1=1, flip=flip, gain=gain).cpu().numpy()
        else:
            ref = orc.upfirdn2d(x, f, padding=[2, 2, 2, 2], flip_filter=flip, gain=gain).numpy()
            if kind == 'pad2':
                assert lib.shg_fir_pad2_sep_supported(h, w, 0)
                y = kk.upfirdn2d(x.to(DEV), f.to(DEV), padx0=2, padx1=2, pady0=2, pady1=2, flip=flip, gain=gain).cpu().numpy()
            else:
                pp = (w // 2 + 1 + 3) // 4 * 4 + 4 * int(rs.choice([0, 0, 1, 3, 7]))
                assert lib.shg_fir_pad2_sep_supported(h, w, pp)
                xp = torch.full((4, n, c, h // 2 + 1, pp), float('nan'), device=DEV)
                xd, fd = x.to(DEV), f.to(DEV)           # (kept alive across the raw C call)
                kk.check(lib.shg_fir_pad2_sep_f32(kk._ptr(xd), kk.sep_taps(fd), kk._ptr(xp), n, c, h, w, pp, int(flip), gain, None), 'pad2')
                torch.cuda.synchronize()
                full = np.zeros((n, c, 2 * (h // 2 + 1), 2 * pp), np.float32)
                full[:, :, :h + 1, :w + 1] = ref
                got = xp.cpu().numpy()
                assert not np.isnan(got).any(), (case, 'unwritten plane entries')
                y = np.zeros_like(full)
                for a in range(2):
                    for b in range(2):
                        y[:, :, a::2, b::2] = got[a * 2 + b]
                ref = full
                assert (y[ref == 0] == 0).all() or np.abs(y[ref == 0]).max() < 1e-6
    elif kind == 'up2':
        w = int(rs.choice([4, 8, 16, 32, 64, 128, 256])); h = int(rs.choice([1, 2, 3, 8, 17, 64, 100]))
        if w >= 128:
            n, c = min(n, 2), min(c, 5)
        x = torch.from_numpy(rs.standard_normal((n, c, h, w)).astype(np.float32))
        ref = orc.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], flip_filter=flip, gain=gain).numpy()
        assert lib.shg_fir_resample2_sep_supported(h, w, 2)
        y = kk.upfirdn2d(x.to(DEV), f.to(DEV), upx=2, upy=2, padx0=2, padx1=1, pady0=2, pady1=1, flip=flip, gain=gain).cpu().numpy()
    else:
        w = int(rs.choice([4, 8, 16, 32, 64, 128, 256])); h = int(rs.choice([1, 2, 3, 8, 17, 64, 100]))
        if w >= 128:
            n, c = min(n, 2), min(c, 5)
        mid = torch.from_numpy(rs.standard_normal((4, n, c, h + 1, w + 1)).astype(np.float32))
        full = torch.zeros(n, c, 2 * h + 1, 2 * w + 1)
        for a in range(2):
            for b in range(2):
                full[:, :, a::2, b::2] = mid[a * 2 + b][:, :, :h + 1 - a, :w + 1 - b]
        ref = orc.upfirdn2d(full, f, padding=[1, 1, 1, 1], flip_filter=flip, gain=4.0)
        kw = {}
        if rs.randint(2):
            kw['scale'] = torch.from_numpy(rs.rand(n * c).astype(np.float32) + 0.5); ref = ref * kw['scale'].view(n, c, 1, 1)
        if rs.randint(2):
            per = bool(rs.randint(2)) and n > 1
            kw['noise'] = torch.from_numpy(rs.standard_normal((n if per else 1, 1, 2 * h, 2 * w)).astype(np.float32)); ref = ref + kw['noise'] * 0.3
        if rs.randint(2):
            kw['bias'] = torch.from_numpy(rs.standard_normal(c).astype(np.float32)); ref = ref + kw['bias'].view(1, c, 1, 1)
        act = bool(rs.randint(2))
        if act:
            ref = orc.lrelu_agc(ref, gain=0.8)
        if rs.randint(2):
            kw['residual'] = torch.from_numpy(rs.standard_normal((n, c, 2 * h, 2 * w)).astype(np.float32)); ref = ref + kw['residual']
        ref = ref.numpy()
        assert lib.shg_upfir_planar_sep_supported(h, w)
        y = kk.upfir_planar(mid.to(DEV), f.to(DEV), noise_strength=0.3, act=act, gain=0.8, flip=flip,
                            **{k: v.to(DEV) for k, v in kw.items()}).cpu().numpy()
    e = rel(y, ref)
    worst = max(worst, e)
    ran[kind] = ran.get(kind, 0) + 1
    assert y.shape == ref.shape and e < 2e-5, (case, kind, n, c, h, w, flip, gain, e)
print(f'{n_cases} cases ok {ran}, worst rel err {worst:.2e}')
