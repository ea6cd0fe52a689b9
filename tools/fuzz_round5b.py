"""Random-shape fuzz of what the second half of round 5 added: the native op over the plugin's whole operand range (shg_upfirdn2d_strided:
dtype x layout x geometry vs the oracle's float64 upfirdn2d), the fused modulated form on halves (per-sample weights; vs a float64
evaluation of stylegan.py:136-193 on the same half inputs, and vs the non-fused form), the closed double backward of the style factors
(vs torch's float64 double backward of the tensor-op formulation).
usage: python tools/fuzz_round5b.py [cases=150] [seed=0]"""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import shgan_amd  # noqa: F401
from shgan_amd.model_zoo import stylegan as sg
from shgan_amd.model_zoo.stylegan_utils import custom_ops, upfirdn2d as ufd
from oracle import shgan_oracle as orc

cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); torch.manual_seed(seed)
DEV = 'cuda:0'
plugin = custom_ops.get_plugin('upfirdn2d_plugin')
bad = {'strided': 0, 'fused_half': 0, 'style_bwd': 0}
worst = {'strided': {}, 'fused_half': 0.0, 'style_bwd': 0.0}
TOL = {torch.float64: 1e-12, torch.float32: 3e-6, torch.float16: 2e-3}


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))


for it in range(cases):
    # ---- strided native op
    dt = rnd.choice([torch.float64, torch.float32, torch.float16])
    n, c, h, w = rnd.randint(1, 3), rnd.randint(1, 9), rnd.randint(1, 20), rnd.randint(1, 20)
    layout = rnd.choice(['nchw', 'cl', 'view', 'tview'])
    if layout == 'view':
        big = torch.randn(n, c + 2, h + 3, w + 5, dtype=torch.float64).to(dt).to(DEV)
        x = big[:, 1:c + 1, 2:h + 2, 1:w + 1]
    elif layout == 'tview':
        x = torch.randn(n, c, w, h, dtype=torch.float64).to(dt).to(DEV).transpose(2, 3)          # H and W strides exchanged
    else:
        x = torch.randn(n, c, h, w, dtype=torch.float64).to(dt).to(DEV)
        if layout == 'cl':
            x = x.contiguous(memory_format=torch.channels_last)
    fh, fw = rnd.randint(1, 6), rnd.randint(1, 6)
    f = torch.randn(fw, fh, device=DEV).t() if rnd.random() < 0.5 else torch.randn(fh, fw, device=DEV)
    upx, upy, dnx, dny = rnd.randint(1, 3), rnd.randint(1, 3), rnd.randint(1, 3), rnd.randint(1, 3)
    pad = [rnd.randint(-2, 5) for _ in range(4)]
    ow = (w * upx + pad[0] + pad[1] - fw + dnx) // dnx
    oh = (h * upy + pad[2] + pad[3] - fh + dny) // dny
    if oh >= 1 and ow >= 1 and h * upy + pad[2] + pad[3] >= fh and w * upx + pad[0] + pad[1] >= fw:
        flip, gain = rnd.random() < 0.5, float(np.float32(rnd.choice([1.0, 4.0, 0.3])))          # (the op's gain is a C float, upfirdn2d.cpp:16)
        from shgan_amd import kernels
        y = kernels.upfirdn2d_strided(x, f, upx, upy, dnx, dny, pad[0], pad[1], pad[2], pad[3], flip, gain)
        ref = orc.upfirdn2d(x.cpu().double(), f.cpu().contiguous().double(), up=[upx, upy], down=[dnx, dny], padding=pad, flip_filter=flip, gain=gain)
        ok = tuple(y.shape) == tuple(ref.shape) and y.dtype == dt
        e = rel(y, ref) if ok and float(ref.abs().max()) > 0 else (0.0 if ok else 1.0)
        worst['strided'][str(dt)] = max(worst['strided'].get(str(dt), 0.0), e)
        if not (ok and e < TOL[dt]):
            bad['strided'] += 1; print('strided', dt, layout, (n, c, h, w), (fh, fw), (upx, upy, dnx, dny), pad, e)
    # ---- fused modulated form on halves
    n, i, o = rnd.randint(1, 3), 8 * rnd.randint(1, 8), rnd.choice([3, 8, 24, 40, 64])
    k = rnd.choice([1, 3]) if o != 3 else 1
    up = rnd.choice([1, 2]) if (k == 3 and o % 8 == 0) else 1
    demod = o != 3
    if not demod:
        up = 1
    hh, ww = rnd.randint(4, 24), rnd.randint(4, 24)
    xh = (torch.randn(n, i, hh, ww) * 2).half().to(DEV).to(memory_format=torch.channels_last)
    wt = torch.randn(o, i, k, k, device=DEV)
    st = torch.randn(n, i, device=DEV) + 1.0
    noise = (torch.randn(hh * up, ww * up, device=DEV) * 0.1) if (demod and rnd.random() < 0.6) else None
    f4 = ufd.setup_filter([1, 3, 3, 1]).to(DEV)
    kw = dict(weight=wt, styles=st, noise=noise, up=up, padding=k // 2, resample_filter=f4 if up > 1 else None, demodulate=demod, flip_weight=(up == 1))
    with torch.no_grad():
        y = sg.modulated_conv2d(x=xh, fused_modconv=True, **kw)
        y_nf = sg.modulated_conv2d(x=xh, fused_modconv=False, **kw)
        ref = orc.modulated_conv2d(xh.cpu().double().contiguous(), wt.cpu().double(), st.cpu().double(), noise=None if noise is None else noise.cpu().double(),
                                   up=up, padding=k // 2, resample_filter=f4.cpu().double() if up > 1 else None, demodulate=demod, flip_weight=(up == 1))
    e, e_nf = rel(y, ref), rel(y_nf, ref)
    worst['fused_half'] = max(worst['fused_half'], e)
    if not (y.dtype == torch.float16 and e < 4e-3):
        bad['fused_half'] += 1; print('fused_half', (n, i, o, k, up, hh, ww), e, e_nf)
    # ---- closed double backward of the style factors
    n, i, o, half = rnd.randint(1, 8), rnd.choice([8, 24, 64, 200, 512]), rnd.choice([3, 16, 64, 300, 512]), rnd.random() < 0.5
    g = torch.Generator().manual_seed(seed * 100003 + it)
    mk = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64).to(DEV)          # noqa: E731
    s64, w64 = mk(n, i) + 1.0, torch.rand(o, i, generator=g, dtype=torch.float64).to(DEV) * 0.01
    a64, b64, A64, B64 = mk(n, i), mk(n, o), mk(n, i), mk(o, i)

    def ref_fn(s, w):
        if half:
            s = s / s.norm(float('inf'), dim=1, keepdim=True)
        s = s * s.square().mean().rsqrt()
        return s, (s.square().matmul(w.t()) + 1e-8).rsqrt()

    def run(fn, dt):
        s, w, a, b = (t.to(dt).clone().requires_grad_(True) for t in (s64, w64, a64, b64))
        with torch.enable_grad():
            sn, d = fn(s, w)
            gs, gw = torch.autograd.grad([sn, d], [s, w], [a, b], create_graph=True)
            return torch.autograd.grad((gs * A64.to(dt)).sum() + (gw * B64.to(dt)).sum(), [s, w, a, b])
    if n * i <= 8192:
        want, got = run(ref_fn, torch.float64), run(lambda s, w: sg._StyleFactorsFn.apply(s, w, half), torch.float32)
        e = max(rel(x, y) for x, y in zip(got, want))
        worst['style_bwd'] = max(worst['style_bwd'], e)
        if not e < 2e-4:
            bad['style_bwd'] += 1; print('style_bwd', (n, i, o, half), e)
torch.cuda.synchronize()
print(f'{cases} cases: mismatches {bad}; worst relative errors {worst}')
sys.exit(1 if any(bad.values()) else 0)
