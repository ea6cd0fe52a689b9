"""Random-shape fuzz of the round-5 kernels: Winograd-domain weight gradient (vs float64 torch and the direct kernel), the persistent fp16 ring
kernel with its fused tail and the merged-phase transposed kernel (bit-equality with the gather kernel through shg_conv2d_f16_set_routes;
plain cases also vs float64 torch), the modulation tail's second product (vs tensor operators).
usage: python tools/fuzz_round5.py [cases=150] [seed=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import shgan_amd
from shgan_amd import kernels, kernels_f16 as kf, _lib
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); torch.manual_seed(seed)
lib = _lib.get_lib()
CL = torch.channels_last
bad = 0


def routes(fn):
    old = lib.shg_conv2d_f16_set_routes(7)
    try:
        a = fn(); lib.shg_conv2d_f16_set_routes(0); b = fn()
    finally:
        lib.shg_conv2d_f16_set_routes(old)
    torch.cuda.synchronize()
    return a, b


for it in range(cases):
    # ---- Winograd-domain weight gradient
    n, ci, co = rnd.randint(1, 6), rnd.choice([1, 3, 8, 20, 32, 33, 64, 70, 96, 130]), rnd.choice([2, 8, 16, 31, 64, 65, 100, 128])
    h, w = rnd.randint(4, 70), 4 * rnd.randint(1, 18)
    x, g = torch.randn(n, ci, h, w, device='cuda'), torch.randn(n, co, h, w, device='cuda')
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), g.double(), stride=1, padding=1)
    got = kernels.conv2d_wgrad(x, g, 3, 3, 1, 1)
    e = float((got.double() - ref).abs().max() / ref.abs().max())
    if not e < 2e-5:
        bad += 1; print('wgrad_wino', (n, ci, co, h, w), e)
    # ---- ring kernel (+ tail) against the gather kernel
    n, i, o = rnd.randint(1, 4), 32 * rnd.randint(1, 6), 8 * rnd.randint(1, 20)
    h, w = rnd.randint(3, 80), rnd.randint(3, 80)
    xh = torch.randn(n, i, h, w, device='cuda').half().to(memory_format=CL)
    wt = (torch.randn(o, i, 3, 3, device='cuda') / (i * 9) ** 0.5).half()
    kw, b = {}, (torch.randn(o, device='cuda') if rnd.random() < 0.6 else None)
    if rnd.random() < 0.7:
        kw.update(act=rnd.random() < 0.7, gain=rnd.choice([1.0, 0.5, 2 ** 0.5]), clamp=rnd.choice([256.0, 1.0, -1.0]))
        if rnd.random() < 0.5:
            kw['out_scale'] = torch.rand(n, o, device='cuda') + 0.5
        if rnd.random() < 0.5 and w % 4 == 0:
            kw['noise'], kw['noise_strength'] = (torch.randn(h, w, device='cuda') if rnd.random() < 0.5 else torch.randn(n, 1, h, w, device='cuda')), 0.3
    a, c = routes(lambda: kf.conv2d(xh, wt, b, 1, 1, **kw))
    if not torch.equal(a, c):
        bad += 1; print('ring vs gather', (n, i, o, h, w), {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()}, int((a != c).sum()))
    if not kw:
        ref = F.conv2d(xh.double(), wt.double(), None if b is None else b.double(), 1, 1)
        e = float((a.double() - ref).abs().max() / ref.abs().max())
        if not e < 2e-3:
            bad += 1; print('ring vs float64', (n, i, o, h, w), e)
    # ---- merged-phase transposed kernel
    n, i, o = rnd.randint(1, 4), 32 * rnd.randint(1, 6), 8 * rnd.randint(1, 16)
    h, w, pad = rnd.randint(1, 50), rnd.randint(1, 50), rnd.choice([0, 1])
    out_hw = None if rnd.random() < 0.6 else (2 * h + rnd.randint(-1, 2), 2 * w + rnd.randint(-1, 2))
    if out_hw is None and (2 * h + 1 - 2 * pad < 1 or 2 * w + 1 - 2 * pad < 1):
        pad = 0
    xh = torch.randn(n, i, h, w, device='cuda').half().to(memory_format=CL)
    wt = (torch.randn(i, o, 3, 3, device='cuda') / (i * 9 / 4) ** 0.5).half()
    s = (torch.rand(n, i, device='cuda') + 0.5) if rnd.random() < 0.5 else None
    a, c = routes(lambda: kf.conv_transpose2d(xh, wt, None, pad, out_hw, in_scale=s))
    if not torch.equal(a, c):
        bad += 1; print('upring vs per-phase', (n, i, o, h, w, pad, out_hw, s is not None), int((a != c).sum()))
    xs = xh.double() if s is None else (xh * s.half().reshape(n, i, 1, 1)).double()
    ref = F.conv_transpose2d(xs, wt.double(), stride=2, padding=0)
    oh, ow = out_hw if out_hw else (ref.shape[2] - 2 * pad, ref.shape[3] - 2 * pad)
    full = torch.zeros(n, o, pad + oh + 4, pad + ow + 4, dtype=torch.float64, device='cuda')
    full[:, :, :ref.shape[2], :ref.shape[3]] = ref
    ref = full[:, :, pad:pad + oh, pad:pad + ow]
    e = float((a.double() - ref).abs().max() / max(float(ref.abs().max()), 1e-30))
    if not e < 2e-3:
        bad += 1; print('upring vs float64', (n, i, o, h, w, pad, out_hw), e)
    # ---- modulation tail backward with the second product: A'(y) (gy d + u e) and sum_hw gz t
    n, c8, hw = rnd.randint(1, 4), rnd.choice([1, 2, 4, 8]), 4 * rnd.randint(1, 60)
    c, hh = 8 * c8, rnd.choice([1, 2, 4])
    while hw % hh:
        hh //= 2
    shape = (n, c, hh, hw // hh)
    for half in (False, True):
        dt = torch.float16 if half else torch.float32
        mk = lambda: (torch.randn(shape, device='cuda').to(dt).contiguous(memory_format=CL) if half else torch.randn(shape, device='cuda'))
        gy, y, t, u = mk(), mk(), mk(), mk()
        d, ee = torch.rand(n, c, device='cuda') + 0.5, torch.randn(n, c, device='cuda')
        act = rnd.random() < 0.7
        mod = kf if half else kernels
        gt, s1, s0, _ = mod.modtail_backward(gy, y, t, d, want_sums=True, want_noise=False, act=act, gain=1.0, clamp=1.5, u=u, e=ee)
        yd = y.double()
        slope = torch.where(yd.abs() >= 1.5, torch.zeros_like(yd), torch.where(yd > 0, torch.full_like(yd, 2 ** 0.5), torch.full_like(yd, 0.2 * 2 ** 0.5))) if act else torch.ones_like(yd)
        gz = gy.double() * slope
        ref_gt = gz * d.double().reshape(n, c, 1, 1) + u.double() * slope * ee.double().reshape(n, c, 1, 1)
        ref_s1 = (gz * t.double()).sum([2, 3])
        tol = 4e-3 if half else 2e-5
        e1 = float((gt.double() - ref_gt).abs().max() / ref_gt.abs().max())
        e2 = float((s1.double() - ref_s1).abs().max() / max(float(ref_s1.abs().max()), 1e-30))
        if not (e1 < tol and e2 < (2e-3 if half else 2e-5)):
            bad += 1; print('modtail second product', shape, half, act, e1, e2)
torch.cuda.synchronize()
print(f'fuzz_round5: {cases} cases, {bad} mismatches')
sys.exit(1 if bad else 0)
