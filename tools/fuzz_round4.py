"""Random-shape fuzz of the round-4 kernels against torch: relayout (both directions), style_factors (forward + first-order backward vs
float64 autograd), direct weight pack (vs the staged pack).  usage: python tools/fuzz_round4.py [cases=200] [seed=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import kernels, kernels_f16
from shgan_amd.model_zoo import stylegan as sg
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed); torch.manual_seed(seed)
bad = 0
for it in range(cases):
    # relayout
    n, c, h, w = rnd.randint(1, 5), rnd.choice([1, 3, 4, 8, 16, 24, 40, 64, 72, 128, 520]), rnd.randint(1, 70), rnd.randint(1, 70)
    x = torch.randn(n, c, h, w, device='cuda') * 10 ** rnd.uniform(-3, 5)
    if kernels_f16.relayout_supported(x):
        ref = x.to(dtype=torch.float16, memory_format=torch.channels_last)
        a = kernels_f16.relayout(x); b = kernels_f16.relayout(ref)
        if not (torch.equal(a, ref) and torch.equal(b, ref.float())):
            bad += 1; print('relayout mismatch', (n, c, h, w))
    # style factors
    N, I, O, half = rnd.randint(1, 16), rnd.choice([8, 32, 64, 100, 128, 256, 512]), rnd.choice([3, 8, 64, 70, 128, 512]), rnd.random() < 0.5
    if N * I <= 8192:
        s64 = (torch.randn(N, I, dtype=torch.float64, device='cuda') + rnd.uniform(-1, 1))
        w64 = torch.rand(O, I, dtype=torch.float64, device='cuda') * 0.02
        ga, gb = torch.randn(N, I, dtype=torch.float64, device='cuda'), torch.randn(N, O, dtype=torch.float64, device='cuda')
        def ref_fn(s, wq):
            if half:
                s = s / s.norm(float('inf'), dim=1, keepdim=True)
            s = s * s.square().mean().rsqrt()
            return s, (s.square().matmul(wq.t()) + 1e-8).rsqrt()
        outs = []
        for fn, dt in ((ref_fn, torch.float64), (lambda s, wq: sg._StyleFactorsFn.apply(s, wq, half), torch.float32)):
            s = s64.to(dt).clone().requires_grad_(True); wq = w64.to(dt).clone().requires_grad_(True)
            sn, d = fn(s, wq)
            gs, gw = torch.autograd.grad((sn * ga.to(dt)).sum() + (d * gb.to(dt)).sum(), [s, wq])
            outs.append((sn.detach().double(), d.detach().double(), gs.double(), gw.double()))
        for name, u, v in zip(('sn', 'd', 'gs', 'gw'), outs[1], outs[0]):
            e = float((u - v).abs().max() / (v.abs().max() + 1e-30))
            if not e < 5e-5:
                bad += 1; print('style_factors', name, (N, I, O, half), e)
    # weight pack
    o, i, k = rnd.choice([8, 24, 32, 64, 96, 128]), rnd.choice([3, 4, 16, 40, 64, 128]), rnd.choice([1, 3])
    wt = torch.randn(o, i, k, k, device='cuda').half()
    for tr in (False, True):
        for fl in (False, True):
            src = wt.transpose(0, 1).contiguous() if tr else wt
            kernels_f16.PACK_DIRECT = True; p1 = kernels_f16.pack_weight(src, transposed=tr, flip=fl)
            kernels_f16.PACK_DIRECT = False; p2 = kernels_f16.pack_weight(src, transposed=tr, flip=fl)
            kernels_f16.PACK_DIRECT = True
            if not torch.equal(p1.wp, p2.wp):
                bad += 1; print('pack mismatch', (o, i, k, tr, fl))
torch.cuda.synchronize()
print(f'fuzz_round4: {cases} cases, {bad} mismatches')
sys.exit(1 if bad else 0)
