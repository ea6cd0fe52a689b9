"""Library (aten::) operators and HIP-kernel classes of the lazy-regulariser phases (Greg: path length, Dreg: R1) at full width.
usage: python tools/probes/reg_phases_aten.py [--fp16]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import shgan_amd
from shgan_amd import losses
from test_gpu_config5 import build_networks
DEV = torch.device('cuda:0')
G, D = build_networks(512, 61, 62, fp16='--fp16' in sys.argv)
G.requires_grad_(False); D.requires_grad_(False)
rs = np.random.RandomState(63)
real = torch.from_numpy(rs.uniform(-1, 1, size=(8, 3, 512, 512)).astype(np.float32))
mask = torch.from_numpy((rs.uniform(size=(8, 1, 512, 512)) < 0.7).astype(np.float32))
real4 = torch.cat([mask - 0.5, real], dim=1).to(DEV)
L = losses.InpaintingLoss(DEV, G, D, noise_mode='random', style_mixing_prob=0.9)
z, c = torch.randn(8, 512, device=DEV), torch.zeros(8, 0, device=DEV)
for phase, mod in (('Greg', G), ('Dreg', D)):
    def run():
        mod.requires_grad_(True)
        for p in mod.parameters():
            p.grad = None
        L.accumulate_gradients(phase, real4, c, z, c, gain=4 if phase == 'Greg' else 16)
        mod.requires_grad_(False)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        run(); torch.cuda.synchronize()
    ev = prof.key_averages(group_by_input_shape=True)
    rows = sorted(((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]) for e in ev if e.key.startswith('aten::') and e.self_device_time_total > 0), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f'== {phase}: {e0.elapsed_time(e1):.1f} ms per pass; aten:: device time {tot / 1e3:.2f} ms in {sum(r[1] for r in rows)} calls')
    for t, n, k, s in rows[:22]:
        print(f'  {t / 1e3:8.3f} ms {n:5d}x {k:26s} {s}')
