"""The lazy-regulariser phases (Greg: path length on batch / 2, Dreg: R1) of BASELINE config 5 at full width, one pass each: HIP-event time,
library (aten::) operators with their device time and launch count, and this package's kernel classes (kernels.KernelTimer).
usage: python tools/reg_phases.py [--fp16] [--phases Greg,Dreg,Gmain,Dmain] [--ab-tail | --ab-style]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import shgan_amd
from shgan_amd import losses, kernels
from test_gpu_config5 import build_networks
DEV = torch.device('cuda:0')
G, D = build_networks(512, 61, 62, fp16='--fp16' in sys.argv)
G.requires_grad_(False); D.requires_grad_(False)
rs = np.random.RandomState(63)
real = torch.from_numpy(rs.uniform(-1, 1, size=(8, 3, 512, 512)).astype(np.float32))
mask = torch.from_numpy((rs.uniform(size=(8, 1, 512, 512)) < 0.7).astype(np.float32))
real4 = torch.cat([mask - 0.5, real], dim=1).to(DEV)
L = losses.InpaintingLoss(DEV, G, D, composite_fake=True, noise_mode='random', style_mixing_prob=0.9)
z, c = torch.randn(8, 512, device=DEV), torch.zeros(8, 0, device=DEV)
names = sys.argv[sys.argv.index('--phases') + 1].split(',') if '--phases' in sys.argv else ['Greg', 'Dreg']
if '--ab-tail' in sys.argv or '--ab-style' in sys.argv:          # same process, alternating: a closed backward node against the tensor-operator composition
    from shgan_amd.model_zoo.stylegan_utils import grad_ops
    from shgan_amd.model_zoo import stylegan as _sg
    for phase in names:
        mod = G if phase.startswith('G') else D

        def run():
            mod.requires_grad_(True)
            for p in mod.parameters():
                p.grad = None
            L.accumulate_gradients(phase, real4, c, z, c, gain={'Greg': 4, 'Dreg': 16}.get(phase, 1))
            mod.requires_grad_(False)
        ts = {True: [], False: []}
        for rep in range(5):
            for closed in (True, False):
                if '--ab-style' in sys.argv:
                    _sg.CLOSED_STYLE_FACTORS_BACKWARD = closed
                else:
                    grad_ops.CLOSED_TAIL_BACKWARD = closed
                run(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
                ts[closed].append(e0.elapsed_time(e1) / 2)
        print(f'{phase}: closed node {min(ts[True]):.1f} ms (median {sorted(ts[True])[2]:.1f}) | composed {min(ts[False]):.1f} ms (median {sorted(ts[False])[2]:.1f})')
    sys.exit(0)
for phase in names:
    mod = G if phase.startswith('G') else D

    def run():
        mod.requires_grad_(True)
        for p in mod.parameters():
            p.grad = None
        L.accumulate_gradients(phase, real4, c, z, c, gain={'Greg': 4, 'Dreg': 16}.get(phase, 1))
        mod.requires_grad_(False)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    timer = kernels.KernelTimer(); kernels.set_timer(timer); run(); torch.cuda.synchronize(); kernels.set_timer(None)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        run(); torch.cuda.synchronize()
    ev = prof.key_averages(group_by_input_shape=True)
    rows = sorted(((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]) for e in ev if e.key.startswith('aten::') and e.self_device_time_total > 0), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f'== {phase}: {e0.elapsed_time(e1):.1f} ms per pass (eager); aten:: device time {tot / 1e3:.2f} ms in {sum(r[1] for r in rows)} calls')
    for t, n, k, s in rows[:14]:
        print(f'  {t / 1e3:8.3f} ms {n:5d}x {k:26s} {s}')
    own = sorted(timer.summary().items(), key=lambda kv: -kv[1]['ms'])
    print(f'  -- kernel classes of this package: {sum(v["ms"] for _, v in own):.1f} ms in {sum(v["calls"] for _, v in own)} launches')
    for k, v in own[:16]:
        print(f'  {v["ms"]:8.3f} ms {v["calls"]:5d}x {k}')
