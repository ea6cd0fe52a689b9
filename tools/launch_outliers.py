"""Per-LAUNCH efficiency of one training step (kernels.KernelTimer keeps every launch with its algorithmic work): for every kernel class the
launches whose work / time is far below the class's best -- small grids, serial loops, fixed costs that a class total hides.
usage: python tools/launch_outliers.py [--fp16] [--infer [--res 512 --batch 16] | --reg Greg|Dreg] [--min-us 15] [--ratio 0.3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy, torch
from shgan_amd import kernels
arg = lambda k, d: float(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d          # noqa: E731
min_us, ratio = arg('--min-us', 15.0), arg('--ratio', 0.3)
if '--infer' in sys.argv:              # the evaluation step instead: generator forward + composite at 512 x 16 (or --res R --batch B)
    from shgan_amd import configs, eval_harness
    res, batch = int(arg('--res', 512)), int(arg('--batch', 16))
    kw = dict(use_fp16_before_res=64, use_fp16_after_res=32) if '--fp16' in sys.argv else {}
    G = configs.seeded_init_(configs.build_generator(res, **kw), seed=0).eval().requires_grad_(False).to('cuda:0')
    x, z, _, _ = eval_harness.synthetic_items(list(range(batch)), res, 512, seed=1000, device='cuda:0')
    for _ in range(3):
        eval_harness.run_generator(G, x, z, noise_mode='random')
    torch.cuda.synchronize()
    t = kernels.KernelTimer(); kernels.set_timer(t)
    eval_harness.run_generator(G, x, z, noise_mode='random'); torch.cuda.synchronize(); kernels.set_timer(None)
elif '--reg' in sys.argv:              # a lazy-regulariser phase (Greg: path length on batch / 2; Dreg: R1) of config 5 at full width
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from shgan_amd import losses
    from test_gpu_config5 import build_networks
    phase = sys.argv[sys.argv.index('--reg') + 1]
    DEV = torch.device('cuda:0')
    G, D = build_networks(512, 61, 62, fp16='--fp16' in sys.argv)
    G.requires_grad_(False); D.requires_grad_(False)
    rs = np.random.RandomState(63)
    real = torch.from_numpy(rs.uniform(-1, 1, size=(8, 3, 512, 512)).astype(np.float32))
    mask = torch.from_numpy((rs.uniform(size=(8, 1, 512, 512)) < 0.7).astype(np.float32))
    real4 = torch.cat([mask - 0.5, real], dim=1).to(DEV)
    Lz = losses.InpaintingLoss(DEV, G, D, composite_fake=True, noise_mode='random', style_mixing_prob=0.9)
    z, c = torch.randn(8, 512, device=DEV), torch.zeros(8, 0, device=DEV)
    mod = G if phase.startswith('G') else D

    def run():
        mod.requires_grad_(True)
        for p_ in mod.parameters():
            p_.grad = None
        Lz.accumulate_gradients(phase, real4, c, z, c, gain={'Greg': 4, 'Dreg': 16}.get(phase, 1))
        mod.requires_grad_(False)
    run(); run(); torch.cuda.synchronize()
    t = kernels.KernelTimer(); kernels.set_timer(t)
    run(); torch.cuda.synchronize(); kernels.set_timer(None)
else:
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a == '--fp16']
    ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_step_bench.py'), run_name='bench')
    g_phase, d_phase = ns['g_phase'], ns['d_phase']
    t = kernels.KernelTimer(); kernels.set_timer(t)
    g_phase(); d_phase(); torch.cuda.synchronize(); kernels.set_timer(None)
tot_waste = 0.0
for cls, recs in sorted(t.records.items(), key=lambda kv: -sum(r[0].elapsed_time(r[1]) for r in kv[1])):
    rows = [(r[0].elapsed_time(r[1]) * 1e3, r[2]) for r in recs]
    rates = [w / us for us, w in rows if us > 0 and w > 0]
    if not rates:
        continue
    best = sorted(rates)[int(0.9 * (len(rates) - 1))]                 # 90th percentile rate of the class
    slow = [(us, w) for us, w in rows if us >= min_us and w > 0 and w / us < ratio * best]
    waste = sum(us - w / best for us, w in slow)
    tot_waste += waste
    total = sum(us for us, _ in rows)
    print(f'{cls:18s} {len(rows):4d} launches {total / 1e3:7.2f} ms | class rate (p90) {best / 1e3:9.1f} G/s | {len(slow):3d} slow launches, {waste / 1e3:6.2f} ms above the class rate')
    for us, w in sorted(slow, reverse=True)[:6]:
        print(f'        {us:8.1f} us   work {w / 1e6:10.2f} M   {w / us / 1e3:9.1f} G/s')
print(f'sum over classes of time above the class rate: {tot_waste / 1e3:.2f} ms')
