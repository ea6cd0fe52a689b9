"""Per-LAUNCH efficiency of one training step (kernels.KernelTimer keeps every launch with its algorithmic work): for every kernel class the
launches whose work / time is far below the class's best -- small grids, serial loops, fixed costs that a class total hides.
usage: python tools/launch_outliers.py [--fp16] [--infer [--res 512 --batch 16]] [--min-us 15] [--ratio 0.3]"""
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
