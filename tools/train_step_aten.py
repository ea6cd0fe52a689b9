"""ATen operators (the glue around the HIP kernels) of one G + D training step, by device time.  usage: python tools/train_step_aten.py [--fp16]   (SHG_ATEN_STACKS=1: call sites)
SHG_ATEN_STACKS=1: for the largest (operator, shapes) groups the innermost frames of this repository that issued them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy, torch
from torch.profiler import profile, ProfilerActivity
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_step_bench.py'), run_name='bench')
g_phase, d_phase = ns['g_phase'], ns['d_phase']
STACKS = bool(os.environ.get('SHG_ATEN_STACKS'))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=STACKS) as prof:
    g_phase(); d_phase()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:110]) for e in ev if e.key.startswith('aten::') and e.self_device_time_total > 0]
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'aten:: operators with device time: {tot / 1e3:.2f} ms in {sum(r[1] for r in rows)} calls')
byop = {}
for t, c, k, s in rows:
    a = byop.setdefault(k, [0, 0]); a[0] += t; a[1] += c
for k, (t, c) in sorted(byop.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'  {k:34s} {t / 1e3:8.3f} ms {c:6d} calls')
print('largest (op, shapes):')
for t, c, k, s in rows[:70]:
    print(f'  {t / 1e3:8.3f} ms {c:5d}x {k:28s} {s}')

if STACKS:
    print('by call site (innermost frames of this repository):')
    ev2 = prof.key_averages(group_by_input_shape=True, group_by_stack_n=24)
    rows2 = []
    for e in ev2:
        if not e.key.startswith('aten::') or e.self_device_time_total <= 0:
            continue
        fr = [f for f in (e.stack or []) if ('sh-gan_amd' in f or 'losses' in f or 'train_step_bench' in f) and 'profiler' not in f][:3]
        rows2.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:48], ' <- '.join(f.split('sh-gan_amd/')[-1][:80] for f in fr)))
    rows2.sort(reverse=True)
    for t, c, k, shp, where in rows2[:60]:
        print(f'  {t / 1e3:7.3f} ms {c:4d}x {k:16s} {shp:48s} {where}')
