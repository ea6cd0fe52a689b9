"""Which Python lines issue the small library kernels of a G + D training step?  torch.profiler with stacks, grouped by (operator, innermost
frame inside this package).  usage: python tools/probes/aten_callsites.py [--fp16] [op substring ...]"""
import os, sys, runpy, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from torch.profiler import profile, ProfilerActivity
ops = [a for a in sys.argv[1:] if not a.startswith('--')] or ['fill_', 'zero_', 'copy_', 'add_', 'mul', 'sum', 'cat']
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a.startswith('--')]
ns = runpy.run_path(os.path.join(R, 'tools', 'train_step_bench.py'), run_name='bench')
g_phase, d_phase = ns['g_phase'], ns['d_phase']
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    g_phase(); d_phase()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith('aten::') or not any(o in e.name for o in ops) or e.self_device_time_total <= 0:
        continue
    site = 'autograd engine / unknown'
    for fr in (e.stack or []):
        if 'sh-gan_amd' in fr or 'shgan_amd' in fr or 'train_step_bench' in fr:
            site = fr.split('/')[-1] if '/' in fr else fr
            break
    a = agg[(e.name, site)]
    a[0] += e.self_device_time_total; a[1] += 1
for (name, site), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{t / 1e3:8.3f} ms {c:5d}x  {name:22s} {site[:110]}')
