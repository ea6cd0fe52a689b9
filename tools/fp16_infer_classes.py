"""Kernel classes (kernels.KernelTimer, HIP events, one stream) of one fp16-block inference step at 512 x 16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import configs, eval_harness, kernels
dev = 'cuda:0'
G = configs.seeded_init_(configs.build_generator(512, use_fp16_before_res=64, use_fp16_after_res=32), seed=0).eval().requires_grad_(False).to(dev)
x, z, _, _ = eval_harness.synthetic_items(list(range(16)), 512, 512, seed=1000, device=dev)
for _ in range(3):
    eval_harness.run_generator(G, x, z, noise_mode='random')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    eval_harness.run_generator(G, x, z, noise_mode='random')
e1.record(); torch.cuda.synchronize()
print(f'one stream: {e0.elapsed_time(e1) / 5:.2f} ms per step')
t = kernels.KernelTimer(); kernels.set_timer(t)
eval_harness.run_generator(G, x, z, noise_mode='random'); torch.cuda.synchronize(); kernels.set_timer(None)
tot = 0
for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    print(f'  {v["ms"]:7.3f} ms {v["calls"]:4d}x {k}' + (f'   {v["work"] / v["ms"] / 1e9:8.1f} TFLOP/s' if k.startswith('conv') else ''))
print(f'  sum of classes {tot:.2f} ms')
