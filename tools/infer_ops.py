"""torch.profiler view of one float32 inference step: the tensor ops that remain beside the HIP kernels.  usage: python tools/infer_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from torch.profiler import profile, ProfilerActivity
from shgan_amd import configs, eval_harness
dev = 'cuda:0'
G = configs.seeded_init_(configs.build_generator(512), seed=0).eval().requires_grad_(False).to(dev)
x, z, _, _ = eval_harness.synthetic_items(list(range(16)), 512, 512, seed=1000, device=dev)
for _ in range(3):
    eval_harness.run_generator(G, x, z, noise_mode='random')
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eval_harness.run_generator(G, x, z, noise_mode='random')
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = 0.0
for e in rows[:30]:
    tot += e.self_device_time_total
    print(f'{e.key:28s} calls {e.count:4d}  self {e.self_device_time_total:9.1f} us  total {e.device_time_total:9.1f} us  {str(e.input_shapes)[:90]}')
print('sum of self device time of aten ops: %.1f us' % sum(e.self_device_time_total for e in rows))
