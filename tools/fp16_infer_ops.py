"""torch.profiler view of one fp16-block inference step (which tensor ops remain beside the HIP kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from torch.profiler import profile, ProfilerActivity
from shgan_amd import configs, eval_harness
dev = 'cuda:0'
G = configs.seeded_init_(configs.build_generator(512, use_fp16_before_res=64, use_fp16_after_res=32), seed=0).eval().requires_grad_(False).to(dev)
x, z, _, _ = eval_harness.synthetic_items(list(range(16)), 512, 512, seed=1000, device=dev)
for _ in range(3):
    eval_harness.run_generator(G, x, z, noise_mode='random')
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    eval_harness.run_generator(G, x, z, noise_mode='random')
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=28, max_name_column_width=55, max_shapes_column_width=70))
