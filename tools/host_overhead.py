"""Host-side (Python + ctypes + HIP launch) time per generator step vs GPU time: how far the CPU runs ahead of the GPU.
usage: python tools/host_overhead.py [resolution] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import configs, eval_harness

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
G = configs.seeded_init_(configs.build_generator(res), seed=0)
G = G.eval().requires_grad_(False).cuda()
x, z, _, _ = eval_harness.synthetic_batch(batch, res, G.z_dim, seed=1, device='cuda', masks='bernoulli')
for _ in range(3):
    eval_harness.run_generator(G, x, z)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    eval_harness.run_generator(G, x, z)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{res}x{res} batch {batch}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, GPU-complete {1e3 * (t2 - t0) / n:.2f} ms/step')
