"""How fast is the CPU oracle on this host at different thread counts?  (informational, for cpu_baseline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import shgan_oracle as orc
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sd = orc.init_state_dict(res, seed=0)
x, z, _, _ = orc.synthetic_batch(2, res, 512, seed=1)
print('cpu_count', os.cpu_count())
for th in [int(v) for v in (sys.argv[2:] or [16, 32, 64, 128])]:
    torch.set_num_threads(th)
    with torch.no_grad():
        orc.generator_forward(sd, x[:1], z[:1], res, noise_mode='none')
        t0 = time.perf_counter(); orc.generator_forward(sd, x, z, res, noise_mode='const'); dt = time.perf_counter() - t0
    print(f'threads {th}: {2 / dt:.3f} img/s ({dt:.1f} s for 2 images)', flush=True)
