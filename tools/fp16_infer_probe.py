import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, shgan_amd
from shgan_amd import _lib
if os.environ.get('SHG_VARIANT'):        # A/B: python sh-gan_amd/build.py --variant=<tag> [-D...]
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libshgan_hip_%s.so' % os.environ['SHG_VARIANT']))
from shgan_amd import configs, eval_harness, kernels
dev='cuda:0'
for fp16 in (False, True):
    G = configs.seeded_init_(configs.build_generator(512, **(dict(use_fp16_before_res=64, use_fp16_after_res=32) if fp16 else {})), seed=0).eval().requires_grad_(False).to(dev)
    x, z, _, _ = eval_harness.synthetic_items(list(range(16)), 512, 512, seed=1000, device=dev)
    def step(): return eval_harness.run_generator(G, x, z, noise_mode='random')
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): out = step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print('fp16 blocks' if fp16 else 'fp32', f'{dt*1e3:.2f} ms/step  {16/dt:.1f} img/s')
    if fp16:
        kt = kernels.KernelTimer(); kernels.set_timer(kt); step(); torch.cuda.synchronize(); kernels.set_timer(None)
        for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['ms'])[:10]: print('   ', k, round(v['ms'],2), v['calls'])
