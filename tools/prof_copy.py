import sys, os
sys.path.insert(0, os.getcwd())
import torch, shgan_amd
from shgan_amd import configs, eval_harness
from torch.profiler import profile, ProfilerActivity
dev='cuda:0'
G = configs.seeded_init_(configs.build_generator(512), seed=0).to(dev).eval()
x, z, _, _ = eval_harness.synthetic_batch(16, 512, 512, seed=1, device=dev, masks='bernoulli')
c = torch.zeros(16, 0, device=dev)
with torch.no_grad():
    for _ in range(2): G(x=x, z=z, c=c, noise_mode='random')
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        G(x=x, z=z, c=c, noise_mode='random')
        torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::cat', 'aten::contiguous', 'aten::clone', 'aten::_to_copy', 'aten::index', 'aten::add', 'aten::mul', 'aten::add_', 'aten::arange', 'aten::randint', 'aten::rand'):
        st = [s for s in (ev.stack or []) if 'sh-gan_amd' in s or 'shgan_amd' in s]
        cnt[(ev.name, st[0] if st else '?')] += 1
for k, v in cnt.most_common(40): print(v, k)
