"""Does sample 0's output depend on its batch-mates?  (tests/test_gpu_bench_ranks.py compares uint8 digests across batch compositions.)
Runs the generator on ids [0,1,2,3] and [0,2,4,6] in one process and reports the first module whose output for sample 0 differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import shgan_amd
from shgan_amd import configs, eval_harness
dev = torch.device('cuda:0')
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = configs.build_generator(res); configs.seeded_init_(G, seed=0); G = G.eval().requires_grad_(False).to(dev)
outs = {}
def hook(name):
    def f(m, i, o):
        t = o[0] if isinstance(o, (tuple, list)) else o
        if isinstance(t, torch.Tensor): outs.setdefault(tag[0], []).append((name, t[0].detach().clone() if t.ndim > 0 and t.shape[0] == 4 else t.detach().clone()))
    return f
tag = ['a']
for n, m in G.named_modules():
    if n: m.register_forward_hook(hook(n))
res_ = {}
for t, ids in (('a', [0, 1, 2, 3]), ('b', [0, 2, 4, 6]), ('c', [0, 1, 2, 3])):
    tag[0] = t
    x, z, _, _ = eval_harness.synthetic_items(ids, res, G.z_dim, seed=1000, device=dev)
    res_[t] = eval_harness.run_generator(G, x, z, noise_mode='const')
    torch.cuda.synchronize()
print('uint8 sample 0 equal a/b:', torch.equal(res_['a'][0], res_['b'][0]), ' a/c (same ids):', torch.equal(res_['a'], res_['c']))
print('differing pixels a/b:', int((res_['a'][0] != res_['b'][0]).sum()))
for (na, ta), (nb, tb) in zip(outs['a'], outs['b']):
    if ta.shape == tb.shape and not torch.equal(ta, tb):
        d = (ta.float() - tb.float()).abs().max().item()
        print(f'first difference at module {na}: max abs {d:.3e}, shape {tuple(ta.shape)}')
        break
else:
    print('no module output differs for sample 0')
