"""Which call inside the mask rasteriser blocks once the host is many batches ahead?  Per-batch timeline of the product loop's stages."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shgan_amd  # noqa: E402,F401
from shgan_amd import configs, eval_harness as hz, masks, kernels  # noqa: E402

DEV = 'cuda:0'
R, B = 512, 16
NB = int(os.environ.get('PROBE_BATCHES', '20'))
G = configs.seeded_init_(configs.build_generator(R), seed=0).eval().requires_grad_(False).to(DEV)
log = []
T0 = [0.0]


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            if (t1 - t0) * 1e3 > 1.0:
                log.append((round((t0 - T0[0]) * 1e3, 1), label, round((t1 - t0) * 1e3, 2)))
    setattr(obj, name, w)


wrap(torch, 'empty', 'torch.empty')
wrap(torch, 'zeros', 'torch.zeros')
wrap(torch, 'stack', 'torch.stack')
wrap(torch.Tensor, 'to', 'Tensor.to')
wrap(torch.Tensor, 'cpu', 'Tensor.cpu')
wrap(masks, 'rasterize', 'rasterize')
wrap(hz, 'run_generator', 'run_generator')
for rep in range(2):
    log.clear()
    loop = hz.EvalLoop(G, DEV, R, B * NB, noise_mode='random', feature_fn=hz.standin_features if not os.environ.get('NOFID') else None, depth=int(os.environ.get('DEPTH', '3')),
                       device_masks=not os.environ.get('HOSTMASK'), feeder_stream=bool(os.environ.get('FEEDSTREAM')))
    ld = hz.PinnedU8Loader(loop.ids, B, R, seed=1, pool=int(os.environ.get('POOL', '4')))
    if os.environ.get('HOSTMASK'):
        inner = ld
        mk = torch.ones(B, R, R)
        class WithMasks:
            def __iter__(self):
                for img, ids in inner:
                    yield img, mk[:len(ids)], ids
        ld = WithMasks()
        ld._cache = inner._cache
    if rep:
        (inner if os.environ.get('HOSTMASK') else ld)._cache = cache
    np.random.seed(3)
    torch.cuda.synchronize()
    T0[0] = t0 = time.perf_counter()
    loop.run(ld)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    cache = ld._cache if not os.environ.get('HOSTMASK') else inner._cache
    print(f'rep {rep}: host issue {t_issue / NB * 1e3:.2f} ms/batch, loop {t_all / NB * 1e3:.2f} ms/batch; mem reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB')
print([ev for ev in log if ev[2] > 8])
