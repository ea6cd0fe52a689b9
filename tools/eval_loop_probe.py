"""Where does the host time of one evaluation-loop batch go?  (round 6: EvalLoop measured host-bound at 19.5 ms per batch against 2.6 ms
for the generator step's own launches.)  Times the host side of every stage of a batch with perf_counter while three streams of
generator work keep the device busy, as in the loop."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shgan_amd  # noqa: E402,F401
from shgan_amd import configs, eval_harness as hz, masks  # noqa: E402

DEV = 'cuda:0'
R, B = 512, 16
G = configs.seeded_init_(configs.build_generator(R), seed=0).eval().requires_grad_(False).to(DEV)
loader = hz.PinnedU8Loader(list(range(B * 4)), B, R, seed=1, pool=4)
imgs = [im for im, _ in loader]
pipe = hz.StreamPipeline(DEV, depth=3)
copy = torch.cuda.Stream(device=DEV)
acc = {}


def tick(name, t0):
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


np.random.seed(0)
for it in range(14):
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    with torch.cuda.stream(copy):
        xd = imgs[it % 4].to(DEV, non_blocking=True)
    ev = torch.cuda.Event(); ev.record(copy)
    torch.cuda.current_stream().wait_event(ev)
    tick('h2d_issue', t0)
    t0 = time.perf_counter()
    recs = [masks.mask_attempt_records(R, (0, 1)) for _ in range(B)]
    tick('mask_records_host', t0)
    t0 = time.perf_counter()
    offs = [0]
    for r, _, _ in recs:
        offs.append(offs[-1] + len(r))
    m, holes = masks.rasterize(np.concatenate([r for r, _, _ in recs], 0), offs, [(int(a), int(b)) for _, a, b in recs], R, DEV)
    tick('raster_issue(3 pageable H2D + kernel)', t0)
    t0 = time.perf_counter()
    holes.cpu()
    tick('holes.cpu() sync', t0)
    t0 = time.perf_counter()
    x4 = hz.assemble_input(xd, m)
    z = torch.randn(B, G.z_dim, device=DEV)
    tick('assemble+randn', t0)
    t0 = time.perf_counter()
    out = pipe.run(lambda a, b: hz.run_generator(G, a, b, noise_mode='random'), x4, z)
    tick('G enqueue', t0)
    tick('batch total', t_all)
pipe.join()
torch.cuda.synchronize()
for k, v in acc.items():
    v = v[4:]
    print(f'{k:45s} mean {np.mean(v):7.3f} ms   min {np.min(v):7.3f}   max {np.max(v):7.3f}')

# ---- the product loop itself, its stages wrapped with host timers
print('--- EvalLoop (product) with wrapped stages')
from shgan_amd import datasets, fid_stats  # noqa: E402
acc2 = {}


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc2.setdefault(label or name, []).append((time.perf_counter() - t0) * 1e3)
    setattr(obj, name, w)


wrap(masks, 'random_masks')
wrap(masks, 'mask_attempt_records')
wrap(masks, 'rasterize')
wrap(hz, 'run_generator')
wrap(hz, 'assemble_input')
wrap(fid_stats.FidStats, 'add_shard')
wrap(datasets.DeviceFeeder, '_stage')
wrap(datasets.DeviceFeeder, '_finish')
feat = hz.standin_features


def feat_timed(u8):
    t0 = time.perf_counter()
    try:
        return feat(u8)
    finally:
        acc2.setdefault('standin_features', []).append((time.perf_counter() - t0) * 1e3)


NB = int(os.environ.get('PROBE_BATCHES', '12'))
if os.environ.get('PROBE_HISTORY'):
    # the history bench.py has behind it when its evaluation-loop block starts
    x, z, _, _ = hz.synthetic_items(list(range(B)), R, G.z_dim, seed=1000, device=DEV)
    pp = hz.StreamPipeline(DEV, depth=3)
    for _ in range(26):
        pp.run(lambda: hz.run_generator(G, x, z, noise_mode='random'))
    pp.join()
    if os.environ.get('PROBE_HISTORY') == 'graph':
        gp = hz.GraphPipeline(DEV, lambda x_, z_: hz.run_generator(G, x_, z_, noise_mode='random'), (x, z), depth=3)
        for _ in range(20):
            gp.run(x, z)
        gp.join()
        del gp
    torch.cuda.synchronize()
    del pp
    torch.cuda.empty_cache()
for rep in range(2):
    acc2.clear()
    loop = hz.EvalLoop(G, DEV, R, B * NB, noise_mode='random', feature_fn=feat_timed)
    ld = hz.PinnedU8Loader(loop.ids, B, R, seed=1, pool=4)
    ld._cache = loader._cache
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(ld)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'rep {rep}: host issue {t_issue / NB * 1e3:.2f} ms/batch, loop {t_all / NB * 1e3:.2f} ms/batch')
    for k, v in acc2.items():
        print(f'   {k:28s} calls {len(v):4d}  total/batch {np.sum(v) / NB:7.3f} ms   max {np.max(v):7.3f}')
